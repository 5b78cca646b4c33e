#!/usr/bin/env python3
"""Writes tests/golden/skimage_kat.json: the literal known-answer vectors of the reference's skimage facade tests
(data only -- inputs and expected outputs with the reference file:line they are transcribed from), plus expected
outputs that the reference's tests obtain by calling its own ndimage layer, computed here with scipy.ndimage 1.15.3
(the library the reference's ndimage layer is itself tested against).  scikit-image is not installable in this image,
so these vectors -- not in-test restatements -- are what pins cupyimg_amd/skimage.

    python tests/golden/make_skimage_kat.py
"""
import json
import os

import numpy as np
import scipy
import scipy.ndimage as sndi

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "cupyimg/skimage/morphology/tests/"
cases = []


def add(name, func, image, expected, cite, **kw):
    cases.append({"name": name, "func": func, "dtype": str(np.asarray(image).dtype), "image": np.asarray(image).tolist(),
                  "expected_dtype": str(np.asarray(expected).dtype), "expected": np.asarray(expected).tolist(), "cite": cite, **kw})


# ---- test_grey.py:238-275 float images (default cross element), :278-287 the same as uint16 (img_as_uint)
im = np.array([[0.55, 0.72, 0.6, 0.54, 0.42], [0.65, 0.44, 0.89, 0.96, 0.38], [0.79, 0.53, 0.57, 0.93, 0.07],
               [0.09, 0.02, 0.83, 0.78, 0.87], [0.98, 0.8, 0.46, 0.78, 0.12]])
lit = {
    "erosion": [[0.55, 0.44, 0.54, 0.42, 0.38], [0.44, 0.44, 0.44, 0.38, 0.07], [0.09, 0.02, 0.53, 0.07, 0.07],
                [0.02, 0.02, 0.02, 0.78, 0.07], [0.09, 0.02, 0.46, 0.12, 0.12]],
    "dilation": [[0.72, 0.72, 0.89, 0.96, 0.54], [0.79, 0.89, 0.96, 0.96, 0.96], [0.79, 0.79, 0.93, 0.96, 0.93],
                 [0.98, 0.83, 0.83, 0.93, 0.87], [0.98, 0.98, 0.83, 0.78, 0.87]],
    "opening": [[0.55, 0.55, 0.54, 0.54, 0.42], [0.55, 0.44, 0.54, 0.44, 0.38], [0.44, 0.53, 0.53, 0.78, 0.07],
                [0.09, 0.02, 0.78, 0.78, 0.78], [0.09, 0.46, 0.46, 0.78, 0.12]],
    "closing": [[0.72, 0.72, 0.72, 0.54, 0.54], [0.72, 0.72, 0.89, 0.96, 0.54], [0.79, 0.79, 0.79, 0.93, 0.87],
                [0.79, 0.79, 0.83, 0.78, 0.87], [0.98, 0.83, 0.78, 0.78, 0.78]],
}


def as_uint(a):          # skimage.util.img_as_uint for floats in [0, 1]: round(a * 65535)
    return np.rint(np.asarray(a) * 65535.0).astype(np.uint16)


for f, want in lit.items():
    add("float_" + f, f, im, np.array(want), REF + "test_grey.py:238-275")
    add("uint16_" + f, f, as_uint(im), as_uint(want), REF + "test_grey.py:278-287")

# ---- test_grey.py:292-326 strided `out`
img = np.array([[5, 6, 2], [7, 2, 2], [3, 5, 1]], np.uint8)
add("strided_out_dilation", "dilation", img, np.array([[7, 0, 6, 0, 6], [0] * 5, [7, 0, 7, 0, 2], [0] * 5, [7, 0, 5, 0, 5]], np.uint8),
    REF + "test_grey.py:292-326", out_big_shape=[5, 5], out_step=2)
add("strided_out_erosion", "erosion", img, np.array([[5, 0, 2, 0, 2], [0] * 5, [2, 0, 2, 0, 1], [0] * 5, [3, 0, 1, 0, 1]], np.uint8),
    REF + "test_grey.py:292-326", out_big_shape=[5, 5], out_step=2)
# ---- test_grey.py:329-333
add("erosion_1d", "erosion", np.array([1, 2, 3, 2, 1]), np.array([1, 1, 2, 1, 1]), REF + "test_grey.py:329-333")

# ---- test_grey.py:222-236: default element == ndimage's 4-connected structure on the 9 x 9 pyramid
pyr = np.zeros((9, 9), np.uint8)
pyr[2:-2, 2:-2] = 128
pyr[3:-3, 3:-3] = 196
pyr[4, 4] = 255
cross = sndi.generate_binary_structure(2, 1)
add("pyramid_opening", "opening", pyr, sndi.grey_opening(pyr, footprint=cross), REF + "test_grey.py:222-236 (expected: scipy.ndimage.grey_opening)")
add("pyramid_closing", "closing", pyr, sndi.grey_closing(pyr, footprint=cross), REF + "test_grey.py:222-236 (expected: scipy.ndimage.grey_closing)")

# ---- test_binary.py:134-148: binary opening / closing of the uint16 pyramid == ndimage with the 4-connected structure
pyr16 = np.zeros((9, 9), np.uint16)
pyr16[2:-2, 2:-2] = 2 ** 14
pyr16[3:-3, 3:-3] = 2 ** 15
pyr16[4, 4] = 2 ** 16 - 1
add("binary_opening_pyramid", "binary_opening", pyr16, sndi.binary_opening(pyr16, structure=cross), REF + "test_binary.py:134-148")
add("binary_closing_pyramid", "binary_closing", pyr16, sndi.binary_closing(pyr16, structure=cross), REF + "test_binary.py:134-148")
# ---- test_binary.py:52-58: 17 x 17 element on a 20 x 20 mask (uint8 overflow of the element sum): binary == grey
big = np.zeros((20, 20), bool)
big[2:19, 2:19] = True
e = sndi.binary_erosion(big, structure=np.ones((17, 17), bool), border_value=1)      # skimage erosion: border_value=True (binary.py:42)
add("binary_erosion_17x17", "binary_erosion", big, e, REF + "test_binary.py:52-58", selem_ones=[17, 17])

# ---- docstring examples of grey.py (opening :265-292, closing :295-322, white_tophat :325-371, black_tophat :374-420)
bad = np.array([[1, 0, 0, 0, 1], [1, 1, 0, 1, 1], [1, 1, 1, 1, 1], [1, 1, 0, 1, 1], [1, 0, 0, 0, 1]], np.uint8)
add("doc_opening_square3", "opening", bad, np.array([[0] * 5, [1, 1, 0, 1, 1], [1, 1, 0, 1, 1], [1, 1, 0, 1, 1], [0] * 5], np.uint8),
    "cupyimg/skimage/morphology/grey.py opening docstring", selem_ones=[3, 3])
broken = np.zeros((5, 5), np.uint8)
broken[2] = [1, 1, 0, 1, 1]
want = np.zeros((5, 5), np.uint8)
want[2] = 1
add("doc_closing_square3", "closing", broken, want, "cupyimg/skimage/morphology/grey.py closing docstring", selem_ones=[3, 3])
bright = np.array([[2, 3, 3, 3, 2], [3, 4, 5, 4, 3], [3, 5, 9, 5, 3], [3, 4, 5, 4, 3], [2, 3, 3, 3, 2]], np.uint8)
th = np.array([[0] * 5, [0, 0, 1, 0, 0], [0, 1, 5, 1, 0], [0, 0, 1, 0, 0], [0] * 5], np.uint8)
add("doc_white_tophat_square3", "white_tophat", bright, th, "cupyimg/skimage/morphology/grey.py white_tophat docstring", selem_ones=[3, 3])
add("doc_black_tophat_square3", "black_tophat", (11 - bright).astype(np.uint8), th, "cupyimg/skimage/morphology/grey.py black_tophat docstring",
    selem_ones=[3, 3])

meta = {"generator": "tests/golden/make_skimage_kat.py", "scipy": scipy.__version__, "numpy": np.__version__,
        "note": "literal vectors transcribed from the reference's tests; 'expected: scipy.ndimage' cases computed with SciPy here"}
with open(os.path.join(HERE, "skimage_kat.json"), "w") as f:
    json.dump({"meta": meta, "cases": cases}, f, indent=0)
print(len(cases), "cases")
