"""Known-answer vectors of the reference's own tests, as data.

Every entry is (function, input arrays, keyword arguments, expected output)
transcribed from the literal arrays in
/root/reference/cupyimg/scipy/ndimage/tests/test_ndimage.py and
tests/test_filters.py (line numbers in `src`).  No reference code is kept --
only the numbers.  At generation time each vector is also checked against
SciPy (the reference's oracle) so a transcription slip cannot get committed.

    python tests/golden/make_kat_reference.py   ->  tests/golden/kat_reference.json
"""
import json
import os

import numpy as np
import scipy.ndimage as ndi

HERE = os.path.dirname(os.path.abspath(__file__))
MODES = ["nearest", "wrap", "reflect", "mirror", "constant"]   # test_ndimage.py:81
TYPES = ["int8", "uint8", "int16", "uint16", "int32", "uint32", "int64", "uint64", "float32", "float64"]
CASES = []


def K(func, arrays, kwargs, expected, src, in_dtype=None, decimal=6):
    CASES.append({"func": func, "arrays": arrays, "kwargs": kwargs, "expected": expected, "src": src,
                  "in_dtype": in_dtype, "decimal": decimal})


# ---------------------------------------------------------------- correlate 1-D / n-D (test_ndimage.py:83-236)
for fn in ["correlate", "convolve", "correlate1d", "convolve1d"]:
    K(fn, {"input": [1, 2], "weights": [2]}, {}, [2, 4], "test_ndimage.py:83-98")
    K(fn, {"input": [1, 2, 3], "weights": [1]}, {}, [1, 2, 3], "test_ndimage.py:100-114")
    K(fn, {"input": [1], "weights": [1, 1]}, {}, [2], "test_ndimage.py:116-131")
    K(fn, {"input": [1, 2, 3], "weights": [1, 2, 1]}, {}, [5, 8, 11], "test_ndimage.py:175-186")
for fn, exp in [("correlate", [2, 3]), ("convolve", [3, 4]), ("correlate1d", [2, 3]), ("convolve1d", [3, 4])]:
    K(fn, {"input": [1, 2], "weights": [1, 1]}, {}, exp, "test_ndimage.py:133-145")
for fn, exp in [("correlate", [2, 3, 5]), ("convolve", [3, 5, 6]), ("correlate1d", [2, 3, 5]), ("convolve1d", [3, 5, 6])]:
    K(fn, {"input": [1, 2, 3], "weights": [1, 1]}, {}, exp, "test_ndimage.py:147-159")
for fn, exp in [("correlate", [9, 14, 17]), ("convolve", [7, 10, 15]), ("correlate1d", [9, 14, 17]),
                ("convolve1d", [7, 10, 15])]:
    K(fn, {"input": [1, 2, 3], "weights": [1, 2, 3]}, {}, exp, "test_ndimage.py:161-173")
for fn, exp in [("correlate", [1, 2, 5]), ("convolve", [3, 6, 7]), ("correlate1d", [1, 2, 5]), ("convolve1d", [3, 6, 7])]:
    K(fn, {"input": [1, 2, 3], "weights": [1, 2, -1]}, {}, exp, "test_ndimage.py:188-200")
K("correlate", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 1], [1, 1]]}, {}, [[4, 6, 10], [10, 12, 16]],
  "test_ndimage.py:222-228")
K("convolve", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 1], [1, 1]]}, {}, [[12, 16, 18], [18, 22, 24]],
  "test_ndimage.py:222-228")
K("correlate", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]}, {}, [[2, 3, 5], [5, 6, 8]],
  "test_ndimage.py:230-236")
K("convolve", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]}, {}, [[6, 8, 9], [9, 11, 12]],
  "test_ndimage.py:230-236")
# dtype x dtype matrix (test_ndimage.py:238-263)
for t1 in TYPES:
    for t2 in TYPES:
        K("correlate", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]}, {"output": t2},
          [[2, 3, 5], [5, 6, 8]], "test_ndimage.py:238-249", in_dtype=t1)
        K("convolve", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]}, {"output": t2},
          [[6, 8, 9], [9, 11, 12]], "test_ndimage.py:238-249", in_dtype=t1)
for t1 in TYPES:
    K("correlate", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[0.5, 0], [0, 0.5]]}, {"output": "float32"},
      [[1, 1.5, 2.5], [2.5, 3, 4]], "test_ndimage.py:277-287", in_dtype=t1)
    K("convolve", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[0.5, 0], [0, 0.5]]}, {"output": "float32"},
      [[3, 4, 4.5], [4.5, 5.5, 6]], "test_ndimage.py:277-287", in_dtype=t1)
    K("correlate", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]},
      {"output": "float32", "mode": "nearest", "origin": -1}, [[6, 8, 9], [9, 11, 12]], "test_ndimage.py:303-317",
      in_dtype=t1)
    K("convolve", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]},
      {"output": "float32", "mode": "nearest", "origin": -1}, [[2, 3, 5], [5, 6, 8]], "test_ndimage.py:303-317",
      in_dtype=t1)
    K("correlate", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]},
      {"output": "float32", "mode": "nearest", "origin": [-1, 0]}, [[5, 6, 8], [8, 9, 11]],
      "test_ndimage.py:319-341", in_dtype=t1)
    K("convolve", {"input": [[1, 2, 3], [4, 5, 6]], "weights": [[1, 0], [0, 1]]},
      {"output": "float32", "mode": "nearest", "origin": [-1, 0]}, [[3, 5, 6], [6, 8, 9]],
      "test_ndimage.py:319-341", in_dtype=t1)
for fn, exp in [("correlate", [3, 5, 6]), ("convolve", [2, 3, 5]), ("correlate1d", [3, 5, 6]), ("convolve1d", [2, 3, 5])]:
    K(fn, {"input": [1, 2, 3], "weights": [1, 1]}, {"origin": -1}, exp, "test_ndimage.py:289-301")
# axis / mode / origin on 1-D passes, every dtype pair (test_ndimage.py:343-448)
for t1 in TYPES:
    for t2 in TYPES:
        a = [[1, 2, 3], [2, 4, 6]]
        for fn in ["correlate1d", "convolve1d"]:
            K(fn, {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "output": t2}, [[5, 10, 15], [7, 14, 21]],
              "test_ndimage.py:343-353", in_dtype=t1)
            K(fn, {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "mode": "wrap", "output": t2},
              [[6, 12, 18], [6, 12, 18]], "test_ndimage.py:364-378", in_dtype=t1)
            K(fn, {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "mode": "nearest", "output": t2},
              [[5, 10, 15], [7, 14, 21]], "test_ndimage.py:380-394", in_dtype=t1)
        K("correlate1d", {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "mode": "nearest", "origin": -1, "output": t2},
          [[7, 14, 21], [8, 16, 24]], "test_ndimage.py:396-421", in_dtype=t1)
        K("convolve1d", {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "mode": "nearest", "origin": -1, "output": t2},
          [[4, 8, 12], [5, 10, 15]], "test_ndimage.py:396-421", in_dtype=t1)
        K("correlate1d", {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "mode": "nearest", "origin": 1, "output": t2},
          [[4, 8, 12], [5, 10, 15]], "test_ndimage.py:423-448", in_dtype=t1)
        K("convolve1d", {"input": a, "weights": [1, 2, 1]}, {"axis": 0, "mode": "nearest", "origin": 1, "output": t2},
          [[7, 14, 21], [8, 16, 24]], "test_ndimage.py:423-448", in_dtype=t1)

# ---------------------------------------------------------------- uniform (test_ndimage.py:713-752)
K("uniform_filter1d", {"input": [2, 4, 6]}, {"size": 2, "origin": -1}, [3, 5, 6], "test_ndimage.py:713-717")
K("uniform_filter", {"input": [1, 2, 3]}, {"size": [0]}, [1, 2, 3], "test_ndimage.py:719-723")
K("uniform_filter", {"input": [1, 2, 3]}, {"size": [1]}, [1, 2, 3], "test_ndimage.py:725-729")
K("uniform_filter", {"input": [2, 4, 6]}, {"size": [2]}, [2, 3, 5], "test_ndimage.py:731-735")
for t1 in TYPES:
    for t2 in TYPES:
        K("uniform_filter", {"input": [[4, 8, 12], [16, 20, 24]]}, {"size": [2, 2], "output": t2},
          [[4, 6, 10], [10, 12, 16]], "test_ndimage.py:743-752", in_dtype=t1)
K("uniform_filter", {"input": [[1.0, 0.0, 0.0], [1.0, 1.0, 0.0], [0.0, 0.0, 0.0]]},
  {"size": 5, "mode": ["reflect", "wrap"]}, [[0.32, 0.40, 0.48], [0.20, 0.28, 0.32], [0.28, 0.32, 0.40]],
  "test_filters.py:299-310", in_dtype="float64")

# ---------------------------------------------------------------- min / max (test_ndimage.py:754-904)
A5 = [[3, 2, 5, 1, 4], [7, 6, 9, 3, 5], [5, 8, 3, 7, 1]]
FP = [[1, 0, 1], [1, 1, 0]]
K("minimum_filter", {"input": [1, 2, 3, 4, 5]}, {"size": [2]}, [1, 1, 2, 3, 4], "test_ndimage.py:754-758")
K("minimum_filter", {"input": [1, 2, 3, 4, 5]}, {"size": [3]}, [1, 1, 2, 3, 4], "test_ndimage.py:760-764")
K("minimum_filter", {"input": [3, 2, 5, 1, 4]}, {"size": [2]}, [3, 2, 2, 1, 1], "test_ndimage.py:766-770")
K("minimum_filter", {"input": [3, 2, 5, 1, 4]}, {"size": [3]}, [2, 2, 1, 1, 1], "test_ndimage.py:772-776")
K("minimum_filter", {"input": A5}, {"size": [2, 3]}, [[2, 2, 1, 1, 1], [2, 2, 1, 1, 1], [5, 3, 3, 1, 1]],
  "test_ndimage.py:778-786")
K("minimum_filter", {"input": A5, "footprint": [[1, 1, 1], [1, 1, 1]]}, {},
  [[2, 2, 1, 1, 1], [2, 2, 1, 1, 1], [5, 3, 3, 1, 1]], "test_ndimage.py:788-796")
K("minimum_filter", {"input": A5, "footprint": FP}, {}, [[2, 2, 1, 1, 1], [2, 3, 1, 3, 1], [5, 5, 3, 3, 1]],
  "test_ndimage.py:798-806")
K("minimum_filter", {"input": A5, "footprint": FP}, {"origin": -1}, [[3, 1, 3, 1, 1], [5, 3, 3, 1, 1], [3, 3, 1, 1, 1]],
  "test_ndimage.py:808-816")
K("minimum_filter", {"input": A5, "footprint": FP}, {"origin": [-1, 0]},
  [[2, 3, 1, 3, 1], [5, 5, 3, 3, 1], [5, 3, 3, 1, 1]], "test_ndimage.py:818-828")
K("maximum_filter", {"input": [1, 2, 3, 4, 5]}, {"size": [2]}, [1, 2, 3, 4, 5], "test_ndimage.py:830-834")
K("maximum_filter", {"input": [1, 2, 3, 4, 5]}, {"size": [3]}, [2, 3, 4, 5, 5], "test_ndimage.py:836-840")
K("maximum_filter", {"input": [3, 2, 5, 1, 4]}, {"size": [2]}, [3, 3, 5, 5, 4], "test_ndimage.py:842-846")
K("maximum_filter", {"input": [3, 2, 5, 1, 4]}, {"size": [3]}, [3, 5, 5, 5, 4], "test_ndimage.py:848-852")
K("maximum_filter", {"input": A5}, {"size": [2, 3]}, [[3, 5, 5, 5, 4], [7, 9, 9, 9, 5], [8, 9, 9, 9, 7]],
  "test_ndimage.py:854-862")
K("maximum_filter", {"input": A5, "footprint": [[1, 1, 1], [1, 1, 1]]}, {},
  [[3, 5, 5, 5, 4], [7, 9, 9, 9, 5], [8, 9, 9, 9, 7]], "test_ndimage.py:864-872")
K("maximum_filter", {"input": A5, "footprint": FP}, {}, [[3, 5, 5, 5, 4], [7, 7, 9, 9, 5], [7, 9, 8, 9, 7]],
  "test_ndimage.py:874-882")
K("maximum_filter", {"input": A5, "footprint": FP}, {"origin": -1}, [[7, 9, 9, 5, 5], [9, 8, 9, 7, 5], [8, 8, 7, 7, 7]],
  "test_ndimage.py:884-892")
K("maximum_filter", {"input": A5, "footprint": FP}, {"origin": [-1, 0]},
  [[7, 7, 9, 9, 5], [7, 9, 8, 9, 7], [8, 8, 8, 7, 7]], "test_ndimage.py:894-904")
# 1-D per mode (test_filters.py:415-441)
R10 = list(range(10))
K("minimum_filter1d", {"input": R10}, {"size": 1}, R10, "test_filters.py:417-419")
K("maximum_filter1d", {"input": R10}, {"size": 1}, R10, "test_filters.py:420-421")
K("minimum_filter1d", {"input": R10}, {"size": 5, "mode": "reflect"}, [0, 0, 0, 1, 2, 3, 4, 5, 6, 7], "test_filters.py:423-424")
K("maximum_filter1d", {"input": R10}, {"size": 5, "mode": "reflect"}, [2, 3, 4, 5, 6, 7, 8, 9, 9, 9], "test_filters.py:425-426")
K("minimum_filter1d", {"input": R10}, {"size": 5, "mode": "constant", "cval": -1}, [-1, -1, 0, 1, 2, 3, 4, 5, -1, -1],
  "test_filters.py:428-429")
K("maximum_filter1d", {"input": R10}, {"size": 5, "mode": "constant", "cval": 10}, [10, 10, 4, 5, 6, 7, 8, 9, 10, 10],
  "test_filters.py:430-431")
K("minimum_filter1d", {"input": R10}, {"size": 5, "mode": "nearest"}, [0, 0, 0, 1, 2, 3, 4, 5, 6, 7], "test_filters.py:433-434")
K("maximum_filter1d", {"input": R10}, {"size": 5, "mode": "nearest"}, [2, 3, 4, 5, 6, 7, 8, 9, 9, 9], "test_filters.py:435-436")
K("minimum_filter1d", {"input": R10}, {"size": 5, "mode": "wrap"}, [0, 0, 0, 1, 2, 3, 4, 5, 0, 0], "test_filters.py:438-439")
K("maximum_filter1d", {"input": R10}, {"size": 5, "mode": "wrap"}, [9, 9, 4, 5, 6, 7, 8, 9, 9, 9], "test_filters.py:440-441")

# ---------------------------------------------------------------- boundary modes (test_ndimage.py:1122-1272)
def extend(fn, arr, w, per_mode, src, **kw):
    for mode, exp in zip(MODES, per_mode):
        K(fn, {"input": arr, "weights": w}, dict(mode=mode, cval=0, **kw), exp, src)


extend("correlate1d", [1, 2, 3], [1, 0], [[1, 1, 2], [3, 1, 2], [1, 1, 2], [2, 1, 2], [0, 1, 2]],
       "test_ndimage.py:1122-1136", axis=0)
extend("correlate1d", [1, 2, 3], [1, 0, 0, 0, 0, 0, 0, 0], [[1, 1, 1], [3, 1, 2], [3, 3, 2], [1, 2, 3], [0, 0, 0]],
       "test_ndimage.py:1138-1150", axis=0)
extend("correlate1d", [1, 2, 3], [0, 0, 1], [[2, 3, 3], [2, 3, 1], [2, 3, 3], [2, 3, 2], [2, 3, 0]],
       "test_ndimage.py:1152-1166", axis=0)
extend("correlate1d", [1, 2, 3], [0, 0, 0, 0, 0, 0, 0, 0, 1], [[3, 3, 3], [2, 3, 1], [2, 1, 1], [1, 2, 3], [0, 0, 0]],
       "test_ndimage.py:1168-1180", axis=0)
extend("correlate", [[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[1, 0], [0, 0]],
       [[[1, 1, 2], [1, 1, 2], [4, 4, 5]], [[9, 7, 8], [3, 1, 2], [6, 4, 5]], [[1, 1, 2], [1, 1, 2], [4, 4, 5]],
        [[5, 4, 5], [2, 1, 2], [5, 4, 5]], [[0, 0, 0], [0, 1, 2], [0, 4, 5]]], "test_ndimage.py:1182-1194")
extend("correlate", [[1, 2, 3], [4, 5, 6], [7, 8, 9]], [[0, 0, 0], [0, 0, 0], [0, 0, 1]],
       [[[5, 6, 6], [8, 9, 9], [8, 9, 9]], [[5, 6, 4], [8, 9, 7], [2, 3, 1]], [[5, 6, 6], [8, 9, 9], [8, 9, 9]],
        [[5, 6, 5], [8, 9, 8], [5, 6, 5]], [[5, 6, 0], [8, 9, 0], [0, 0, 0]]], "test_ndimage.py:1196-1208")
extend("correlate", [1, 2, 3], [0, 0, 0, 0, 0, 0, 0, 0, 1], [[3, 3, 3], [2, 3, 1], [2, 1, 1], [1, 2, 3], [0, 0, 0]],
       "test_ndimage.py:1210-1224")
extend("correlate", [[1], [2], [3]], [[0], [0], [0], [0], [0], [0], [0], [0], [1]],
       [[[3], [3], [3]], [[2], [3], [1]], [[2], [1], [1]], [[1], [2], [3]], [[0], [0], [0]]],
       "test_ndimage.py:1226-1240")

# ---------------------------------------------------------------- binary morphology (test_ndimage.py:1466-2948)
for t in TYPES:
    K("binary_erosion", {"input": [1]}, {}, [0], "test_ndimage.py:1478-1482", in_dtype=t)
    K("binary_erosion", {"input": [1]}, {"border_value": 1}, [1], "test_ndimage.py:1484-1488", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 1]}, {}, [0, 1, 0], "test_ndimage.py:1490-1494", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 1]}, {"border_value": 1}, [1, 1, 1], "test_ndimage.py:1496-1500", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 1, 1, 1]}, {}, [0, 1, 1, 1, 0], "test_ndimage.py:1502-1506", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 0, 1, 1]}, {}, [0, 0, 0, 0, 0], "test_ndimage.py:1514-1519", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 0, 1, 1]}, {"border_value": 1}, [1, 0, 0, 0, 1], "test_ndimage.py:1521-1526",
      in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 0, 1, 1], "structure": [1, 0, 1]}, {"border_value": 1}, [1, 0, 1, 0, 1],
      "test_ndimage.py:1528-1534", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 0, 1, 1], "structure": [1, 0, 1]}, {"border_value": 1, "origin": -1},
      [0, 1, 0, 1, 1], "test_ndimage.py:1536-1544", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 0, 1, 1], "structure": [1, 0, 1]}, {"border_value": 1, "origin": 1},
      [1, 1, 0, 1, 0], "test_ndimage.py:1546-1552", in_dtype=t)
    K("binary_erosion", {"input": [1, 1, 0, 1, 1], "structure": [1, 1]}, {"border_value": 1}, [1, 1, 0, 0, 1],
      "test_ndimage.py:1554-1560", in_dtype=t)
D8 = [[0, 0, 0, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 1, 1], [0, 0, 1, 1, 1, 0, 1, 1],
      [0, 0, 1, 0, 1, 1, 0, 0], [0, 1, 0, 1, 1, 1, 1, 0], [0, 1, 1, 0, 0, 1, 1, 0], [0, 0, 0, 0, 0, 0, 0, 0]]
CROSS0 = [[0, 1, 0], [1, 0, 1], [0, 1, 0]]
for t in TYPES:
    K("binary_erosion", {"input": D8, "structure": CROSS0}, {"border_value": 1},
      [[0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 1, 0, 0],
       [0, 0, 0, 1, 0, 0, 0, 0], [0, 0, 1, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]],
      "test_ndimage.py:1733-1764", in_dtype=t)
    K("binary_erosion", {"input": D8, "structure": CROSS0}, {"border_value": 1, "origin": [-1, -1]},
      [[0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 1, 0, 0, 1], [0, 0, 1, 0, 0, 0, 0, 0],
       [0, 1, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 1]],
      "test_ndimage.py:1766-1799", in_dtype=t)
DIAMOND = [[0, 0, 0, 1, 0, 0, 0], [0, 0, 1, 1, 1, 0, 0], [0, 1, 1, 1, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1],
           [0, 1, 1, 1, 1, 1, 0], [0, 0, 1, 1, 1, 0, 0], [0, 0, 0, 1, 0, 0, 0]]
CROSS = [[0, 1, 0], [1, 1, 1], [0, 1, 0]]
CENTER = [[0] * 7, [0] * 7, [0] * 7, [0, 0, 0, 1, 0, 0, 0], [0] * 7, [0] * 7, [0] * 7]
K("binary_erosion", {"input": DIAMOND, "structure": CROSS}, {"border_value": 1, "iterations": 3}, CENTER,
  "test_ndimage.py:2255-2288", in_dtype="bool")
MASK8 = [[0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 1, 1, 1, 0, 0, 0], [0, 0, 1, 0, 1, 0, 0, 0],
         [0, 0, 1, 1, 1, 0, 0, 0], [0, 0, 1, 1, 1, 0, 0, 0], [0, 0, 1, 1, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0]]
TMP8 = [[0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 1], [0, 0, 0, 0, 1, 0, 0, 1], [0, 0, 1, 0, 0, 0, 0, 0],
        [0, 1, 0, 0, 1, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 0], [0, 0, 0, 0, 0, 0, 0, 1]]
_m, _t, _d = np.array(MASK8, bool), np.array(TMP8, bool), np.array(D8, bool)
K("binary_erosion", {"input": D8, "structure": CROSS0, "mask": MASK8}, {"border_value": 1, "origin": [-1, -1]},
  ((_t & _m) | (_d & ~_m)).astype(int).tolist(), "test_ndimage.py:2167-2215")
# dilation, 1-D cases with every dtype (test_ndimage.py:2337-2429)
for t in TYPES:
    K("binary_dilation", {"input": [1]}, {}, [1], "test_ndimage.py:2337-2341", in_dtype=t)
    K("binary_dilation", {"input": [0]}, {}, [0], "test_ndimage.py:2343-2347", in_dtype=t)
    K("binary_dilation", {"input": [1, 1, 1]}, {}, [1, 1, 1], "test_ndimage.py:2349-2353", in_dtype=t)
    K("binary_dilation", {"input": [0, 0, 0]}, {}, [0, 0, 0], "test_ndimage.py:2355-2359", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0]}, {}, [1, 1, 1], "test_ndimage.py:2361-2366", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 1, 0]}, {}, [1, 1, 1, 1, 1], "test_ndimage.py:2368-2374", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0]}, {}, [1, 1, 1, 0, 0], "test_ndimage.py:2376-2381", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0]}, {"origin": -1}, [0, 1, 1, 1, 0], "test_ndimage.py:2383-2388",
      in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0]}, {"origin": 1}, [1, 1, 0, 0, 0], "test_ndimage.py:2390-2395",
      in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0], "structure": [1, 0, 1]}, {}, [1, 0, 1, 0, 0],
      "test_ndimage.py:2397-2403", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0], "structure": [1, 0, 1]}, {"border_value": 1}, [1, 0, 1, 0, 1],
      "test_ndimage.py:2405-2411", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0], "structure": [1, 0, 1]}, {"origin": -1}, [0, 1, 0, 1, 0],
      "test_ndimage.py:2413-2419", in_dtype=t)
    K("binary_dilation", {"input": [0, 1, 0, 0, 0], "structure": [1, 0, 1]}, {"origin": -1, "border_value": 1},
      [1, 1, 0, 1, 0], "test_ndimage.py:2421-2429", in_dtype=t)
    K("binary_dilation", {"input": [[1]]}, {}, [[1]], "test_ndimage.py:2431-2435", in_dtype=t)

# ---------------------------------------------------------------- structures (test_ndimage.py:1396-1464)
K("generate_binary_structure", {}, {"rank": 0, "connectivity": 1}, 1, "test_ndimage.py:1396-1398")
K("generate_binary_structure", {}, {"rank": 1, "connectivity": 1}, [1, 1, 1], "test_ndimage.py:1400-1402")
K("generate_binary_structure", {}, {"rank": 2, "connectivity": 1}, [[0, 1, 0], [1, 1, 1], [0, 1, 0]],
  "test_ndimage.py:1404-1406")
K("generate_binary_structure", {}, {"rank": 2, "connectivity": 2}, [[1, 1, 1], [1, 1, 1], [1, 1, 1]],
  "test_ndimage.py:1412-1414")

# ---------------------------------------------------------------- grey morphology (test_ndimage.py:3234-3316)
FPD = [[0, 1, 1], [1, 0, 1]]
K("grey_erosion", {"input": A5, "footprint": FP}, {}, [[2, 2, 1, 1, 1], [2, 3, 1, 3, 1], [5, 5, 3, 3, 1]],
  "test_ndimage.py:3234-3244")
K("grey_erosion", {"input": A5, "footprint": FP, "structure": [[0, 0, 0], [0, 0, 0]]}, {},
  [[2, 2, 1, 1, 1], [2, 3, 1, 3, 1], [5, 5, 3, 3, 1]], "test_ndimage.py:3246-3259")
K("grey_erosion", {"input": A5, "footprint": FP, "structure": [[1, 1, 1], [1, 1, 1]]}, {},
  [[1, 1, 0, 0, 0], [1, 2, 0, 2, 0], [4, 4, 2, 2, 0]], "test_ndimage.py:3261-3274")
K("grey_dilation", {"input": A5, "footprint": FPD}, {}, [[7, 7, 9, 9, 5], [7, 9, 8, 9, 7], [8, 8, 8, 7, 7]],
  "test_ndimage.py:3276-3286")
K("grey_dilation", {"input": A5, "footprint": FPD, "structure": [[0, 0, 0], [0, 0, 0]]}, {},
  [[7, 7, 9, 9, 5], [7, 9, 8, 9, 7], [8, 8, 8, 7, 7]], "test_ndimage.py:3288-3301")
K("grey_dilation", {"input": A5, "footprint": FPD, "structure": [[1, 1, 1], [1, 1, 1]]}, {},
  [[8, 8, 10, 10, 6], [8, 10, 9, 10, 8], [9, 9, 9, 8, 8]], "test_ndimage.py:3303-3316")


def main():
    bad = 0
    for c in CASES:
        if c["func"] == "generate_binary_structure":
            got = ndi.generate_binary_structure(c["kwargs"]["rank"], c["kwargs"]["connectivity"])
        else:
            arrs = {k: np.asarray(v) for k, v in c["arrays"].items()}
            if c["in_dtype"]:
                arrs["input"] = arrs["input"].astype(c["in_dtype"])
            pos = ["input", "weights"] if "weights" in arrs else ["input"]
            args = [arrs.pop(k) for k in pos]
            got = getattr(ndi, c["func"])(*args, **arrs, **c["kwargs"])
        exp = np.asarray(c["expected"])
        if not np.allclose(np.asarray(got, dtype=np.float64), exp.astype(np.float64), atol=10 ** -c["decimal"]):
            bad += 1
            print("MISMATCH vs SciPy:", c["func"], c["kwargs"], c["src"], got, exp)
    assert bad == 0, bad
    out = os.path.join(HERE, "kat_reference.json")
    with open(out, "w") as f:
        json.dump({"source": "literal vectors of mritools/cupyimg tests (see src fields); verified against SciPy",
                   "cases": CASES}, f, separators=(",", ":"))
    print(len(CASES), "cases ->", out, os.path.getsize(out) / 1e3, "kB")


if __name__ == "__main__":
    main()
