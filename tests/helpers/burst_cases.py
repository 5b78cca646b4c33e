"""Burst cases for every kernel that counts its vector-memory operations by hand (`s_waitcnt vmcnt(n)`, n > 0: MI_VMCNT in
csrc/common.hpp) -- shared by tests/test_gpu_burst.py (last launch of a burst against an independent kernel path) and by the
STRICT-BUILD comparison: run as a program,

    python tests/helpers/burst_cases.py --out digests.json [--only name,name] [--burst 40] [--rounds 2]

it launches every case BURST times back to back, ROUNDS times over, and writes the SHA-256 of the last output of every round.
The test runs it twice -- with the product library and with libmi355img_strict.so (MI355IMG_LIB; every counted wait = vmcnt(0))
-- and requires identical digests: a wait that is an operation short shows as a difference between the two builds under load,
whichever kernel it sits in.  Test infrastructure; nothing in the package imports it."""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _rot(axis_pair, deg, n, step=1.0, shift=(0.5, -1.25, 2.0)):
    """rotation by `deg` in the plane of `axis_pair` about the centre of an n^3 volume, `step` along the remaining axis"""
    a = np.deg2rad(deg)
    c, s = np.cos(a), np.sin(a)
    M = np.eye(3)
    i, j = axis_pair
    M[i, i], M[i, j], M[j, i], M[j, j] = c, -s, s, c
    k = 3 - i - j
    M[k, k] = step
    ctr = (n - 1) / 2.0
    return M, ctr - M @ np.array([ctr] * 3) + np.array(shift)


class Cases:
    """name -> (setup() -> (fn(out), shape, dtype, expect_kernel_fragment)); inputs are seeded, so both builds see the same data"""

    def __init__(self, gpu, ndi, lib, n_cubic=512, n_interp=256):
        self.gpu, self.ndi, self.lib = gpu, ndi, lib
        self.n_cubic, self.n_interp = n_cubic, n_interp
        self._vols = {}

    def vol(self, shape, seed):
        key = (tuple(shape), seed)
        if key not in self._vols:
            self._vols.clear()                         # one resident input at a time
            x = np.random.default_rng(seed).standard_normal(shape, dtype=np.float32)
            self._vols[key] = (x, self.gpu.asarray(x))
        return self._vols[key]

    def names(self):
        return [n for n in dir(self) if n.startswith("case_")]

    # ---- separable / min-max (192 x 256 x 512)
    def _sep(self, size, mode, cval=0.0):
        x, xd = self.vol((192, 256, 512), 1)
        return (lambda o: self.ndi.uniform_filter(xd, size, mode=mode, cval=cval, output=o)), x.shape, np.float32, "sep3d_long"

    def case_long3_5(self):
        return self._sep(5, "reflect")

    def case_long3_9(self):
        return self._sep(9, "mirror")

    def case_long3_17(self):
        return self._sep(17, "nearest")

    def case_long_const_13(self):
        return self._sep(13, "constant", 0.25)               # a fill value: the r2 stream with its coverage correction

    def case_long3_const0_13(self):
        return self._sep(13, "constant")                     # r5: zero fill on the r3 kernel

    # r6: the ragged build (rows of any length, three stores per step: its own wait counts), one and two x tiles
    def case_long3_ragged_17(self):
        x, xd = self.vol((181, 217, 181), 5)
        return (lambda o: self.ndi.gaussian_filter(xd, 2.0, mode="mirror", output=o)), x.shape, np.float32, "ragged"

    def case_long3_ragged_13_two_tiles(self):
        x, xd = self.vol((96, 130, 301), 6)
        return (lambda o: self.ndi.uniform_filter(xd, 13, mode="wrap", output=o)), x.shape, np.float32, "ragged"

    def case_mm3f32_ragged_9(self):
        x, xd = self.vol((181, 217, 181), 5)
        return (lambda o: self.ndi.grey_erosion(xd, size=9, mode="mirror", output=o)), x.shape, np.float32, "mm3f32_long_kernel<9,min,ragged>"

    def case_mm3f32_5(self):
        x, xd = self.vol((192, 256, 512), 1)
        return (lambda o: self.ndi.maximum_filter(xd, 5, output=o)), x.shape, np.float32, "mm3f32_long_kernel"

    # ---- order-1 interpolation (n_interp^3)
    def _affine1(self, plane, deg, step, kern):
        n = self.n_interp
        x, xd = self.vol((n, n, n), 3)
        M, off = _rot(plane, deg, n, step)
        return (lambda o: self.ndi.affine_transform(xd, M, off, order=1, mode="constant", cval=0.25, output=o)), x.shape, np.float32, kern

    def case_affine_zrect_7(self):
        return self._affine1((1, 2), 7.0, 1.02, "affine3d_zrect_kernel")

    def case_affine_zstream_50(self):
        return self._affine1((1, 2), 50.0, 1.02, "affine3d_zstream_kernel<32,0,true>")

    def case_affine_zstream_sax1_50(self):
        return self._affine1((0, 2), 50.0, 1.0, "affine3d_zstream_kernel<32,1,true>")

    def case_map_zstream(self):
        n = self.n_interp
        x, xd = self.vol((n, n, n), 3)
        M, off = _rot((1, 2), 7.0, n, 1.02)
        idx = np.indices((n, n, n), dtype=np.float32).reshape(3, -1)
        coords = (M.astype(np.float32) @ idx + off.astype(np.float32)[:, None]).reshape(3, n, n, n)
        cd = self.gpu.asarray(coords)
        return (lambda o: self.ndi.map_coordinates(xd, cd, order=1, mode="constant", cval=0.25, output=o)), x.shape, np.float32, "map_coords3d_zstream_kernel"

    # ---- order 3 on float32 coefficients (n_cubic^3; prefilter=False: the interpolation kernel alone).  r4b = True: the
    # 64-tap kernel of round 4 (cubic3_zstream_kernel, bit-identical to the gather kernel) instead of the r5 default
    # (cubic3_zfactor_kernel: same staging, ring and waits, in-plane values once per input plane)
    def _cubic(self, plane, deg, step, kern, mode="constant", r4b=False):
        n = self.n_cubic
        x, xd = self.vol((n, n, n), 5)
        M, off = _rot(plane, deg, n, step)
        if r4b:
            kern = kern.replace("cubic3_zfactor_kernel", "cubic3_zstream_kernel")

        def fn(o):
            self.lib.mi_debug_set_cubic_zfactor(0 if r4b else 1)
            try:
                self.ndi.affine_transform(xd, M, off, order=3, mode=mode, cval=0.25, prefilter=False, output=o)
            finally:
                self.lib.mi_debug_set_cubic_zfactor(1)
        return fn, x.shape, np.float32, kern

    def case_cubic_zstream0_7(self):
        return self._cubic((1, 2), 7.0, 1.0, "cubic3_zfactor_kernel<0>")

    def case_cubic_zstream0_7_step102(self):
        return self._cubic((1, 2), 7.0, 1.02, "cubic3_zfactor_kernel<0>")                # the BASELINE matrix: every 50th step fetches a plane late

    def case_cubic_zstream0_30(self):
        return self._cubic((1, 2), 30.0, 1.0, "cubic3_zfactor_kernel<0>")

    def case_cubic_zstream0_80_mirror(self):
        return self._cubic((1, 2), 80.0, 0.97, "cubic3_zfactor_kernel<0>", "mirror")

    def case_cubic_zstream1_7(self):
        return self._cubic((0, 2), 7.0, 1.0, "cubic3_zfactor_kernel<1>")

    def case_cubic_zstream1_30(self):
        return self._cubic((0, 2), 30.0, 1.02, "cubic3_zfactor_kernel<1>")

    def case_cubic_zstream1_80(self):
        return self._cubic((0, 2), 80.0, 1.0, "cubic3_zfactor_kernel<1>")

    def case_cubic_r4b_zstream0_7_step102(self):
        return self._cubic((1, 2), 7.0, 1.02, "cubic3_zfactor_kernel<0>", r4b=True)

    def case_cubic_r4b_zstream0_30(self):
        return self._cubic((1, 2), 30.0, 1.0, "cubic3_zfactor_kernel<0>", r4b=True)

    def case_cubic_r4b_zstream1_80(self):
        return self._cubic((0, 2), 80.0, 1.0, "cubic3_zfactor_kernel<1>", r4b=True)

    def case_cubic_rowblend_7(self):
        return self._cubic((0, 1), 7.0, 1.0, "cubic3_rowblend_kernel")


def digest(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--only", default="")
    ap.add_argument("--burst", type=int, default=40)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--n-cubic", type=int, default=512)
    a = ap.parse_args()
    import cupyimg_amd as gpu
    from cupyimg_amd import _lib, last_kernel
    from cupyimg_amd.scipy import ndimage as ndi
    lib = _lib.load()
    cases = Cases(gpu, ndi, lib, n_cubic=a.n_cubic)
    only = [s for s in a.only.split(",") if s]
    res = {"library": _lib.library_path(), "burst": a.burst, "cases": {}}
    for name in cases.names():
        if only and name not in only and name[5:] not in only:
            continue
        fn, shape, dtype, kern = getattr(cases, name)()
        out = gpu.empty(shape, dtype)
        ds = []
        for _ in range(a.rounds):
            for _ in range(a.burst):
                fn(out)
            ds.append(digest(out.get()))
        res["cases"][name] = {"kernel": last_kernel(), "expected": kern, "digests": ds}
        del out, fn
    with open(a.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
