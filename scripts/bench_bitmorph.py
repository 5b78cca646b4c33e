"""Binary erosion / dilation on bool volumes: the bit-packed fused kernel (csrc/bitmorph3d.hip) against the byte kernel
(binary3d.hip), tile sweeps.   python scripts/bench_bitmorph.py [--sweep]
Fractions price the ALGORITHMIC 2 B/voxel (read 1 + write 1, once per CALL whatever the iteration count) at 8 TB/s."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cupyimg_amd as ca
from cupyimg_amd import _lib
from cupyimg_amd.scipy import ndimage as ndi

knob = _lib.load().mi_debug_set_bitmorph
knob.argtypes = [ctypes.c_int] * 3


def timeit(fn, min_ms=30.0):
    for _ in range(3): fn()
    ca.synchronize()
    e0, e1 = ca.Event(), ca.Event()
    e0.record(); fn(); fn(); e1.record(); ca.synchronize()
    reps = max(5, int(min_ms / max(e0.elapsed_ms(e1) / 2, 1e-3)))
    for _ in range(reps): fn()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); ca.synchronize()
    return e0.elapsed_ms(e1) / reps * 1e3


def ball(r):
    z, y, x = np.mgrid[-r:r + 1, -r:r + 1, -r:r + 1]
    return (x * x + y * y + z * z) <= r * r


def cases(b, bo, m):
    full = np.ones((3, 3, 3), bool)
    return [("erosion cross", lambda: ndi.binary_erosion(b, output=bo)),
            ("erosion 3^3", lambda: ndi.binary_erosion(b, structure=full, output=bo)),
            ("dilation ball2", lambda: ndi.binary_dilation(b, structure=ball(2), output=bo)),
            ("erosion cross x2", lambda: ndi.binary_erosion(b, iterations=2, output=bo)),
            ("erosion cross x3", lambda: ndi.binary_erosion(b, iterations=3, output=bo)),
            ("erosion cross x4", lambda: ndi.binary_erosion(b, iterations=4, output=bo)),
            ("erosion cross x8", lambda: ndi.binary_erosion(b, iterations=8, output=bo)),
            ("erosion cross masked", lambda: ndi.binary_erosion(b, mask=m, output=bo)),
            ("opening cross", lambda: ndi.binary_opening(b, output=bo)),
            ("opening cross x3", lambda: ndi.binary_opening(b, iterations=3, output=bo))]


def main():
    sweep = "--sweep" in sys.argv
    for shape in [(512, 512, 512), (1024, 1024, 1024), (256, 256, 256), (176, 256, 256)]:
        rng = np.random.default_rng(0)
        b = ca.asarray(rng.random(shape) > 0.3); bo = ca.empty(shape, bool); m = ca.asarray(rng.random(shape) > 0.3)
        n = float(np.prod(shape))
        print("shape", shape, flush=True)
        for name, fn in cases(b, bo, m):
            knob(0, 0, 0); t0 = timeit(fn)
            knob(1, 0, 0); t1 = timeit(fn)
            print("   %-22s byte kernel %8.1f us (%.3f)   bit kernel %8.1f us (%.3f of 8 TB/s)   %s" % (
                name, t0, 2 * n / t0 / 1e6 / 8, t1, 2 * n / t1 / 1e6 / 8, ca.last_kernel()[24:80]), flush=True)
        if sweep and shape[0] >= 512:
            for name, fn in cases(b, bo, m)[:1] + cases(b, bo, m)[4:5]:
                for nt in (256,):
                    res = []
                    for ty in (8, 16, 24, 30, 32, 40, 48, 62, 64, 80, 100, 126):
                        for nzc in (2, 4, 8, 12, 16, 24, 32, 48, 64):
                            knob(1, ty, nzc)
                            try:
                                t = timeit(fn, 6.0)
                                if ("tile=%dx" % ty) in ca.last_kernel():
                                    res.append((t, ty, nzc))
                            except Exception as e:
                                pass
                    res.sort()
                    print("   sweep nt=%4d %-18s best (us, ty, nzc): %s" % (nt, name, ["%.1f/%d/%d" % r for r in res[:8]]), flush=True)
            knob(1, 0, 0)
        b = bo = m = None
        ca.free_all_blocks()


if __name__ == "__main__":
    main()
