"""Diagnostic for the streaming-pass failure on launches of > 1024 waves: delta
kernels on an index-coded volume, so that every wrong sample names the voxel it
was taken from.   python scripts/diag_stream.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cupyimg_amd as ca
from cupyimg_amd import _lib

lib = _lib.load()
fp = ctypes.POINTER(ctypes.c_float)
lib.mi_debug_stream_pass.argtypes = [ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 4 + [fp] + [ctypes.c_int] * 3 + [
    fp, ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_void_p]
lib.mi_debug_set_stream_slice.argtypes = [ctypes.c_int]


def farr(v):
    a = (ctypes.c_float * len(v))(*v)
    return a


def run(x, axis, wav, wxv, slice_):
    nz, ny, nx = x.shape
    xd = ca.asarray(x)
    out = ca.empty(x.shape, np.float32)
    lib.mi_debug_set_stream_slice(slice_)
    wa, wx = len(wav), len(wxv)
    rc = lib.mi_debug_stream_pass(xd.ptr, out.ptr,
                                  nz, ny, nx, axis, farr(wav), wa, wa // 2, 0, farr(wxv), wx, 0, 0.0, None)
    assert rc == 0, (rc, _lib.last_error())
    ca.synchronize()
    return out.get()


def delta(n, k):
    w = [0.0] * n
    w[k] = 1.0
    return w


def main():
    shape = (60, 256, 1024)
    nz, ny, nx = shape
    x = np.arange(nz * ny * nx, dtype=np.float32).reshape(shape)   # exact below 2^24
    for wx, wa, axis in [(3, 1, 1), (3, 3, 0), (9, 9, 0), (17, 17, 0), (17, 1, 1)]:
        for kx in sorted({0, wx // 2, wx - 1}):
            wav = delta(wa, wa // 2)
            wxv = delta(wx, kx)
            good = run(x, axis, wav, wxv, 1000)
            bad = run(x, axis, wav, wxv, 0)
            # expectation: in[x + kx - wx//2] with reflect
            idx = np.arange(nx) + kx - wx // 2
            idx = np.where(idx < 0, -1 - idx, idx)
            idx = np.where(idx >= nx, 2 * nx - 1 - idx, idx)
            want = x[:, :, idx]
            print("wx %d wa %d axis %d tap %d: sliced==want %s, unsliced mismatches %d" % (
                wx, wa, axis, kx, np.array_equal(good, want), int((bad != want).sum())), flush=True)
            m = np.argwhere(bad != want)
            if len(m):
                lanes = sorted(set(((m[:, 2] % 256) // 4).tolist()))
                comps = sorted(set((m[:, 2] % 4).tolist()))
                print("   lanes", lanes[:64], "comps", comps, "z range", m[:, 0].min(), m[:, 0].max(), "y range", m[:, 1].min(), m[:, 1].max())
                for (z, y, xx) in m[:12]:
                    g = bad[z, y, xx]
                    w_ = want[z, y, xx]
                    gi = int(g)
                    gz, gy, gx = gi // (ny * nx), (gi // nx) % ny, gi % nx
                    wi = int(w_)
                    print("   at (z %d y %d x %d lane %d comp %d): want %d = (%d,%d,%d)  got %r = (%d,%d,%d)" % (
                        z, y, xx, (xx % 256) // 4, xx % 4, wi, wi // (ny * nx), (wi // nx) % ny, wi % nx, g, gz, gy, gx))
                # histogram of source displacement in x
                g = bad[tuple(m.T)].astype(np.int64)
                w_ = want[tuple(m.T)].astype(np.int64)
                d = g - w_
                vals, cnt = np.unique(d, return_counts=True)
                order = np.argsort(-cnt)[:10]
                print("   got-want (linear index) histogram:", [(int(vals[i]), int(cnt[i])) for i in order])


main()
