"""Config D (map_coordinates order 1, 512^3): the z-streaming kernel against the L1-gather kernel, z chunkings, phase
ablations (mi_debug_set_affine_dbg: 1 no DMA, 4 no stores, 8 no interpolation), whole-volume parity -> profiles/r4_map_zstream.txt"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "scripts"))
import cupyimg_amd as ca
from cupyimg_amd import _lib, last_kernel
from cupyimg_amd.scipy import ndimage as ndi
from helpers import fullsize as fs
from bench_configs import timeit
lib = _lib.load()
x = fs.volume_f32((512,) * 3); xd = ca.asarray(x); out = ca.empty(x.shape, np.float32)
coords = fs.affine_coords_f32(512); cd = ca.asarray(coords)
ref = None
for zs, zc in [(0, 0), (1, 0), (1, 2), (1, 8), (2, 0), (0, 0), (1, 0)]:
    lib.mi_debug_set_map_zstream(zs); lib.mi_debug_set_map_zchunks(zc)
    s, f = timeit(lambda: ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out), 20)
    got = out.get()
    if ref is None: ref = got
    print(json.dumps({"zstream": zs, "zchunks": zc, "us": round(s * 1e6, 1), "frac": round(20 * 512**3 / s / 8e12, 4), "identical_to_gather_kernel": bool(np.array_equal(got, ref)), "kernel": last_kernel()[:60]}), flush=True)
for dbg in (1, 4, 8, 5, 9, 12, 13, 0):
    lib.mi_debug_set_map_zstream(1); lib.mi_debug_set_map_zchunks(0); lib.mi_debug_set_affine_dbg(dbg)
    s, f = timeit(lambda: ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out), 20)
    print("ablation dbg", dbg, round(s * 1e6, 1), "us", flush=True)
lib.mi_debug_set_map_zstream(1); lib.mi_debug_set_map_zchunks(0)
ndi.map_coordinates(xd, cd, order=1, mode="constant", output=out)
print("parity whole volume:", fs.whole_volume_map_coordinates(x, coords, out.get()))
