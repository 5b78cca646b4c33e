#!/usr/bin/env python3
"""BASELINE config A -- "uniform_filter size=5 on 128^3 float32 via scipy.ndimage CPU reference (plumbing, no GPU)".

Runs entirely without a GPU: the same report format as bench.py with the CPU restatement
(oracle/, `kind: "port"`) as the backend and scipy.ndimage -- the library the reference's own tests compare with --
as the comparator.  It proves the fixtures, the tolerance arithmetic and the report format before any GPU time is
spent; the product path is not involved (the oracle is test infrastructure and is never shipped).

    python scripts/bench_config_a.py [--steps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    import scipy
    import scipy.ndimage as sndi
    from oracle import ndimage as orc
    orc.build()
    n, size = 128, 5
    x = np.random.default_rng(0).standard_normal((n, n, n), dtype=np.float32)
    orc.uniform3d_f32(x, size, "reflect")                       # warm
    t0 = time.perf_counter()
    for _ in range(a.steps):
        got = orc.uniform3d_f32(x, size, "reflect")
    t_orc = (time.perf_counter() - t0) / a.steps
    t0 = time.perf_counter()
    for _ in range(a.steps):
        ref = sndi.uniform_filter(x, size=size)
    t_sp = (time.perf_counter() - t0) / a.steps
    ref64 = sndi.uniform_filter(x.astype(np.float64), size=size)
    err = float(np.abs(got.astype(np.float64) - ref64).max() / np.abs(ref64).max())
    line = {
        "metric": "Mvoxels/s, uniform_filter size=5 on 128^3 float32 (CPU plumbing config)", "value": round(x.size / t_orc / 1e6, 2),
        "unit": "Mvoxels/s", "n_gpus": 0, "steps": a.steps, "warmup": 1, "ms_per_step": round(t_orc * 1e3, 3),
        "higher_is_better": True, "scaling": "none", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "uniform_filter size=5 mode=reflect on 128x128x128 float32, host arrays",
                   "backend": "oracle/ndimage_oracle.c orc_uniform3d_f32 (CPU restatement, scalar, 1 thread)"},
        "roofline": None,
        "cpu_baseline": {"value": round(x.size / t_sp / 1e6, 2), "unit": "Mvoxels/s", "cores": 1, "kind": "reference",
                         "sample": "scipy.ndimage.uniform_filter {} (the reference's own comparator) on the same array".format(scipy.__version__)},
        "parity_vs_scipy_maxnorm_rel": err, "parity_tol": 1e-6, "parity_ok": err <= 1e-6,
    }
    print(json.dumps(line))
    return 0 if err <= 1e-6 else 1


if __name__ == "__main__":
    sys.exit(main())
