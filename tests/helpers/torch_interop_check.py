"""Run in a child process by tests/test_gpu_interop.py: torch is imported FIRST so that
libmi355img.so binds to the HIP runtime torch ships (one runtime per process is what
zero-copy interop needs).  Prints INTEROP_OK, or INTEROP_SKIP <reason>."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

try:
    import torch
except Exception as exc:          # pragma: no cover
    print("INTEROP_SKIP no torch:", exc)
    sys.exit(0)
if not torch.cuda.is_available():
    print("INTEROP_SKIP torch sees no device")
    sys.exit(0)
import cupyimg_amd as ca
from cupyimg_amd.scipy import ndimage as ndi

runtimes = sorted({line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line})
if len(runtimes) != 1:
    print("INTEROP_SKIP", len(runtimes), "HIP runtimes in the process:", runtimes)
    sys.exit(0)
if not ca.is_available():
    print("INTEROP_SKIP cupyimg_amd sees no device on torch's runtime", runtimes)
    sys.exit(0)

import scipy.ndimage as sndi

rng = np.random.default_rng(5)
x = rng.standard_normal((96, 128, 256)).astype(np.float32)
t = torch.from_numpy(x).cuda()
torch.cuda.synchronize()
for trial in range(5):
    big = torch.empty((256, 1024, 1024), device="cuda")        # 1 GiB of queued work in front of the producer
    for _ in range(4):
        big.normal_()
    prod = t * float(trial + 2) + 1.0                           # still queued behind `big` when it is imported
    a = ca.asarray(prod)                                        # zero copy; waits on the producer's stream
    assert a.ptr == prod.data_ptr()
    r = ndi.uniform_filter(a, size=3, mode="nearest")
    back = torch.as_tensor(r, device="cuda")                    # consumer waits on cai["stream"]
    got = (back * 1.0).cpu().numpy()
    want = sndi.uniform_filter(x * np.float32(trial + 2) + np.float32(1.0), 3, mode="nearest")
    err = np.abs(got - want).max() / np.abs(want).max()
    assert err <= 2e-6, (trial, err)
    del big

# in-place output into foreign memory, then hand the stream over explicitly
out = torch.zeros_like(t)
ndi.gaussian_filter(ca.asarray(t), 1.0, output=ca.asarray(out))
ca.core.stream_waits_for_us(torch.cuda.current_stream().cuda_stream or 1)
want = sndi.gaussian_filter(x, 1.0)
assert np.abs(out.cpu().numpy() - want).max() <= 2e-6 * np.abs(want).max()
print("INTEROP_OK", runtimes[0])
