"""More than one RCCL rank: tests/helpers/multirank_check.py on 2 and 4 real GPUs (one process per GPU, launched by
torch.distributed.run before anything in the children has touched a device).  Skipped on boxes with fewer GPUs -- the
single-rank leg below runs the same checks without neighbours everywhere, so the helper is exercised on every box and
the first multi-GPU box proves the xGMI path without a human."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HELPER = os.path.join(ROOT, "tests", "helpers", "multirank_check.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _env():
    """The children get NO private environment: HSA_ENABLE_IPC_MODE_LEGACY (dmabuf IPC, which RCCL across processes needs
    on this driver) is REMOVED here, so what the ranks run with is the default cupyimg_amd._lib.load() sets before the
    first HIP call -- the same mechanism bench.py and any other program using the library gets."""
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    return env


def test_helper_single_rank(gpu):
    out = subprocess.run([sys.executable, HELPER], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600,
                         env=_env())
    assert out.returncode == 0 and "MULTIRANK OK 1" in out.stdout, out.stdout[-3000:]
    assert "FAIL" not in out.stdout


@pytest.mark.parametrize("world", [2, 4, 8])
def test_real_ranks_over_rccl(gpu, world):
    if gpu.device_count() < world:
        pytest.skip("needs {} GPUs, this box has {}".format(world, gpu.device_count()))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node={}".format(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), HELPER]
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200, env=_env())
    assert out.returncode == 0 and "MULTIRANK OK {}".format(world) in out.stdout, out.stdout[-4000:]
    assert "FAIL" not in out.stdout
