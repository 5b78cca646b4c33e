"""HIP path vs the reference's literal known-answer vectors (GPU)."""
import pytest

from test_oracle_kat import run_kat

pytestmark = pytest.mark.gpu


def test_hip_reproduces_reference_known_answers(gpu):
    from cupyimg_amd.scipy import ndimage as ndi
    assert run_kat(ndi, to_device=gpu.asarray) > 1500
