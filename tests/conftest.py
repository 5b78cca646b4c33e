import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def _has_gpu():
    try:
        import cupyimg_amd
        return cupyimg_amd.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    """Skips when no device is visible; GPU tests also carry @pytest.mark.gpu."""
    if not _has_gpu():
        pytest.skip("no MI355X visible")
    import cupyimg_amd
    return cupyimg_amd
