import os
import sys

import pytest

# Tests that run several ranks as THREADS of this process give every rank its
# own streams; the HIP runtime multiplexes a process's streams onto a few
# hardware queues (four by default), where a kernel that polls for a neighbour
# can sit in front of the neighbour's kernel.  More queues, before HIP starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def _have_gpu():
    import spmv_amd
    from spmv_amd import hip
    try:
        return hip.device_count() > 0
    except Exception:
        return False


@pytest.fixture(scope="session")
def ctx():
    """One spmv_hip context on GPU 0 for the whole session.  GPU tests do not
    skip when the device or the HIP library is missing: they fail."""
    from spmv_amd import hip
    c = hip.Context(0)
    yield c
    c.synchronize()
    c.close()
