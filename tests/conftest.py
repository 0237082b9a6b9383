import os
import sys

import pytest

# the launcher's job, not the library's (liblpx never touches the environment): HIP multiplexes streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and reads the variable on its first call
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", params=["lists", "search"])
def ctx(request):
    """One liblpx context per neighbour mode for the whole GPU session (goes through the C-ABI): every test that
    takes `ctx` runs once with the radius lists materialised (the default of a single-frame context) and once
    with the expansion-driven search (the default of a batch context)."""
    from lidar_processing_amd import Context
    c = Context(0)
    c.set_neighbour_mode(request.param)
    c.reserve(200_000)
    yield c
    c.close()
