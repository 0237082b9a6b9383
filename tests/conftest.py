import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ctx():
    """One liblpx context for the whole GPU session (goes through the C-ABI)."""
    from lidar_processing_amd import Context
    c = Context(0)
    c.reserve(200_000)
    yield c
    c.close()
