import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def lib_built():
    """libvti.so must exist (build() makes it); the CPU suite only loads it, never computes with it."""
    import vti_amd
    if not os.path.exists(vti_amd.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return vti_amd
