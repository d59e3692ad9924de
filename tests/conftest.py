import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """The native libraries and the oracle must exist; build them if this is a fresh checkout."""
    so = os.path.join(ROOT, "voxelraytracing_amd", "libvrt.so")
    so_h = os.path.join(ROOT, "voxelraytracing_amd", "libvrt_host.so")
    if not (os.path.exists(so) and os.path.exists(so_h)):
        import __graft_entry__
        __graft_entry__.build()
    from oracle import orc
    orc.build()


@pytest.fixture(scope="session")
def orc():
    from oracle import orc as o
    return o
