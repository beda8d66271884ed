import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def pkg():
    return ge.load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    from dslam_amd.harness import synth as s
    return s


@pytest.fixture(scope="session")
def oracle(pkg):
    """CPU oracle bound through the generic ctypes wrapper (test infrastructure)."""
    o = ge.load_oracle()
    o.build()
    return o.open_oracle(pkg.CApi)


@pytest.fixture(scope="session")
def gpu(pkg):
    """HIP engine on device 0.  Fails loudly if the library is not built or no GPU is present."""
    assert os.path.exists(pkg.LIB_PATH), "libdslam_fusion.so not built: run python __graft_entry__.py"
    return pkg.open_engine(0)
