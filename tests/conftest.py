import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as ge  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, as __graft_entry__.build() does.
    hipcc cross-compiles without a GPU; on the GPU box the prebuilt files travel with the snapshot and nothing is built."""
    pkg = ge.load_package()
    shim = os.path.join(ROOT, ge.PKG_DIRNAME, "itmlib", "tests")
    needed = [pkg.LIB_PATH, os.path.join(shim, "driver_harness"), os.path.join(shim, "reintegrate_rccl"),
              os.path.join(ROOT, "oracle", "liboracle.so")]
    if not all(os.path.exists(p) for p in needed):
        ge.build()


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
