import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU emulation test")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLD


@pytest.fixture(scope="session")
def emu_engine():
    """TEST-ONLY engine on the g++ emulation of the kernel sources (tests/emu)."""
    from lungmask_amd import _native as nat
    from lungmask_amd.build import build_emu

    lib = nat.Library(build_emu(), allow_emulation=True)
    eng = nat.Engine(0, lib)
    yield eng
    eng.close()


@pytest.fixture(scope="session")
def gpu_engine():
    """The product engine: liblungmask_hip.so on cuda:0.  Fails loudly if the HIP
    library is missing -- there is no fallback."""
    from lungmask_amd import _native as nat

    eng = nat.Engine(0)
    assert eng.L.is_gpu
    yield eng
    eng.close()
