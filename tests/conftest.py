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


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a device: on a box without one (no /dev/kfd) they are skipped instead of erroring in
    lm_engine_create, so that a plain `pytest tests` works everywhere.  LM_REQUIRE_GPU=1 (set it on a GPU box) turns the
    skip back into a hard failure -- there is no CPU fallback to hide behind."""
    if os.path.exists("/dev/kfd") or os.environ.get("LM_REQUIRE_GPU") == "1":
        return
    skip = pytest.mark.skip(reason="no AMD GPU on this machine (/dev/kfd missing)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


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
