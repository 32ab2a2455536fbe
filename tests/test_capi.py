"""The C-ABI library: loads, and exports every symbol include/lungmask_hip.h declares.
No compute calls here (no GPU in the CPU suite)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "lungmask_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lm_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_symbols():
    syms = declared_symbols()
    assert "lm_engine_create" in syms and "lm_forward_dev" in syms and len(syms) >= 10


def test_hip_library_exports_all_declared_symbols():
    from lungmask_amd.build import build

    lib = ctypes.CDLL(build(verbose=False))
    for s in declared_symbols():
        assert hasattr(lib, s), f"liblungmask_hip.so does not export {s}"
    assert lib.lm_is_gpu_build() == 1


def test_product_path_refuses_emulation_library():
    from lungmask_amd import _native as nat
    from lungmask_amd.build import build_emu

    with pytest.raises(nat.LMError):
        nat.Library(build_emu())  # allow_emulation defaults to False
    with pytest.raises(nat.LMError):
        nat.Library("/nonexistent/liblungmask_hip.so")
