"""CPU suite: host-side logic of the mirrored operator interface (lungmask_amd/mask.py) that needs no device."""
import gc

import numpy as np
import pytest


def test_result_arrays_are_the_callers_alone():
    """mask.py:210 -- every apply() returns an array of its own.  LMInferer carves the results out of blocks it recycles, and a
    block may only be used again once the previous result AND every view / slice taken from it are gone."""
    from lungmask_amd.mask import LMInferer

    inf = LMInferer.__new__(LMInferer)  # no engine: only the result-memory bookkeeping is exercised
    inf._blocks = []
    a = inf._result_array((3, 4, 5))
    a[:] = 7
    addr = a.ctypes.data
    view = a[1]
    del a
    gc.collect()
    b = inf._result_array((3, 4, 5))
    assert b.ctypes.data != addr and not np.shares_memory(b, view)  # a slice of the first result is still alive
    b[:] = 9
    assert (view == 7).all()
    del view
    gc.collect()
    c = inf._result_array((3, 4, 5))
    assert c.ctypes.data == addr            # everything dropped: the block is handed out again
    d = inf._result_array((2, 4, 5))        # another size never aliases
    assert not np.shares_memory(c, d) and not np.shares_memory(b, d)
    e, f = inf._result_array((3, 4, 5)), inf._result_array((3, 4, 5))
    assert len({x.ctypes.data for x in (b, c, e, f)}) == 4 and len(inf._blocks) <= 2  # live results never share; two idle blocks kept at most


def test_force_cpu_is_an_error_without_the_opt_in(monkeypatch):
    """mask.py:118-134 selects a device; this engine has no CPU to select (checked before any engine is created)."""
    from lungmask_amd.mask import LMInferer

    monkeypatch.delenv("LUNGMASK_AMD_ALLOW_CPU_FLAG", raising=False)
    with pytest.raises(RuntimeError, match="MI355X-only"):
        LMInferer(force_cpu=True)
    with pytest.raises(AssertionError):
        LMInferer(modelname="no-such-model")  # mask.py:95-97
