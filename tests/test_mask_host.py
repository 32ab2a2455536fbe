"""CPU suite: host-side logic of the mirrored operator interface (lungmask_amd/mask.py) that needs no device."""
import gc

import numpy as np
import pytest


def test_result_arrays_are_the_callers_alone(emu_engine):
    """mask.py:210 -- every apply() returns an array of its own.  LMInferer hands out root arrays over page-locked blocks of its pool
    (lm_host_alloc); a block goes back to the pool from a weakref.finalize on the root, i.e. only once the result AND every view /
    slice taken from it are gone -- no interpreter reference count is inspected (the getrefcount constant)."""
    from lungmask_amd.mask import LMInferer
    from oracle import unet_oracle as uo

    inf = LMInferer(state_dict=uo.synthetic_state_dict(3), engine=emu_engine)
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
    assert len({x.ctypes.data for x in (b, c, e, f)}) == 4  # live results never share
    del b, c, d, e, f
    gc.collect()
    assert len(inf._pool.idle) <= 2  # two idle blocks kept at most, the others were freed


def test_force_cpu_is_an_error_without_the_opt_in(monkeypatch):
    """mask.py:118-134 selects a device; this engine has no CPU to select (checked before any engine is created)."""
    from lungmask_amd.mask import LMInferer

    monkeypatch.delenv("LUNGMASK_AMD_ALLOW_CPU_FLAG", raising=False)
    with pytest.raises(RuntimeError, match="MI355X-only"):
        LMInferer(force_cpu=True)
    with pytest.raises(AssertionError):
        LMInferer(modelname="no-such-model")  # mask.py:95-97
