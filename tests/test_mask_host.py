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


def test_a_kept_result_does_not_keep_the_engine_alive():
    """a result array used to reach the engine through its finalizer (result -> pool -> engine), so a caller who kept
    the masks of several `apply()` calls of the deprecated shims held one ~5 GB device workspace per mask.  The pool now holds the
    library and a weak reference: the engine goes away with the inferer, the kept result stays valid, and its page-locked block
    is freed through lm_host_free(NULL, p) when it is dropped."""
    import weakref

    from lungmask_amd import _native as nat
    from lungmask_amd.build import build_emu
    from lungmask_amd.mask import LMInferer
    from oracle import unet_oracle as uo

    eng = nat.Engine(0, nat.Library(build_emu(), allow_emulation=True))
    inf = LMInferer(state_dict=uo.synthetic_state_dict(3), engine=eng)
    res = inf._result_array((2, 8, 8))
    res[:] = 3
    eng_ref, pool = weakref.ref(eng), inf._pool
    del inf, eng
    gc.collect()
    assert eng_ref() is None, "the result array must not pin the engine"
    assert (res == 3).all()  # the block outlives the engine
    assert pool.closed  # LMInferer.__del__ closed the pool: the block will be freed, not parked, when the result goes
    del res
    gc.collect()
    assert pool.idle == []


def test_result_array_falls_back_to_pageable_memory(emu_engine, monkeypatch):
    """when page-locked memory runs out the idle blocks are released and `apply` gets an ordinary array
    (lm_apply_host accepts either) instead of raising."""
    from lungmask_amd import _native as nat
    from lungmask_amd.mask import LMInferer
    from oracle import unet_oracle as uo

    inf = LMInferer(state_dict=uo.synthetic_state_dict(3), engine=emu_engine)
    a = inf._result_array((2, 4, 4))
    del a
    gc.collect()
    assert len(inf._pool.idle) == 1

    def no_memory(n):
        raise nat.LMError("lm_host_alloc failed (-4): out of page-locked memory")

    monkeypatch.setattr(emu_engine, "host_alloc", no_memory)
    b = inf._result_array((3, 4, 4))  # another size: must allocate -> fails -> idle blocks freed -> pageable array
    assert b.shape == (3, 4, 4) and b.dtype == np.uint8 and inf._pool.idle == []
    inf.close()
