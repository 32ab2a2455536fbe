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


class _FakePipeEngine:
    """Stands in for _native.Engine under `LMInferer.apply_async`'s two threads: records the lm_pipe_* calls, checks the caller's part
    of the protocol (include/lungmask_hip.h: an input buffer is only refilled after the hot path of the volume that used it has
    returned; a volume's hot path runs behind its own copy-in; downloads follow their hot path) and "labels" a volume with a
    function of its contents."""

    class _Lib:
        lm_pipe_upload = True  # (apply_async looks for the entry point)

    class _L:
        pass

    def __init__(self, fail_on=None):
        import threading

        self.L = self._L()
        self.L.lib = self._Lib()
        self.lock = threading.Lock()
        self.log = []
        self.buf = [None, None]
        self.busy = [False, False]   # buffer k is between its upload and the return of its hot path
        self.out = [None, None]
        self.fail_on = fail_on
        self.n_apply = 0

    def host_alloc(self, n):
        raise RuntimeError("no page-locked memory in the fake")  # -> pageable result arrays

    def pipe_upload(self, k, vol):
        import time

        if vol is None:
            return
        with self.lock:
            assert not self.busy[k], "input buffer refilled while its volume is still on the hot path"
            self.busy[k] = True
            self.log.append(("upload", k))
        time.sleep(0.002)
        self.buf[k] = vol.copy()

    def pipe_apply(self, k, slot, shape, dtype, fill_slot=-1, batch_size=20, volume_postprocessing=True):
        import time

        with self.lock:
            assert self.busy[k] and self.buf[k] is not None and tuple(shape) == self.buf[k].shape, "hot path before its copy-in"
            self.log.append(("apply", k))
            self.n_apply += 1
            n = self.n_apply
        time.sleep(0.005)
        if self.fail_on == n:
            with self.lock:
                self.busy[k] = False
            raise RuntimeError("device error in volume %d" % n)
        self.out[k] = (self.buf[k] % 7).astype(np.uint8)
        with self.lock:
            self.busy[k] = False

    def pipe_download(self, k, out):
        with self.lock:
            self.log.append(("download", k))
        out[...] = self.out[k]

    def pipe_wait(self, k):
        with self.lock:
            self.log.append(("wait", k))


def _fake_inferer(engine):
    from lungmask_amd.mask import LMInferer, _ResultPool

    inf = LMInferer.__new__(LMInferer)
    inf.engine, inf._engines, inf._shard, inf._async, inf._own_engine = engine, [engine], None, None, False
    inf.fill_slot, inf.batch_size, inf.volume_postprocessing, inf.fillmodel, inf.modelname = -1, 20, True, None, "R231"
    inf.reuse_output, inf._out = False, None
    inf._pool = _ResultPool.__new__(_ResultPool)
    import threading

    inf._pool.L, inf._pool._engine, inf._pool.lock, inf._pool.idle, inf._pool.closed = engine.L, (lambda: None), threading.Lock(), [], False
    return inf


def test_apply_async_keeps_two_volumes_in_flight_in_order():
    """`LMInferer.apply_async`: results in submission order and equal to what each volume alone gives; the uploader
    runs ahead of the hot path by at most one volume; the protocol of lm_pipe_* holds under the two host threads."""
    eng = _FakePipeEngine()
    inf = _fake_inferer(eng)
    rng = np.random.default_rng(4)
    vols = [rng.integers(-1000, 400, size=(3 + i % 2, 6, 5)).astype(np.int16) for i in range(7)]
    pend, got = [], []
    for v in vols:
        pend.append(inf.apply_async(v))
        if len(pend) > 1:
            got.append(pend.pop(0).result())
    while pend:
        got.append(pend.pop(0).result())
    for v, r in zip(vols, got):
        assert r.dtype == np.uint8 and np.array_equal(r, (v % 7).astype(np.uint8))
    ups = [i for i, (what, _) in enumerate(eng.log) if what == "upload"]
    aps = [i for i, (what, _) in enumerate(eng.log) if what == "apply"]
    assert len(ups) == len(aps) == 7 and all(u < a for u, a in zip(ups, aps))
    assert all(ups[i + 2] > aps[i] for i in range(5))                # buffer k refilled only after its volume left the hot path
    assert [k for what, k in eng.log if what == "apply"] == [i % 2 for i in range(7)]
    # other dtypes are widened like apply() does; an unsupported one raises from result()
    assert np.array_equal(inf.apply_async(vols[0].astype(np.uint8).astype(np.int8)).result(), (vols[0].astype(np.uint8).astype(np.int8).astype(np.int32) % 7).astype(np.uint8))
    with pytest.raises(TypeError):
        inf.apply_async(vols[0].astype(np.complex64)).result()
    inf._async.close()


def test_apply_async_propagates_a_failing_volume_and_carries_on():
    eng = _FakePipeEngine(fail_on=2)
    inf = _fake_inferer(eng)
    vols = [np.full((2, 4, 4), i, np.int16) for i in range(4)]
    hs = [inf.apply_async(v) for v in vols]
    assert np.array_equal(hs[0].result(), (vols[0] % 7).astype(np.uint8))
    with pytest.raises(RuntimeError, match="device error in volume 2"):
        hs[1].result()
    assert np.array_equal(hs[2].result(), (vols[2] % 7).astype(np.uint8)) and np.array_equal(hs[3].result(), (vols[3] % 7).astype(np.uint8))
    inf._async.flush()
    inf._async.close()


def test_apply_async_threads_do_not_keep_the_inferer_alive():
    """The queue's two host threads hold a weak reference to the inferer: an inferer that is dropped without close() is collected, and
    its finalizer ends the threads (they would otherwise pin it, and through it the engine, for the life of the process)."""
    import threading
    import time
    import weakref

    eng = _FakePipeEngine()
    inf = _fake_inferer(eng)
    assert np.array_equal(inf.apply_async(np.full((2, 4, 4), 5, np.int16)).result(), np.full((2, 4, 4), 5, np.uint8))
    pipe = inf._async
    threads = list(pipe.threads)
    assert all(t.is_alive() for t in threads)
    ref = weakref.ref(inf)
    del inf, pipe
    gc.collect()
    assert ref() is None, "the queue's threads must not pin the inferer"
    for t in threads:
        t.join(timeout=5)
    assert not any(t.is_alive() for t in threads)
