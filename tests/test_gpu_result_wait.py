"""The three ways a host-result call can wait (np_runtime_set_variant: 0 hipStreamSynchronize, 1 a stream-written flag,
2 watching the pinned result slots change — the default) return the same values, back to back and interleaved with
asynchronous work, for one-slot (sum / allclose / all) and two-slot (order statistics, weighted sums) results."""
import ctypes as C

import numpy as np
import pytest

from numpower_amd import synth

pytestmark = pytest.mark.gpu


def test_wait_modes_agree(hip):
    from numpower_amd import _lib
    lib = _lib.load()
    n = 300_007
    h = synth.uniform((n,), 11, -1.0, 1.0)
    w = synth.uniform((n,), 12, 0.0, 1.0)
    a, b, o = _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * n)
    _lib.check(lib.np_memcpy_h2d(a.ptr, h.ctypes.data, 4 * n))
    _lib.check(lib.np_memcpy_h2d(b.ptr, w.ctypes.data, 4 * n))
    want = None
    try:
        for variant in (0, 1, 2, 2, 1, 0):
            _lib.check(lib.np_runtime_set_variant(variant))
            got = []
            for rep in range(20):
                s = C.c_float(0.0)
                _lib.check(lib.np_unary(_lib.UNARY_OPS["exp"], a.ptr, o.ptr, n, 0.0, 0.0))     # asynchronous work in front
                _lib.check(lib.np_reduce_all(0, o.ptr, n, C.byref(s)))
                two = (C.c_float * 2)()
                _lib.check(lib.np_order_stat(a.ptr, n, (rep * 7919) % n, two))
                flag = C.c_int(-1)
                _lib.check(lib.np_count_mismatch(1, a.ptr, b.ptr, n, 1e-5, 1e-8, C.byref(flag)))
                saw, sw = C.c_float(0.0), C.c_float(0.0)
                _lib.check(lib.np_weighted_sums(a.ptr, b.ptr, n, C.byref(saw), C.byref(sw)))
                mn = C.c_float(0.0)
                _lib.check(lib.np_reduce_all(2, a.ptr, 1000 + rep, C.byref(mn)))              # tiny: one launch
                got.append((s.value, two[0], two[1], flag.value, saw.value, sw.value, mn.value))
            if want is None:
                want = got
                assert abs(got[0][0] - float(np.exp(h.astype(np.float64)).sum())) <= 1e-5 * float(np.exp(h.astype(np.float64)).sum())
                assert got[0][6] == h[:1000].min()
            assert got == want, variant
    finally:
        _lib.check(lib.np_runtime_set_variant(2))
        a.free(); b.free(); o.free()


def test_device_error_is_sticky_per_device_and_needs_an_acknowledgement():
    """A device-side wait that gives up (np_comm's bounded stream-ordering wait, a stream-K finisher whose peers never
    posted) raises the error word OF ITS DEVICE instead of letting NP_OK travel with a wrong result.  np_sync, np_memcpy_d2h
    and the host-result calls on that device return NP_ERR_DEVICE — EVERY time and to EVERY thread, until
    np_clear_device_error() acknowledges: the first reader no longer eats the error of whoever owns the failed launch
    (until round 6 the word was process-wide and cleared by its first reader)."""
    import ctypes as C
    import threading
    from numpower_amd._lib import load
    lib = load()
    assert lib.np_init(0) == 0
    assert lib.np_clear_device_error(None) == 0
    x = np.arange(1024, dtype=np.float32)
    dev = C.c_void_p()
    assert lib.np_malloc(C.byref(dev), x.nbytes) == 0
    assert lib.np_memcpy_h2d(dev, x.ctypes.data, x.nbytes) == 0
    out = C.c_float(0.0)
    back = np.empty_like(x)
    calls = ((b"np_comm", lambda: lib.np_sync()),
             (b"np_comm", lambda: lib.np_memcpy_d2h(back.ctypes.data, dev, x.nbytes)),
             (b"np_comm", lambda: lib.np_reduce_all(0, dev, 1024, C.byref(out))))
    for bits, what in ((1, b"np_comm"), (2, b"stream-K"), (3, b"stream-K")):
        assert lib.np_debug_raise_device_error(bits) == 0
        for _ in range(2):                                   # sticky: every sync point reports it, as often as it is asked
            for _, call in calls:
                assert call() != 0
                msg = lib.np_last_error()
                assert b"device-side wait gave up" in msg and what in msg and b"device 0" in msg, msg
        # another thread that synchronises first does not take the error away from this one
        seen = []
        t = threading.Thread(target=lambda: seen.append(lib.np_sync()))
        t.start()
        t.join()
        assert seen and seen[0] != 0
        assert lib.np_sync() != 0
        got = C.c_uint(0)
        assert lib.np_clear_device_error(C.byref(got)) == 0 and got.value == bits
        for _, call in calls:
            assert call() == 0, lib.np_last_error()
        assert lib.np_clear_device_error(C.byref(got)) == 0 and got.value == 0      # nothing left to acknowledge
    assert lib.np_sync() == 0
    assert (back == x).all() and out.value == float(x.sum())
    assert lib.np_free(dev) == 0


def test_the_binding_reports_a_device_error_as_its_exception_and_acknowledges_it(hip):
    """At the C ABI a device error is sticky until np_clear_device_error() (above).  A BINDING reports it the way it reports every
    failed call — an exception (zend_throw_error; `Error` in the stand-in) — and acknowledges it with that report
    (ext/np_ext_hooks.h np_ext_throw_last, dev_ok of the host library): one failed call per error, not a poisoned process."""
    from numpower_amd._lib import load
    from numpower_amd.ndarray import Error, NDArray
    lib = load()
    assert lib.np_clear_device_error(None) == 0
    x = np.arange(4096, dtype=np.float32)
    g = NDArray.array(x).gpu()
    assert NDArray.sum(g) == float(x.sum())
    assert lib.np_debug_raise_device_error(2) == 0
    with pytest.raises(Error, match="device-side wait gave up"):
        NDArray.sum(g)                                        # the report ...
    assert NDArray.sum(g) == float(x.sum())                   # ... was the acknowledgement: the next call is a normal call
    assert lib.np_sync() == 0
