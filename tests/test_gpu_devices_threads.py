"""Two things a one-GPU, one-thread test run cannot see by accident:

  * NDArray::setDevice semantics (numpower.c:615-635 -> cudaSetDevice): switching the current device must leave the
    arrays of the device used before alive and usable.  np_set_device used to trim the pool and destroy the old stream
    (VERDICT r02 missing #4); it now keeps per-device state.  The switching test needs two devices and SKIPS on the
    usual one-GPU box — it is written so that it runs the day a multi-GPU box does.  What CAN run on one GPU — that
    re-selecting the same device is free and keeps everything, and that out-of-range devices are refused — runs always.
  * host-result entry points from several threads at once: np_reduce_all / np_all /
    np_count_mismatch / np_moments / np_order_stat share one set of pinned result slots; ctypes releases the GIL, so
    Python threads really do enter the library concurrently.  Every thread must read its OWN result.
"""
import ctypes as C
import threading

import numpy as np
import pytest

from numpower_amd import synth
from numpower_amd._lib import NumPowerError, check, load

pytestmark = pytest.mark.gpu


def _device_count():
    n = C.c_int(0)
    check(load().np_device_count(C.byref(n)))
    return n.value


def test_reselecting_the_device_keeps_arrays_pool_and_stream(hip):
    lib = load()
    x = synth.uniform((1 << 16,), 71, -1.0, 1.0)
    d = hip.DeviceArray.from_host(x)
    stream, live, reserved = lib.np_get_stream(), lib.np_live_allocs(), lib.np_pool_reserved_bytes()
    check(lib.np_set_device(0))
    assert lib.np_get_stream() == stream and lib.np_live_allocs() == live and lib.np_pool_reserved_bytes() == reserved
    assert (d.to_host() == x).all()
    with pytest.raises(NumPowerError, match="out of range"):
        check(lib.np_set_device(_device_count()))
    with pytest.raises(NumPowerError, match="out of range"):
        check(lib.np_set_device(-1))
    assert (hip.unary("negate", d).to_host() == -x).all()      # the failed switches changed nothing
    d.free()


def test_set_device_switches_with_live_arrays(hip):
    if _device_count() < 2:
        pytest.skip("needs two GPUs: np_set_device(1) with arrays of device 0 alive")
    lib = load()
    x = synth.uniform((1 << 20,), 72, -1.0, 1.0)
    y = synth.uniform((1 << 20,), 73, -1.0, 1.0)
    try:
        check(lib.np_set_device(0))
        d0 = hip.DeviceArray.from_host(x)
        e0 = hip.unary("exp", d0)                    # work in flight on device 0's stream when the switch happens
        s0 = lib.np_get_stream()
        check(lib.np_set_device(1))
        assert lib.np_get_stream() != s0             # device 1 has its own stream
        d1 = hip.DeviceArray.from_host(y)
        got1 = hip.binary("add", d1, "full", d1, "full", 1, y.size).to_host().reshape(-1)
        assert (got1 == y + y).all()
        live_before = lib.np_live_allocs()
        e0.free()                                    # freeing an array of the OTHER device: goes back to that device's cache
        assert lib.np_live_allocs() == live_before - 1
        check(lib.np_set_device(0))
        assert lib.np_get_stream() == s0             # device 0's stream survived
        assert (d0.to_host() == x).all()             # ... and so did its array
        back = hip.unary("negate", d0)               # reuses the cached block on device 0
        assert (back.to_host() == -x).all()
        check(lib.np_set_device(1))
        assert (d1.to_host() == y).all()
        # device errors are per device: a wait that gave up on device 0 is not seen — let alone consumed — by device 1's sync
        import ctypes as C
        check(lib.np_set_device(0))
        check(lib.np_debug_raise_device_error(2))
        check(lib.np_set_device(1))
        assert lib.np_sync() == 0
        assert (d1.to_host() == y).all()
        check(lib.np_set_device(0))
        assert lib.np_sync() != 0 and b"device 0" in lib.np_last_error()
        bits = C.c_uint(0)
        check(lib.np_clear_device_error(C.byref(bits)))
        assert bits.value == 2 and lib.np_sync() == 0
        check(lib.np_set_device(1))
        d1.free()
        check(lib.np_set_device(0))
        for d in (d0, back):
            d.free()
    finally:
        check(lib.np_set_device(0))


def test_host_result_calls_from_many_threads(hip, oracle):
    lib = load()
    n_threads, rounds = 8, 40
    arrays = [synth.uniform((50_000 + 1013 * t,), 80 + t, -1.0, 1.0) for t in range(n_threads)]
    devs = [hip.DeviceArray.from_host(a) for a in arrays]
    want_sum = [hip.reduce_all("sum", d) for d in devs]            # single-threaded reference values (deterministic kernel)
    want_max = [float(a.max()) for a in arrays]
    want_med = [float(np.sort(a)[a.size // 2]) for a in arrays]
    errors = []
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            v, two, flag = C.c_float(), (C.c_float * 2)(), C.c_int()
            start.wait()
            for _ in range(rounds):
                check(lib.np_reduce_all(0, devs[t].ptr, devs[t].size, C.byref(v)))
                assert v.value == want_sum[t], ("sum", t, v.value, want_sum[t])
                check(lib.np_reduce_all(3, devs[t].ptr, devs[t].size, C.byref(v)))
                assert v.value == want_max[t], ("max", t)
                check(lib.np_order_stat(devs[t].ptr, devs[t].size, devs[t].size // 2, two))
                assert two[0] == want_med[t], ("median", t)
                check(lib.np_count_mismatch(0, devs[t].ptr, devs[t].ptr, devs[t].size, 0.0, 0.0, C.byref(flag)))
                assert flag.value == 0, ("array_equal", t)
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors[:3]
    for d in devs:
        d.free()
