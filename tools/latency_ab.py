"""Host-result latency: np_reduce_all / np_order_stat / np_count_mismatch called back to back from the host (every call
returns a host value, i.e. waits), with the wait done by hipStreamSynchronize (variant 0) or by spinning on a
stream-written flag (variant 1).  Usage: python tools/latency_ab.py"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
out = C.c_float(0.0)
two = (C.c_float * 2)()
flag = C.c_int(0)
for n in (1024, 100_000, 1_000_000, 10_000_000, 100_000_000):
    h = synth.uniform((n,), 3, 0.0, 1.0)
    a, b = _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * n)
    _lib.check(lib.np_memcpy_h2d(a.ptr, h.ctypes.data, 4 * n))
    _lib.check(lib.np_memcpy_h2d(b.ptr, h.ctypes.data, 4 * n))
    row = {"n": n}
    for rnd in range(2):
        for variant in (0, 1, 2):
            _lib.check(lib.np_runtime_set_variant(variant))
            calls = {
                "sum": lambda: _lib.check(lib.np_reduce_all(0, a.ptr, n, C.byref(out))),
                "allclose": lambda: _lib.check(lib.np_count_mismatch(1, a.ptr, b.ptr, n, 1e-5, 1e-8, C.byref(flag))),
                "median": lambda: _lib.check(lib.np_order_stat(a.ptr, n, n // 2, two)),
            }
            for name, fn in calls.items():
                reps = 200 if n <= 10_000_000 else 50
                for _ in range(10):
                    fn()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                us = (time.perf_counter() - t0) / reps * 1e6
                key = "%s_%s_us" % (name, ("sync", "flag", "watch")[variant])
                row[key] = round(min(us, row.get(key, 1e9)), 2)
    _lib.check(lib.np_runtime_set_variant(2))
    print(json.dumps(row), flush=True)
    a.free(); b.free()
