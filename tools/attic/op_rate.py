"""Launch-bound regime: how many small ops per second the library enqueues (np_binary / np_unary on n elements, called back
to back through ctypes, one sync at the end), and the same through the host layer (NDArray_Add_Float: + result allocation
from the pool + NDArray bookkeeping).  Usage: python tools/op_rate.py"""
import ctypes as C
import json
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import _lib, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS

lib = _lib.load()
_lib.check(lib.np_init(0))
host = C.CDLL(str(Path(_lib.lib_path()).parent / "libnumpower_host.so"))
host.NDArray_FromHostBuffer.restype = C.c_void_p
host.NDArray_FromHostBuffer.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_int]
host.NDArray_ToGPU.restype = C.c_void_p
host.NDArray_ToGPU.argtypes = [C.c_void_p]
host.NDArray_Add_Float.restype = C.c_void_p
host.NDArray_Add_Float.argtypes = [C.c_void_p, C.c_void_p]
host.NDArray_FREE.argtypes = [C.c_void_p]
for n in (256, 4096, 65536, 1_048_576):
    h = synth.uniform((n,), 3, 0.0, 1.0)
    a, b, o = _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * n)
    _lib.check(lib.np_memcpy_h2d(a.ptr, h.ctypes.data, 4 * n))
    _lib.check(lib.np_memcpy_h2d(b.ptr, h.ctypes.data, 4 * n))
    reps = 20000
    row = {"n": n}
    # the bare foreign-function call, for scale
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.np_live_allocs()
    row["ctypes_call_us"] = round((time.perf_counter() - t0) / reps * 1e6, 2)
    add, exp = BINARY_OPS["add"], UNARY_OPS["exp"]
    for name, fn in (("np_binary_add", lambda: lib.np_binary(add, a.ptr, 0, b.ptr, 0, o.ptr, 1, n, 0, 0)),
                     ("np_unary_exp", lambda: lib.np_unary(exp, a.ptr, o.ptr, n, 0.0, 0.0))):
        for _ in range(200):
            fn()
        _lib.check(lib.np_sync())
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        enq = time.perf_counter() - t0
        _lib.check(lib.np_sync())
        tot = time.perf_counter() - t0
        row[name + "_enqueue_us"] = round(enq / reps * 1e6, 2)
        row[name + "_total_us"] = round(tot / reps * 1e6, 2)
    shape = (C.c_int * 1)(n)
    ha = host.NDArray_FromHostBuffer(h.ctypes.data, shape, 1)
    ga, gb = host.NDArray_ToGPU(ha), host.NDArray_ToGPU(ha)
    for _ in range(200):
        host.NDArray_FREE(host.NDArray_Add_Float(ga, gb))
    _lib.check(lib.np_sync())
    t0 = time.perf_counter()
    for _ in range(reps):
        host.NDArray_FREE(host.NDArray_Add_Float(ga, gb))
    enq = time.perf_counter() - t0
    _lib.check(lib.np_sync())
    tot = time.perf_counter() - t0
    row["NDArray_Add_Float+FREE_enqueue_us"] = round(enq / reps * 1e6, 2)
    row["NDArray_Add_Float+FREE_total_us"] = round(tot / reps * 1e6, 2)
    for p in (ga, gb, ha):
        host.NDArray_FREE(p)
    print(json.dumps(row), flush=True)
    a.free(); b.free(); o.free()
