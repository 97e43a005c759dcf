"""Tall-skinny matrix . vector (np_sgemv, M rows of N <= 64 floats): ms and GB/s over the bytes of A + y, checked against fp64.
Usage: NP_HIP_LIB=... python tools/sgemv_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
for m, n in ((10_000_000, 10), (10_000_000, 3), (5_000_000, 16), (3_000_000, 33), (2_000_000, 50), (1_600_000, 64), (12_500_000, 8), (1_000_001, 7), (70_000, 13)):
    a = synth.uniform((m, n), 37, -1.0, 1.0)
    x = synth.uniform((n,), 38, -1.0, 1.0)
    da, dx, dy = _lib.DeviceBuffer(4 * m * n), _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * m)
    _lib.check(lib.np_memcpy_h2d(da.ptr, a.ctypes.data, 4 * m * n))
    _lib.check(lib.np_memcpy_h2d(dx.ptr, x.ctypes.data, 4 * n))
    fn = lambda: _lib.check(lib.np_sgemv(m, n, da.ptr, dx.ptr, dy.ptr))
    for _ in range(3):
        fn()
    t = _lib.Timer(); t.start()
    for _ in range(20):
        fn()
    t.stop(); _lib.check(lib.np_sync())
    ms = t.elapsed_ms() / 20
    got = np.empty(m, np.float32)
    _lib.check(lib.np_memcpy_d2h(got.ctypes.data, dy.ptr, 4 * m))
    ref = a.astype(np.float64) @ x.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(x).astype(np.float64)
    ok = bool((np.abs(got - ref) <= 1e-5 * scale).all())
    print("%9d x %-3d  %.4f ms  %5.0f GB/s  %s" % (m, n, ms, 4.0 * (m * n + m) / ms / 1e6, "ok" if ok else "WRONG"), flush=True)
    da.free(); dx.free(); dy.free()
