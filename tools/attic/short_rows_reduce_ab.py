"""Axis-1 reductions of tall matrices with short rows (sum / max / mean over rows of 5 .. 63 floats): ms, GB/s over input +
output, check against numpy fp64.  Usage: NP_HIP_LIB=... python tools/short_rows_reduce_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
for rows, cols in ((10_000_000, 10), (12_500_000, 8), (5_000_000, 16), (3_000_000, 33), (2_000_000, 50), (1_600_000, 63), (20_000_000, 5), (300_001, 7)):
    n = rows * cols
    h = synth.uniform((n,), 9, -1.0, 1.0)
    src, out = _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * rows)
    _lib.check(lib.np_memcpy_h2d(src.ptr, h.ctypes.data, 4 * n))
    res = []
    for op, name in ((0, "sum"), (3, "max"), (4, "mean")):
        fn = lambda: _lib.check(lib.np_reduce_axis(op, src.ptr, rows, cols, 1, out.ptr, 0))
        for _ in range(3):
            fn()
        t = _lib.Timer(); t.start()
        for _ in range(20):
            fn()
        t.stop(); _lib.check(lib.np_sync())
        ms = t.elapsed_ms() / 20
        got = np.empty(rows, np.float32)
        _lib.check(lib.np_memcpy_d2h(got.ctypes.data, out.ptr, 4 * rows))
        x = h.reshape(rows, cols).astype(np.float64)
        ref = x.sum(1) if op == 0 else x.max(1) if op == 3 else x.mean(1)
        scale = np.abs(x).sum(1) if op == 0 else np.abs(x).mean(1) if op == 4 else 1.0
        ok = bool((np.abs(got - ref) <= 1e-5 * np.maximum(scale, 1e-30)).all()) if op != 3 else bool((got == ref.astype(np.float32)).all())
        res.append("%s %.4f ms %5.0f GB/s %s" % (name, ms, 4.0 * (n + rows) / ms / 1e6, "ok" if ok else "WRONG"))
    print("%9d x %-3d  %s" % (rows, cols, "   ".join(res)), flush=True)
    src.free(); out.free()
