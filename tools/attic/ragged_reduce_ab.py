"""Axis-0 reductions of matrices whose row length is not a multiple of 4 (rows start on odd 4-byte boundaries): GB/s over
4 B/elem, checked against numpy in fp64.  Usage: NP_HIP_LIB=... python tools/ragged_reduce_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
for rows, cols in ((10007, 10007), (25000, 4001), (25000, 4000), (100_000, 1001), (5000, 20_001), (65536, 4096), (1_000_000, 67), (40_000, 2502)):
    n = rows * cols
    h = synth.uniform((n,), 7, -1.0, 1.0)
    src, out = _lib.DeviceBuffer(4 * n), _lib.DeviceBuffer(4 * cols)
    _lib.check(lib.np_memcpy_h2d(src.ptr, h.ctypes.data, 4 * n))
    res = []
    for op, name in ((0, "sum"), (3, "max"), (4, "mean")):
        fn = lambda: _lib.check(lib.np_reduce_axis(op, src.ptr, 1, rows, cols, out.ptr, 0))
        for _ in range(3):
            fn()
        t = _lib.Timer(); t.start()
        for _ in range(20):
            fn()
        t.stop(); _lib.check(lib.np_sync())
        ms = t.elapsed_ms() / 20
        got = np.empty(cols, np.float32)
        _lib.check(lib.np_memcpy_d2h(got.ctypes.data, out.ptr, 4 * cols))
        x = h.reshape(rows, cols).astype(np.float64)
        ref = x.sum(0) if op == 0 else x.max(0) if op == 3 else x.mean(0)
        scale = np.abs(x).sum(0) if op == 0 else np.abs(x).mean(0) if op == 4 else 1.0
        ok = bool((np.abs(got - ref) <= 1e-5 * np.maximum(scale, 1e-30)).all()) if op != 3 else bool((got == ref.astype(np.float32)).all())
        res.append("%s %.3f ms %5.0f GB/s %s" % (name, ms, 4.0 * n / ms / 1e6, "ok" if ok else "WRONG"))
    print("%8d x %-6d  %s" % (rows, cols, "   ".join(res)), flush=True)
    src.free(); out.free()
