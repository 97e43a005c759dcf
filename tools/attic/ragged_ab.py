"""Broadcast add with row lengths that are not multiples of 4 (a float4 of the result may straddle two rows): X + row and
X + col, GB/s over 8 B/elem, checked against numpy on a sample.  Usage: NP_HIP_LIB=... python tools/ragged_ab.py"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
N = 101_000_000
h = synth.uniform((N,), 1, -1.0, 1.0)
big, out = _lib.DeviceBuffer(4 * N), _lib.DeviceBuffer(4 * N)
_lib.check(lib.np_memcpy_h2d(big.ptr, h.ctypes.data, 4 * N))
for rows, cols in ((25000, 4000), (25000, 4001), (33333, 3001), (10_000_000, 7), (20_000_000, 5), (33_000_000, 3), (1_000_000, 101), (7, 10_000_001)):
    assert rows * cols <= N
    n = rows * cols
    hr, hc = synth.uniform((cols,), 2, -1.0, 1.0), synth.uniform((rows,), 3, -1.0, 1.0)
    row, col = _lib.DeviceBuffer(4 * cols), _lib.DeviceBuffer(4 * rows)
    _lib.check(lib.np_memcpy_h2d(row.ptr, hr.ctypes.data, 4 * cols))
    _lib.check(lib.np_memcpy_h2d(col.ptr, hc.ctypes.data, 4 * rows))
    res = {}
    for name, ptr, kind in (("row", row.ptr, 2), ("col", col.ptr, 3)):
        fn = lambda: _lib.check(lib.np_binary(0, big.ptr, 0, ptr, kind, out.ptr, rows, cols, 0, 0))
        for _ in range(3):
            fn()
        t = _lib.Timer(); t.start()
        for _ in range(20):
            fn()
        t.stop(); _lib.check(lib.np_sync())
        ms = t.elapsed_ms() / 20
        got = np.empty(n, np.float32)
        _lib.check(lib.np_memcpy_d2h(got.ctypes.data, out.ptr, 4 * n))
        x = h[:n].reshape(rows, cols)
        ref = (x + hr) if name == "row" else (x + hc[:, None])
        ok = bool((got.reshape(rows, cols) == ref).all())
        res[name] = "%.3f ms %5.0f GB/s %s" % (ms, 8.0 * n / ms / 1e6, "ok" if ok else "WRONG")
    print("%9d x %-9d  X+row %s   X+col %s" % (rows, cols, res["row"], res["col"]), flush=True)
    row.free(); col.free()
