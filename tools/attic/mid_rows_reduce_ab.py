"""Axis-1 sum / max / mean over rows of 49 .. 4000 floats (batch x hidden matrices): ms, GB/s over input + output, check against
numpy fp64.  NP_HIP_LIB selects the build for a same-box A/B."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth

lib = _lib.load()
_lib.check(lib.np_init(0))
for rows, cols in ((1_600_000, 64), (1_000_000, 100), (400_000, 256), (400_001, 250), (200_000, 512), (130_000, 768), (100_000, 1000), (100_003, 1001), (50_000, 2000), (25_000, 4000), (25_000, 4095), (5000, 77)):
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
    print("%8d x %-5d  %s" % (rows, cols, "   ".join(res)), flush=True)
    src.free(); out.free()
