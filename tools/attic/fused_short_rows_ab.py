"""sum(exp(X), axis 1) and max(|X - col| * 2, axis 1) over matrices with very many short rows, as ONE fused launch
(np_fused_chain_reduce_axis): ms and GB/s over the 4 B/elem read.  With NP_HIP_LIB pointing at an older build the same call
falls back to materialise + reduce.  Usage: python tools/fused_short_rows_ab.py"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth
from numpower_amd.lazy import Lazy   # noqa: F401  (adds NDArray.lazy)
from numpower_amd.ndarray import NDArray

lib = _lib.load()
_lib.check(lib.np_init(0))
for rows, cols in ((10_000_000, 10), (5_000_000, 16), (20_000_000, 5), (3_000_000, 33)):
    x = synth.uniform((rows, cols), 3, -1.0, 1.0)
    c = synth.uniform((rows, 1), 4, -1.0, 1.0)
    gx, gc = NDArray.array(x).gpu(), NDArray.array(c).gpu()
    out = []
    for label, build, ref in (("sum(exp(X),1)", lambda: gx.lazy().exp().sum(axis=1), lambda: np.exp(x.astype(np.float64)).sum(1)),
                              ("max(|X-col|*2,1)", lambda: ((gx.lazy() - gc).abs() * 2.0).max(axis=1), lambda: (np.abs(x - c) * np.float32(2.0)).max(1))):
        for _ in range(3):
            r = build()
        _lib.check(lib.np_sync())
        t = _lib.Timer(); t.start()
        for _ in range(10):
            r = build()
        t.stop(); _lib.check(lib.np_sync())
        ms = t.elapsed_ms() / 10
        got = r.cpu().numpy()
        want = ref()
        ok = bool((np.abs(got - want) <= 1e-5 * np.maximum(np.abs(want), 1e-30)).all())
        out.append("%s %.4f ms %5.0f GB/s %s" % (label, ms, 4.0 * rows * cols / ms / 1e6, "ok" if ok else "WRONG"))
    print("%9d x %-3d  %s" % (rows, cols, "   ".join(out)), flush=True)
