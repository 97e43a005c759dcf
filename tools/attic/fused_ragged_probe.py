"""exp(X) + row / exp(X) + col as one fused launch with row lengths that are / are not multiples of 4: ms, GB/s over 8 B/elem.
NP_HIP_LIB selects the build for a same-box A/B."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np

from numpower_amd import _lib, synth
from numpower_amd.lazy import Lazy   # noqa: F401
from numpower_amd.ndarray import NDArray

lib = _lib.load()
_lib.check(lib.np_init(0))
for rows, cols in ((25000, 4000), (25000, 4001), (33333, 3001), (10_000_000, 7), (10_000_000, 8)):
    x = synth.uniform((rows, cols), 3, -1.0, 1.0)
    r = synth.uniform((cols,), 4, -1.0, 1.0)
    c = synth.uniform((rows, 1), 5, -1.0, 1.0)
    gx, gr, gc = NDArray.array(x).gpu(), NDArray.array(r).gpu(), NDArray.array(c).gpu()
    out = []
    for label, build, ref in (("exp(X)+row", lambda: (gx.lazy().exp() + gr).eval(), lambda: np.exp(x.astype(np.float64)) + r),
                              ("exp(X)+col", lambda: (gx.lazy().exp() + gc).eval(), lambda: np.exp(x.astype(np.float64)) + c)):
        for _ in range(3):
            y = build()
        _lib.check(lib.np_sync())
        t = _lib.Timer(); t.start()
        for _ in range(10):
            y = build()
        t.stop(); _lib.check(lib.np_sync())
        ms = t.elapsed_ms() / 10
        got = y.cpu().numpy()
        want = ref()
        ok = bool((np.abs(got - want) <= 1e-5 * (np.exp(x.astype(np.float64)) + 1.0)).all())   # |exp(x)| + |operand| bounds the rounding
        out.append("%s %.4f ms %5.0f GB/s %s" % (label, ms, 8.0 * rows * cols / ms / 1e6, "ok" if ok else "WRONG"))
    print("%9d x %-5d  %s" % (rows, cols, "   ".join(out)), flush=True)
