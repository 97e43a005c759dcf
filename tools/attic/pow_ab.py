"""pow at 10^8 elements: time and accuracy against fp64 (positive bases: the fp64 fast path; mixed
signs / zeros: the library powf).  Usage: python tools/pow_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer
D.init(0); t = Timer()
N = 100_000_000
for label, lo, hi in (("base in [0.5, 4)", 0.5, 4.0), ("base in [-2, 2)", -2.0, 2.0), ("base in [0, 1) (bench.py)", 0.0, 1.0)):
    hx = synth.uniform((N,), 8, lo, hi); hy = synth.uniform((N,), 9, 0.5, 4.0)
    x = D.DeviceArray.from_host(hx); y = D.DeviceArray.from_host(hy); o = D.DeviceArray((N,))
    for _ in range(5): D.binary("pow", x, "full", y, "full", 1, N, out=o)
    D.sync(); t.start()
    for _ in range(50): D.binary("pow", x, "full", y, "full", 1, N, out=o)
    t.stop(); ms = t.elapsed_ms() / 50
    got = o.to_host()[:4_000_000]
    with np.errstate(all="ignore"):
        ref64 = np.power(hx[:4_000_000].astype(np.float64), hy[:4_000_000].astype(np.float64))
    ok = np.isfinite(ref64) & (ref64 != 0)
    rel = np.abs(got[ok].astype(np.float64) - ref64[ok]) / np.abs(ref64[ok])
    exact = (got[ok] == ref64[ok].astype(np.float32)).mean()
    nan_same = (np.isnan(got) == np.isnan(ref64)).all()
    print("  %-18s %7.3f ms %6.0f GB/s   max rel %.2e   correctly rounded %.4f   NaN pattern same: %s"
          % (label, ms, 12.0 * N / ms / 1e6, rel.max(), exact, nan_same), flush=True)
    x.free(); y.free(); o.free()
