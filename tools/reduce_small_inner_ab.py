"""Column reductions over a few columns (np_reduce_axis, inner <= 128, long axis): the float4 walk of round 6 against the dword walk
(np_reduce_set_variant(4300001)).  Same box, alternating; results compared bit for bit where the order of additions is the same
(min / max) and within 1e-6 relative for sums.  Usage: python tools/reduce_small_inner_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
t = Timer()
N = 96_000_000
big = D.DeviceArray.from_host(synth.uniform((N,), 11, 0.0, 1.0))
for rows, cols in ((30_000_000, 3), (24_000_000, 4), (12_000_000, 8), (5_000_000, 17), (1_000_000, 64), (900_000, 100), (700_000, 128), (6_000_000, 16)):
    out = D.DeviceArray((cols,))
    res = {}
    best = {}
    for rnd in range(4):
        for variant in (1, 0):
            check(lib.np_reduce_set_variant(4300000 + variant))
            for op in ("sum", "max"):
                fn = lambda: check(lib.np_reduce_axis(0 if op == "sum" else 3, big.ptr, 1, rows, cols, out.ptr, 0))
                for _ in range(3):
                    fn()
                D.sync()
                t.start()
                for _ in range(20):
                    fn()
                t.stop()
                best[(variant, op)] = min(best.get((variant, op), 1e9), t.elapsed_ms() / 20)
                res[(variant, op)] = out.to_host().copy()
    check(lib.np_reduce_set_variant(4300000))
    assert (res[(0, "max")] == res[(1, "max")]).all()
    assert np.allclose(res[(0, "sum")], res[(1, "sum")], rtol=1e-6)
    gb = 4.0 * rows * cols / 1e6
    print("  %9d x %-4d sum: dword %5.0f GB/s  float4 %5.0f GB/s   max: dword %5.0f  float4 %5.0f" % (
        rows, cols, gb / best[(1, "sum")], gb / best[(0, "sum")], gb / best[(1, "max")], gb / best[(0, "max")]), flush=True)
    out.free()
