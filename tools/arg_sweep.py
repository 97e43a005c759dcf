"""np_argreduce over the shapes of tools/misc_sweep.py (and a few more), on data with one clear winner per output (not the
all-equal fill of misc_sweep): GB/s of the one pass over the input, 10 launches behind a warm-up.  Usage: python tools/arg_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer, UNARY_OPS, check, load

D.init(0)
lib = load()
t = Timer()
N = 100_000_000
ramp, out = D.DeviceArray((N,)), D.DeviceArray((N,))
big = D.DeviceArray((N,))
check(lib.np_arange(ramp.ptr, 0.0, 1.0, N))
check(lib.np_unary(UNARY_OPS["sin"], ramp.ptr, big.ptr, N, 0.0, 0.0))       # values in [-1, 1], no long runs of ties
ramp.free()
for _ in range(300):                                                            # warm the clock
    check(lib.np_argreduce(1, big.ptr, 1, N, 1, out.ptr))
D.sync()
for outer, L, inner in ((1, N, 1), (3, 30_000_000, 1), (65536, 1024, 1), (10000, 10000, 1), (1, 30_000_000, 3), (1, 20_000_000, 5), (1, 65536, 1024),
                        (1000, 1000, 100), (1, 9973, 9973), (1, 1_000_000, 64), (1, 390_000, 256), (64, 1500, 1000), (1, 25000, 4000)):
    n = outer * L * inner
    assert n <= N, (outer, L, inner)          # the sweep reads `big`: never past it
    for is_max in (1, 0):
        for _ in range(3):
            check(lib.np_argreduce(is_max, big.ptr, outer, L, inner, out.ptr))
        D.sync()
        t.start()
        for _ in range(10):
            check(lib.np_argreduce(is_max, big.ptr, outer, L, inner, out.ptr))
        t.stop()
        ms = t.elapsed_ms() / 10
        print("  %s outer=%-6d len=%-10d inner=%-6d %8.3f ms %7.0f GB/s" % ("argmax" if is_max else "argmin", outer, L, inner, ms, 4.0 * n / ms / 1e6), flush=True)
