"""argmax over a few columns (the flat float4 walks: argreduce_small_inner / _inner4): workgroups per CU the axis is cut for
(np_reduce_set_variant(4100000 + N)); same box, alternating.  Usage: python tools/arg_flat_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer, UNARY_OPS, check, load

D.init(0)
lib = load()
t = Timer()
N = 100_000_000
ramp, big, out = D.DeviceArray((N,)), D.DeviceArray((N,)), D.DeviceArray((N,))
check(lib.np_arange(ramp.ptr, 0.0, 1.0, N))
check(lib.np_unary(UNARY_OPS["sin"], ramp.ptr, big.ptr, N, 0.0, 0.0))
ramp.free()
for _ in range(300):
    check(lib.np_argreduce(1, big.ptr, 1, 30_000_000, 3, out.ptr))
D.sync()
for outer, L, inner in ((1, 30_000_000, 3), (1, 20_000_000, 5), (1, 1_000_000, 64), (1, 390_000, 256), (1000, 1000, 100), (1, 6_000_000, 16), (16, 100_000, 60)):
    n = outer * L * inner
    assert n <= N
    best = {}
    for rnd in range(3):
        for w in (8, 2, 3, 4, 6, 12):
            check(lib.np_reduce_set_variant(4100000 + w))
            for _ in range(3):
                check(lib.np_argreduce(1, big.ptr, outer, L, inner, out.ptr))
            D.sync()
            t.start()
            for _ in range(10):
                check(lib.np_argreduce(1, big.ptr, outer, L, inner, out.ptr))
            t.stop()
            best[w] = min(t.elapsed_ms() / 10, best.get(w, 1e9))
    check(lib.np_reduce_set_variant(4100000))
    print("  outer=%-4d len=%-9d inner=%-4d " % (outer, L, inner) + "  ".join("%d/CU %5.0f GB/s" % (w, 4.0 * n / best[w] / 1e6) for w in (2, 3, 4, 6, 8, 12)), flush=True)
