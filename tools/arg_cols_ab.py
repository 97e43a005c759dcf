"""argmax over a wide inner axis (argreduce_cols_tile + the coalesced fold): how many workgroups per CU the axis should be cut for
(np_reduce_set_variant(4000000 + N)); same box, alternating.  Usage: python tools/arg_cols_ab.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer, UNARY_OPS, check, load

D.init(0)
lib = load()
t = Timer()
N = 110_000_000
ramp, big, out = D.DeviceArray((N,)), D.DeviceArray((N,)), D.DeviceArray((N,))
check(lib.np_arange(ramp.ptr, 0.0, 1.0, N))
check(lib.np_unary(UNARY_OPS["sin"], ramp.ptr, big.ptr, N, 0.0, 0.0))
ramp.free()
for _ in range(300):
    check(lib.np_argreduce(1, big.ptr, 1, 25000, 4000, out.ptr))
D.sync()
for outer, L, inner in ((1, 65536, 1024), (1, 9973, 9973), (1, 10007, 10007), (1, 9973, 9974), (1, 9973, 9972), (1, 9984, 9984), (1, 20000, 5000), (1, 16384, 6144), (1, 12000, 8192), (1, 6000, 16384), (1, 3000, 32768), (1, 40000, 2048), (1, 30000, 3000), (1, 25000, 4000), (64, 1500, 1000), (1, 390_000, 256), (8, 3000, 4096), (1, 2_000_000, 48 * 1)):
    if inner < 192:
        continue
    n = outer * L * inner
    assert n <= N
    best = {}
    for rnd in range(3):
        for w in (4, 2, 3, 6, 8, 12, 16):
            check(lib.np_reduce_set_variant(4000000 + w))
            for _ in range(3):
                check(lib.np_argreduce(1, big.ptr, outer, L, inner, out.ptr))
            D.sync()
            t.start()
            for _ in range(10):
                check(lib.np_argreduce(1, big.ptr, outer, L, inner, out.ptr))
            t.stop()
            best[w] = min(t.elapsed_ms() / 10, best.get(w, 1e9))
    check(lib.np_reduce_set_variant(4000000))          # back to the default rule
    print("  outer=%-4d len=%-8d inner=%-6d " % (outer, L, inner) + "  ".join("%d/CU %5.0f GB/s" % (w, 4.0 * n / best[w] / 1e6) for w in (2, 3, 4, 6, 8, 12, 16)), flush=True)
