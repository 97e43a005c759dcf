"""argmax / sum over axis 0 of matrices whose rows are not 16-byte multiples (9973^2, 10007^2) next to aligned neighbours
(9984^2 = 39 * 256): time per call and — under `rocprofv3 --pmc FETCH_SIZE` — the bytes fetched, to tell a traffic problem
(boundary lines fetched by two tiles) from a latency problem.  Usage: python tools/arg_unaligned_probe.py [iters]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer, UNARY_OPS, check, load

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
D.init(0)
lib = load()
t = Timer()
N = 110_000_000
ramp, big, out = D.DeviceArray((N,)), D.DeviceArray((N,)), D.DeviceArray((20000,))
check(lib.np_arange(ramp.ptr, 0.0, 1.0, N))
check(lib.np_unary(UNARY_OPS["sin"], ramp.ptr, big.ptr, N, 0.0, 0.0))
ramp.free()
for rows, cols in ((9973, 9973), (9984, 9984), (10007, 10007), (9973, 9972), (9973, 9974), (25000, 4001), (5000, 20001), (65536, 1000)):
    n = rows * cols
    for name, fn, variant in (("argmax, tiles in launch order", lambda: check(lib.np_argreduce(1, big.ptr, 1, rows, cols, out.ptr)), 4200001),
                              ("argmax, tiles in XCD runs    ", lambda: check(lib.np_argreduce(1, big.ptr, 1, rows, cols, out.ptr)), 4200000),
                              ("sum,    tiles in launch order", lambda: check(lib.np_reduce_axis(0, big.ptr, 1, rows, cols, out.ptr, 0)), 4200001),
                              ("sum,    tiles in XCD runs    ", lambda: check(lib.np_reduce_axis(0, big.ptr, 1, rows, cols, out.ptr, 0)), 4200000)):
        check(lib.np_reduce_set_variant(variant))
        check(lib.np_debug_raise_device_error(0))          # marker launch: sections in a counter CSV
        for _ in range(5):
            fn()
        D.sync()
        t.start()
        for _ in range(iters):
            fn()
        t.stop()
        ms = t.elapsed_ms() / iters
        print("%s %5d x %-5d  %7.1f us  %5.0f GB/s" % (name, rows, cols, ms * 1e3, 4.0 * n / ms / 1e6), flush=True)
