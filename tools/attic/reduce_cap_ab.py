"""Grid size of the streaming reductions' first pass (np_reduce_set_variant(k) = k workgroups per CU): full sum of
1e8 floats (np_reduce_all_dev, 4 B/elem), allclose on 2 x 1e8 (np_count_mismatch, 8 B/elem, includes the D2H of
the verdict) and sum(axis 0) of 65536 x 4096 (np_reduce_axis, 4 B/elem).  Interleaved rounds, 30 launches each.
Usage: python tools/reduce_cap_ab.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5))
b = D.DeviceArray.from_host(synth.uniform((N,), 6))
sink = D.DeviceArray((4,))
rows, cols = 65536, 4096
X = D.DeviceArray((rows, cols))
D.fill(X, 0.5)
out = D.DeviceArray((cols,))
flag = C.c_int(0)


def t(fn, iters=30):
    for _ in range(3):
        fn()
    D.sync()
    tm = Timer()
    tm.start()
    for _ in range(iters):
        fn()
    tm.stop()
    return tm.elapsed_ms() / iters


cases = {
    "sum 1e8 (4 B/elem)": (4.0 * N, lambda: check(lib.np_reduce_all_dev(0, a.ptr, N, sink.ptr))),
    "allclose 2 x 1e8 (8 B/elem)": (8.0 * N, lambda: check(lib.np_count_mismatch(1, a.ptr, b.ptr, N, 1e-5, 1e-8, C.byref(flag)))),
    "sum axis 0 65536x4096 (4 B/elem)": (4.0 * rows * cols, lambda: D.reduce_axis("sum", X, 0, out=out)),
}
caps = [8, 12, 2049, 2304, 2560, 2816, 3073, 3328, 3584, 3840, 5120, 6145, 7168]
res = {}
for rnd in range(3):
    for cap in caps:
        check(lib.np_reduce_set_variant(cap))
        for name, (nbytes, fn) in cases.items():
            res.setdefault((name, cap), []).append(nbytes / t(fn) / 1e6)
check(lib.np_reduce_set_variant(0))
for name in cases:
    for cap in caps:
        v = res[(name, cap)]
        print("%-36s %5d %s  GB/s: %s  median %.0f" % (name, cap, "WG/CU" if cap < 1000 else "WGs  ", " ".join("%.0f" % x for x in v), float(np.median(v))))
