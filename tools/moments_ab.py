"""np_moments / np_weighted_sums at 1e8 and a few smaller sizes: microseconds per call (host-result call, events around
batches) and TB/s at the algorithmic bytes (4 B/elem; 8 B/elem for the weighted sums), next to np_reduce_all(sum) of the
same array — the one-read bound a statistics pass is measured against.
    python tools/moments_ab.py [iters]"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth   # noqa: E402
from numpower_amd._lib import check, load     # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
D.init(0)
lib = load()


def run(fn, n=iters):
    for _ in range(10):
        fn()
    check(lib.np_sync())
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        check(lib.np_sync())
        best = min(best, (time.perf_counter() - t0) / n)
    return best * 1e6


for N in (100_000_000, 10_000_000, 1_000_000, 100_000):
    a = synth.uniform((N,), 5, 0.0, 1.0)
    b = synth.uniform((N,), 6, 0.1, 2.0)
    da, db = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b)
    s, mean, m2, saw, sw = C.c_float(), C.c_float(), C.c_float(), C.c_float(), C.c_float()
    two = D.DeviceArray((2,))
    # warm the clocks
    t_end = time.time() + 0.3
    while time.time() < t_end:
        check(lib.np_reduce_all(0, da.ptr, N, C.byref(s)))
    us_sum = run(lambda: check(lib.np_reduce_all(0, da.ptr, N, C.byref(s))))
    us_mom = run(lambda: check(lib.np_moments(da.ptr, N, C.byref(mean), C.byref(m2))))
    us_mom_dev = run(lambda: check(lib.np_moments_dev(da.ptr, N, two.ptr)))
    us_w = run(lambda: check(lib.np_weighted_sums(da.ptr, db.ptr, N, C.byref(saw), C.byref(sw))))
    a64 = a.astype(np.float64)
    err = abs(m2.value / N - a64.var()) / a64.var()
    print("N %10d  sum %8.1f us %5.2f TB/s | moments %8.1f us %5.2f TB/s (dev result %8.1f us %5.2f) rel err %.1e | weighted %8.1f us %5.2f TB/s"
          % (N, us_sum, 4.0 * N / us_sum / 1e6, us_mom, 4.0 * N / us_mom / 1e6, us_mom_dev, 4.0 * N / us_mom_dev / 1e6, err,
             us_w, 8.0 * N / us_w / 1e6), flush=True)
    for d in (da, db, two):
        d.free()
