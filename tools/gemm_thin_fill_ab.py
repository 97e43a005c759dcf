"""17 .. 2047 rows, N <= 64, K >= 16384: the thin K-chunk kernels against the planner's K-chunked k-quartered tiles, by how many
workgroups the thin launch would have — np_sgemm_set_variant(-(40 + q)) hands a thin launch of fewer than q / 4 workgroups per CU
to the planner (q = 0: never, 59: the default).  Same box, alternating; each form checked against fp64 first.
Usage: python tools/gemm_thin_fill_ab.py"""
import sys
import time
from pathlib import Path

import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
t = Timer()
warm = D.DeviceArray.from_host(synth.uniform((2048, 2048), 1, -1, 1))
wc = D.DeviceArray((2048, 2048))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(20):
        D.sgemm(warm, warm, out=wc)
    D.sync()
QS = (0, 1, 2, 4, 8, 59)
SHAPES = ((64, 64, 100000), (64, 64, 400000), (32, 32, 1000000), (32, 64, 2000000), (64, 64, 1000000), (100, 20, 100000), (200, 32, 50000),
          (500, 16, 30000), (40, 24, 200000), (1000, 8, 100000), (300, 12, 65536), (1500, 32, 20000), (48, 48, 16384), (17, 17, 3000000),
          (128, 64, 500000), (700, 48, 40000), (64, 64, 4000000))
for (m, n, k) in SHAPES:
    ha, hb = synth.uniform((m, k), 31, -1.0, 1.0), synth.uniform((k, n), 32, -1.0, 1.0)
    a, b, c = D.DeviceArray.from_host(ha), D.DeviceArray.from_host(hb), D.DeviceArray((m, n))
    ref = ha.astype(np.float64) @ hb.astype(np.float64)
    mag = np.abs(ha).astype(np.float64) @ np.abs(hb).astype(np.float64)
    reps = max(5, min(100, int(5e10 / (2.0 * m * n * k))))
    best, errs = {}, {}
    for rnd in range(3):
        for q in QS:
            check(lib.np_sgemm_set_variant(-(40 + q)))
            if rnd == 0:
                D.fill(c, -7.0)
                D.sgemm(a, b, out=c)
                errs[q] = float((np.abs(c.to_host().astype(np.float64) - ref) / mag).max())
            for _ in range(3):
                D.sgemm(a, b, out=c)
            D.sync()
            t.start()
            for _ in range(reps):
                D.sgemm(a, b, out=c)
            t.stop()
            best[q] = min(t.elapsed_ms() / reps * 1e3, best.get(q, 1e9))
    check(lib.np_sgemm_set_variant(-(40 + 59)))
    mb = 4.0 * (m * k + k * n) / 1e6
    print("%5d x %3d x %8d (%6.0f MB)  " % (m, n, k, mb) + "  ".join("q=%-2d %7.1f us %4.2f TB/s" % (q, best[q], mb / best[q]) for q in QS), flush=True)
    assert max(errs.values()) < 1e-6, errs
    for d in (a, b, c):
        d.free()
