"""Deep-K products of a few tiles (100 x 100 x 100000: four 64 x 64 tiles under 1563 K-tiles): the plan of rounds 1-4 cuts K into
up to 128 chunks that run as the batch of the register-staged 64 x 64 kernel (cfg 2) and folds them with np_reduce_axis; round 5
runs the same chunks on the k-quartered tiles (sgemm_kq_kernel, operands straight from memory).  np_sgemm_set_variant(-22) and (-40): the
old plans, (-23) and (-99): the default planner, (-24): the K-chunked k-quartered plan wherever one exists.  Same box, alternating; every
form's result is checked against fp64 first (1e-6 of sum |a||b|).
Usage: python tools/gemm_deep_k_ab.py"""
import ctypes as C
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
out = (C.c_double * 11)()
SHAPES = ((100, 100, 100000), (128, 128, 65536), (64, 64, 100000), (32, 32, 1000000), (200, 200, 50000), (256, 256, 32768),
          (300, 300, 20000), (384, 384, 30000), (100, 100, 100001), (100, 100, 10000), (512, 512, 16384), (96, 160, 70002),
          (32, 64, 100000), (48, 48, 200000), (64, 64, 30000), (20, 40, 50000), (64, 36, 16384), (64, 64, 400000), (100, 100, 3000),
          (160, 96, 250000), (640, 640, 8192), (100, 2000, 20000),
          (256, 256, 2048), (256, 256, 4096), (512, 512, 4096), (384, 384, 2048), (128, 128, 4096), (640, 640, 2048), (100, 100, 2048),
          (512, 256, 8192), (768, 768, 3072), (128, 128, 2048), (320, 320, 6144), (704, 704, 4096))
for (m, n, k) in SHAPES:
    ha, hb = synth.uniform((m, k), 31, -1.0, 1.0), synth.uniform((k, n), 32, -1.0, 1.0)
    a, b, c = D.DeviceArray.from_host(ha), D.DeviceArray.from_host(hb), D.DeviceArray((m, n))
    ref = ha.astype(np.float64) @ hb.astype(np.float64)
    mag = np.abs(ha).astype(np.float64) @ np.abs(hb).astype(np.float64)
    reps = max(10, min(200, int(1e11 / (2.0 * m * n * k))))
    best, plans, errs = {}, {}, {}
    for rnd in range(3):
        for v in (-22, -23, -24):
            check(lib.np_sgemm_set_variant(v))
            check(lib.np_sgemm_set_variant(-40 if v == -22 else -(40 + 59)))   # (the thin K-chunk kernels' underfilled shapes: old / new)
            if rnd == 0:
                check(lib.np_sgemm_debug_plan(m, n, k, 1, 0, out))
                plans[v] = "cfg %d rows %d S %d model %.1f" % (out[0], out[1], out[2], out[3])
                D.fill(c, -7.0)
                D.sgemm(a, b, out=c)
                errs[v] = float((np.abs(c.to_host().astype(np.float64) - ref) / mag).max())
            for _ in range(5):
                D.sgemm(a, b, out=c)
            D.sync()
            t.start()
            for _ in range(reps):
                D.sgemm(a, b, out=c)
            t.stop()
            us = t.elapsed_ms() / reps * 1e3
            best[v] = min(us, best.get(v, 1e9))
    check(lib.np_sgemm_set_variant(-23))
    print("%4d x %4d x %7d  " % (m, n, k) + "  ".join("[%d] %6.1f us %5.1f TF err %.1e (%s)" % (v, best[v], 2.0 * m * n * k / best[v] / 1e6, errs[v], plans[v]) for v in (-22, -23, -24)), flush=True)
    assert max(errs.values()) < 1e-6, errs
    for d in (a, b, c):
        d.free()
