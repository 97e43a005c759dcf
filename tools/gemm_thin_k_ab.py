"""Thin-K products on the 256 x 128 LDS-DMA kernel (4096 x 4096 x 256: 512 workgroups = one resident round, 16 K-tiles, then 64 MiB
of C stored by everybody at once): does the priority alternation between the two workgroups of a CU (np_sgemm_set_variant(-(100
+ p)): p K-tiles per phase, 0 = off, default 16) help or hurt when the epilogue is a fifth of the kernel?  Same box, alternating.
Usage: python tools/gemm_thin_k_ab.py"""
import sys
import time
from pathlib import Path
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
for (m, n, k) in ((4096, 4096, 256), (4096, 4096, 512), (8192, 8192, 256), (4096, 4096, 128), (4096, 4096, 1024), (4096, 4096, 4096)):
    a = D.DeviceArray.from_host(synth.uniform((m, k), 31, -1.0, 1.0))
    b = D.DeviceArray.from_host(synth.uniform((k, n), 32, -1.0, 1.0))
    c = D.DeviceArray((m, n))
    reps = max(10, min(200, int(3e11 / (2.0 * m * n * k))))
    best = {}
    for rnd in range(3):
        for p in (16, 0, 2, 4, 8, 32):
            check(lib.np_sgemm_set_variant(-(100 + p)))
            for _ in range(10):
                D.sgemm(a, b, out=c)
            D.sync()
            t.start()
            for _ in range(reps):
                D.sgemm(a, b, out=c)
            t.stop()
            us = t.elapsed_ms() / reps * 1e3
            best[p] = min(us, best.get(p, 1e9))
    check(lib.np_sgemm_set_variant(-(100 + 16)))
    print("%5d x %5d x %5d  " % (m, n, k) + "  ".join("p=%-2d %7.1f us %6.1f TF" % (p, best[p], 2.0 * m * n * k / best[p] / 1e6) for p in (16, 0, 2, 4, 8, 32)), flush=True)
    for d in (a, b, c):
        d.free()
