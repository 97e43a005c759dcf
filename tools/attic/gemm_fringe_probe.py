"""Dev tool: what the pieces of a peeled 4097^3 product cost — the 4096 x 4096 x 4097 block, the thin row and
column products along the ragged edges — against the whole product.  Usage: python tools/gemm_fringe_probe.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer, check, load
D.init(0); lib = load()
shapes = [(4097, 4097, 4097), (4096, 4096, 4097), (1, 4097, 4097), (2, 4097, 4097), (8, 4097, 4097), (4096, 1, 4097), (4096, 2, 4097),
          (4096, 8, 4097), (2049, 2049, 2049), (2048, 2048, 2049), (1, 2049, 2049), (2048, 1, 2049), (3001, 3001, 3001), (2816, 2944, 3001),
          (185, 3001, 3001), (2816, 57, 3001), (12, 4096, 4096), (16, 8192, 8192), (32, 8192, 8192), (64, 8192, 8192), (128, 8192, 8192), (16, 4096, 16384), (8192, 16, 8192), (8192, 1, 8192), (20000, 24, 3000), (1, 8192, 8192), (4, 8192, 8192), (8, 8192, 8192), (8, 1024, 1024), (1, 1024, 16384), (1, 100000, 1000), (3, 300, 100000)]
for (m, n, k) in shapes:
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    D.unary("sin", a, out=a); D.unary("cos", b, out=b)
    for _ in range(3): D.sgemm(a, b, out=c)
    D.sync(); t = Timer(); t.start()
    reps = 20
    for _ in range(reps): D.sgemm(a, b, out=c)
    t.stop(); ms = t.elapsed_ms() / reps
    extra = ""
    if m <= 16:   # the tiled kernels these shapes ran on before sgemm_fewrows_kernel
        check(lib.np_sgemm_set_variant(-12))
        for _ in range(3): D.sgemm(a, b, out=c)
        D.sync(); t2 = Timer(); t2.start()
        for _ in range(reps): D.sgemm(a, b, out=c)
        t2.stop(); check(lib.np_sgemm_set_variant(-13))
        extra = "   tiled kernels: %.3f ms" % (t2.elapsed_ms() / reps)
    print("%5d x %5d x %5d : %8.3f ms  %6.1f TF  (operands %5.1f MB -> %5.0f GB/s)" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9,
          4e-6 * (m * k + k * n + m * n), 4e-6 * (m * k + k * n + m * n) / ms) + extra, flush=True)
    a.free(); b.free(); c.free()
