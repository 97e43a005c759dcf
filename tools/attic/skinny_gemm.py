"""np_sgemm on skinny shapes (tiny N and / or tiny M): point transforms, classifier layers, X^T X of an N x 3
array.  Usage: python tools/skinny_gemm.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer
D.init(0); t = Timer()
for (m, n, k) in [(10_000_000, 3, 3), (1_000_000, 16, 16), (1_000_000, 8, 64), (100_000, 10, 784), (100_000, 784, 10), (1_000_000, 4, 4), (3, 3, 10_000_000),
                  (65536, 128, 128), (10, 10, 10), (1_000_000, 1, 64), (4096, 4096, 1), (4096, 4096, 2), (4096, 4096, 8),
                  (3, 10_000_000, 3), (4, 4_000_000, 4), (10, 1_000_000, 16), (1, 10_000_000, 8), (16, 1_000_000, 64),
                  (1_000_000, 10, 128), (1_000_000, 32, 256), (250_000, 24, 1024), (4_000_000, 5, 32), (100_000, 12, 4096), (2049, 31, 36), (100_000, 10, 785), (1_000_000, 16, 30),
                  (784, 10, 200_000), (784, 10, 1_000_000), (256, 32, 500_000), (1000, 3, 100_000), (128, 64, 500_000), (32, 64, 2_000_000), (64, 64, 1_000_000), (512, 48, 100_000), (32, 32, 2_000_000), (24, 10, 2_000_000), (128, 64, 500_000), (100, 20, 300_000)]:
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n)); D.fill(a, 0.5); D.fill(b, 0.25)
    for _ in range(2): D.sgemm(a, b, out=c)
    D.sync(); t.start(); reps = 5
    for _ in range(reps): D.sgemm(a, b, out=c)
    t.stop(); ms = t.elapsed_ms() / reps
    nbytes = 4.0 * (m * k + k * n + m * n)
    print("  %9d x %4d x %8d : %8.3f ms %7.1f TFLOP/s %7.0f GB/s" % (m, n, k, ms, 2.0 * m * n * k / ms / 1e9, nbytes / ms / 1e6), flush=True)
    a.free(); b.free(); c.free()
