"""Gram / covariance shapes: (d x n) . (n x d) with n >> d, as matmul(transpose(X), X) reaches np_sgemm.
Usage: python tools/gram_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer
D.init(0); t = Timer()
for d, n in ((64, 1_000_000), (128, 1_000_000), (256, 500_000), (512, 200_000), (784, 200_000), (1024, 100_000), (2048, 50_000), (100, 1_000_000), (300, 300_000)):
    a = D.DeviceArray((d, n)); b = D.DeviceArray((n, d)); c = D.DeviceArray((d, d)); D.fill(a, 0.5); D.fill(b, 0.25)
    for _ in range(2): D.sgemm(a, b, out=c)
    D.sync(); t.start()
    for _ in range(5): D.sgemm(a, b, out=c)
    t.stop(); ms = t.elapsed_ms() / 5
    print("  %5d x %5d x %8d : %8.3f ms %7.1f TFLOP/s %7.0f GB/s of operand reads" % (d, d, n, ms, 2.0 * d * d * n / ms / 1e9, 8.0 * d * n / ms / 1e6), flush=True)
    a.free(); b.free(); c.free()
