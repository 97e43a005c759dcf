"""np_sgemm_strided_batched over batch x matrix-size combinations.  Usage: python tools/batched_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import load, Timer, check
D.init(0); lib = load(); t = Timer()
for (batch, m, n, k) in [(512, 1024, 1024, 1024), (100, 512, 512, 512), (1000, 128, 128, 128), (4096, 64, 64, 64), (10000, 32, 32, 32),
                         (100000, 8, 8, 8), (64, 2048, 2048, 2048), (8, 4096, 4096, 64), (2000, 100, 100, 100), (50, 1000, 1000, 1000),
                         (10000, 16, 256, 16), (3, 4096, 4096, 4096)]:
    a = D.DeviceArray((batch, m, k)); b = D.DeviceArray((batch, k, n)); c = D.DeviceArray((batch, m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)

    def f():
        check(lib.np_sgemm_strided_batched(batch, m, n, k, a.ptr, m * k, b.ptr, k * n, c.ptr, m * n))
    for _ in range(2): f()
    D.sync(); t.start(); reps = 5
    for _ in range(reps): f()
    t.stop(); ms = t.elapsed_ms() / reps
    flop = 2.0 * batch * m * n * k
    nbytes = 4.0 * batch * (m * k + k * n + m * n)
    print("%6d x (%4d x %4d x %4d) : %8.3f ms %7.1f TFLOP/s  %6.0f GB/s" % (batch, m, n, k, ms, flop / ms / 1e9, nbytes / ms / 1e6), flush=True)
    a.free(); b.free(); c.free()
