"""A/B of the split-K path of np_sgemm (small result, long inner dimension)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import load, Timer, check
D.init(0); lib = load()
shapes = [(100, 100, 100000), (64, 64, 65536), (128, 128, 16384), (256, 256, 8192), (500, 300, 20000),
          (32, 1000, 50000), (1000, 8, 4096), (512, 512, 512), (1024, 1024, 1024), (16, 16, 1000000)]
for (m, n, k) in shapes:
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    D.unary("sin", a, out=a); D.unary("cos", b, out=b)
    line = "%6d x %6d x %8d :" % (m, n, k)
    for variant, label in ((-1, "plain"), (-2, "split-K")):
        check(lib.np_sgemm_set_variant(variant))
        reps = max(3, min(50, int(5e10 / (2.0 * m * n * k))))
        for _ in range(3): D.sgemm(a, b, out=c)
        D.sync(); t = Timer(); t.start()
        for _ in range(reps): D.sgemm(a, b, out=c)
        t.stop(); ms = t.elapsed_ms() / reps
        line += "  %s %8.3f ms %6.1f TFLOP/s" % (label, ms, 2.0 * m * n * k / ms / 1e9)
    print(line, flush=True)
    a.free(); b.free(); c.free()
