"""Dev tool: np_sgemm throughput across shapes: whole-K plans only (variant -1) vs the default
planner with K-splitting tails (variant -2)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import load, Timer, check
D.init(0); lib = load()
shapes = [(512,)*3, (1024,)*3, (1536,)*3, (2048,)*3, (3072,)*3, (4096,)*3, (6144,)*3, (8192,)*3, (1000,)*3, (2000,)*3, (4000,)*3, (4097,)*3,
          (4096, 4096, 256), (256, 4096, 4096), (4096, 256, 4096), (8192, 8192, 512), (16384, 1024, 1024), (100, 100, 100000),
          (2560,)*3, (3584,)*3, (5120,)*3, (7168,)*3, (3000,)*3, (1280, 1280, 8192), (768, 768, 768), (2304, 2304, 4096)]
for (m, n, k) in shapes:
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    D.unary("sin", a, out=a); D.unary("cos", b, out=b)   # non-constant data
    reps = max(3, min(50, int(2e11 / (2.0 * m * n * k))))
    line = "%6d x %6d x %6d :" % (m, n, k)
    for variant, label in ((-1, "whole-K"), (-2, "planner")):
        check(lib.np_sgemm_set_variant(variant))
        for _ in range(3): D.sgemm(a, b, out=c)
        D.sync(); t = Timer(); t.start()
        for _ in range(reps): D.sgemm(a, b, out=c)
        t.stop(); ms = t.elapsed_ms() / reps
        line += "  %s %8.3f ms %6.1f TFLOP/s" % (label, ms, 2.0 * m * n * k / ms / 1e9)
    print(line, flush=True)
    a.free(); b.free(); c.free()
