"""Dev tool: np_sgemm throughput across shapes: whole-K plans only (variant -1: one tile per workgroup), the planner
with K-splitting tails but no stream-K (variant -5), and the default planner (variant -2: + stream-K where its model
says so)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import load, Timer, check
D.init(0); lib = load()
import os
shapes = [(512,)*3, (1024,)*3, (1536,)*3, (2048,)*3, (3072,)*3, (4096,)*3, (6144,)*3, (8192,)*3, (1000,)*3, (2000,)*3, (4000,)*3, (4097,)*3,
          (4096, 4096, 256), (256, 4096, 4096), (4096, 256, 4096), (8192, 8192, 512), (16384, 1024, 1024), (100, 100, 100000),
          (2560,)*3, (3584,)*3, (5120,)*3, (7168,)*3, (3000,)*3, (1280, 1280, 8192), (768, 768, 768), (2304, 2304, 4096)]
if os.environ.get('NP_SWEEP_SHORT'):
    shapes = [(1536,)*3, (2048,)*3, (3072,)*3, (4096,)*3, (2000,)*3, (4097,)*3, (2560,)*3, (3584,)*3, (3000,)*3, (5120,)*3, (6144,)*3, (2304, 2304, 4096), (1280, 1280, 8192), (4000,)*3]
for (m, n, k) in shapes:
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    D.unary("sin", a, out=a); D.unary("cos", b, out=b)   # non-constant data
    reps = max(3, min(50, int(2e11 / (2.0 * m * n * k))))
    line = "%6d x %6d x %6d :" % (m, n, k)
    for variant, label in ((-1, "whole-K"), (-5, "no-SK"), (-4, "SK"), (-2, "default")):
        check(lib.np_sgemm_set_variant(variant))
        for _ in range(3): D.sgemm(a, b, out=c)
        D.sync(); t = Timer(); t.start()
        for _ in range(reps): D.sgemm(a, b, out=c)
        t.stop(); ms = t.elapsed_ms() / reps
        line += "  %s %7.3f ms %5.1f TF" % (label, ms, 2.0 * m * n * k / ms / 1e9)
    check(lib.np_sgemm_set_variant(-2))
    print(line, flush=True)
    a.free(); b.free(); c.free()
