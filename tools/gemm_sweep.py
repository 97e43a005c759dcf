"""Dev tool: np_sgemm throughput across shapes: whole-K plans only (variant -1: one tile per workgroup), the planner
with K-splitting tails but no stream-K (variant -5), and the default planner (variant -2: + stream-K where its model
says so).  NP_SWEEP_ODD=1: shapes whose rows are not float4-loadable (odd K / N), the default planner with the
pad-copy path they took before round 3 (np_sgemm_set_variant(-6)), the LDS-DMA kernel reading them as they are whatever
the size (-8), and the default (-7: as they are from a size threshold up, and where the model prefers it to the copies)."""
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
odd = bool(os.environ.get('NP_SWEEP_ODD'))
if odd:
    shapes = [(1001,)*3, (1537,)*3, (2001,)*3, (2049,)*3, (3001,)*3, (4097,)*3, (5001,)*3, (4096, 4096, 4097), (4096, 4097, 4096), (4097, 4096, 4096),
              (2049, 1003, 3001), (8191,)*3, (1000, 1002, 1000), (3000, 3001, 3000)]
for (m, n, k) in ([] if os.environ.get('NP_SWEEP_PEEL') else shapes):
    a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
    D.fill(a, 0.5); D.fill(b, 0.25)
    D.unary("sin", a, out=a); D.unary("cos", b, out=b)   # non-constant data
    reps = max(3, min(50, int(2e11 / (2.0 * m * n * k))))
    line = "%6d x %6d x %6d :" % (m, n, k)
    for variant, label in (((-6, "pad-copy path"), (-8, "as they are"), (-7, "default")) if odd else ((-1, "whole-K"), (-5, "no-SK"), (-4, "SK"), (-2, "default"))):
        check(lib.np_sgemm_set_variant(variant))
        for _ in range(3): D.sgemm(a, b, out=c)
        D.sync(); t = Timer(); t.start()
        for _ in range(reps): D.sgemm(a, b, out=c)
        t.stop(); ms = t.elapsed_ms() / reps
        line += "  %s %7.3f ms %5.1f TF" % (label, ms, 2.0 * m * n * k / ms / 1e9)
    check(lib.np_sgemm_set_variant(-7))
    check(lib.np_sgemm_set_variant(-2))
    print(line, flush=True)
    a.free(); b.free(); c.free()
if os.environ.get('NP_SWEEP_PEEL'):   # thin ragged edges: the whole product (-9) / the edges peeled off whenever thin enough (-11) / default (-10)
    for (m, n, k) in [(4097,)*3, (4097, 4096, 4096), (4096, 4097, 4096), (4098, 4098, 4098), (4104, 4097, 4097), (2049,)*3, (3073,)*3, (5121,)*3, (6145,)*3,
                      (8193,)*3, (8200, 8193, 8192), (4097, 4097, 1024), (2305, 8193, 4096), (4112, 4096, 4096), (4128, 4097, 4097), (4120, 8192, 4096)]:
        a = D.DeviceArray((m, k)); b = D.DeviceArray((k, n)); c = D.DeviceArray((m, n))
        D.fill(a, 0.5); D.fill(b, 0.25)
        D.unary("sin", a, out=a); D.unary("cos", b, out=b)
        reps = max(3, min(50, int(2e11 / (2.0 * m * n * k))))
        line = "%6d x %6d x %6d :" % (m, n, k)
        for variant, label in ((-9, "whole"), (-11, "peeled"), (-10, "default")):
            check(lib.np_sgemm_set_variant(variant))
            for _ in range(3): D.sgemm(a, b, out=c)
            D.sync(); t = Timer(); t.start()
            for _ in range(reps): D.sgemm(a, b, out=c)
            t.stop(); ms = t.elapsed_ms() / reps
            line += "  %s %7.3f ms %5.1f TF" % (label, ms, 2.0 * m * n * k / ms / 1e9)
        check(lib.np_sgemm_set_variant(-10))
        print(line, flush=True)
        a.free(); b.free(); c.free()
    sys.exit(0)
if odd:   # batched, every matrix at an odd offset: before, these ran on the register-staged kernels (no pad path for batches)
    for (batch, m, n, k) in ((16, 1001, 1001, 1001), (64, 513, 515, 517), (8, 2049, 2049, 2049), (32, 1000, 1002, 1000)):
        a = D.DeviceArray((batch, m, k)); b = D.DeviceArray((batch, k, n)); c = D.DeviceArray((batch, m, n))
        D.fill(a, 0.5); D.fill(b, 0.25)
        D.unary("sin", a, out=a); D.unary("cos", b, out=b)
        line = "%3d x (%5d x %5d x %5d) :" % (batch, m, n, k)
        for variant, label in ((-6, "register-staged"), (-8, "LDS-DMA as they are"), (-7, "default")):
            check(lib.np_sgemm_set_variant(variant))
            run = lambda: check(lib.np_sgemm_strided_batched(batch, m, n, k, a.ptr, m * k, b.ptr, k * n, c.ptr, m * n))
            for _ in range(3): run()
            D.sync(); t = Timer(); t.start()
            for _ in range(10): run()
            t.stop(); ms = t.elapsed_ms() / 10
            line += "  %s %7.3f ms %5.1f TF" % (label, ms, 2.0 * batch * m * n * k / ms / 1e9)
        check(lib.np_sgemm_set_variant(-7))
        print(line, flush=True)
        a.free(); b.free(); c.free()
