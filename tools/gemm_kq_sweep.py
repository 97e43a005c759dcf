"""Dev tool: sgemm_kq_kernel (one tile per workgroup, the four waves split K; np_sgemm.hip) forced through
np_sgemm_set_variant(-(2000 + shape)) against the default planner, in alternation behind a warm-up: time, TFLOP/s and the
largest difference from the fp64 product in units of |A|.|B|.  Shapes: 0 = 48x48, 1 = 32x32; the default planner (which takes them where its
model says so) is the first column.
Usage: python tools/gemm_kq_sweep.py [check]     (check: small and ragged shapes, correctness only)"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import load, Timer, check

D.init(0)
lib = load()
t = Timer()
NAMES = ["48x48", "32x32", "64x64", "48x32", "64x32", "64x48", "80x48"]
check_only = len(sys.argv) > 1 and sys.argv[1] == "check"
shapes = [(512,) * 3, (576,) * 3, (640,) * 3, (704,) * 3, (768,) * 3, (832,) * 3, (896,) * 3, (960,) * 3, (1024,) * 3, (1088,) * 3, (1216,) * 3, (1344,) * 3, (1600,) * 3, (1920,) * 3, (2560,) * 3, (768, 768, 3072), (384, 384, 384),
          (256, 256, 256), (768, 768, 256), (512, 1024, 512), (600,) * 3, (700,) * 3, (760,) * 3, (500,) * 3, (128, 4096, 4096),
          (1000,) * 3, (1152,) * 3, (1280,) * 3, (1536,) * 3, (2000,) * 3, (2048,) * 3, (256, 4096, 4096), (4096, 4096, 256), (1024, 1024, 4096)]
if check_only:
    shapes = [(48, 48, 16), (48, 48, 64), (50, 72, 80), (100, 200, 64), (16, 16, 16), (1, 4, 16), (97, 132, 208), (720, 720, 720), (333, 444, 176), (64, 64, 1024), (130, 68, 4096)]


def run(a, b, c, reps, warm_s=0.08):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < warm_s:
        for _ in range(50):
            D.sgemm(a, b, out=c)
        D.sync()
    t.start()
    for _ in range(reps):
        D.sgemm(a, b, out=c)
    t.stop()
    return t.elapsed_ms() / reps


bad = 0
for (m, n, k) in shapes:
    A = synth.uniform((m, k), 31, -1.0, 1.0)
    B = synth.uniform((k, n), 32, -1.0, 1.0)
    a, b, c = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((m, n))
    want = A.astype(np.float64) @ B.astype(np.float64)
    scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
    flop = 2.0 * m * n * k
    reps = max(5, min(200, int(4e10 / flop)))
    line = "%5d x %5d x %5d " % (m, n, k)
    forms = [("default", -999)] + [(NAMES[s], -(2000 + s)) for s in range(len(NAMES))]
    times = {name: [] for name, _ in forms}
    errs = {}
    for rnd in range(1 if check_only else 3):
        for name, code in forms:
            check(lib.np_sgemm_set_variant(-999))
            check(lib.np_sgemm_set_variant(code))
            if rnd == 0:
                D.fill(c, float("nan"))
                D.sgemm(a, b, out=c)
                got = c.to_host().astype(np.float64)
                errs[name] = float((np.abs(got - want) / scale).max()) if not np.isnan(got).any() else float("nan")
                if not errs[name] <= 1e-6:
                    bad += 1
            if not check_only:
                times[name].append(run(a, b, c, reps))
    check(lib.np_sgemm_set_variant(-999))
    best = None
    for name, _ in forms:
        if check_only:
            line += "  %s %.0e" % (name, errs[name])
        else:
            ms = float(np.median(times[name]))
            line += "  %s %.1f" % (name, ms * 1e3)
            if name != "default" and (best is None or ms < best[0]):
                best = (ms, name)
    if not check_only:
        d = float(np.median(times["default"]))
        line += "   | default %.1f us %.1f TF, best forced %s %.1f us %.1f TF" % (d * 1e3, flop / d / 1e9, best[1], best[0] * 1e3, flop / best[0] / 1e9)
    print(line, flush=True)
    for d in (a, b, c):
        d.free()
print("bad:", bad)
assert lib.np_sync() == 0
