"""Dev tool: the mid-size tile family (sgemm_dmas_kernel: 128x128 / 128x64 / 64x64 LDS-DMA tiles, K split S ways inside
the launch with a distributed fold) against the default planner, per shape: time, TFLOP/s and the largest difference from
the default plan's result in units of |A|.|B| (both are fp32 products of the same inputs; ~1e-7 is rounding).
np_sgemm_set_variant(-(1000 + 100 * shape + S)) forces the form; shapes/S that do not apply fall back and print '-'.
Usage: python tools/gemm_mid_sweep.py [short]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import load, Timer, check

D.init(0)
lib = load()
shapes = [(512,) * 3, (768,) * 3, (1000,) * 3, (1024,) * 3, (1280,) * 3, (1536,) * 3, (2000,) * 3, (2048,) * 3,
          (256, 4096, 4096), (4096, 4096, 256), (4096, 256, 4096), (1024, 1024, 4096), (2048, 2048, 512), (1001, 1003, 1002),
          (640, 640, 640), (896, 896, 896), (1152, 1152, 1152), (512, 512, 4096), (3072, 3072, 3072)]
if len(sys.argv) > 1 and sys.argv[1] == "plans":
    shapes += [(4096,) * 3, (2560,) * 3, (4000,) * 3, (4097,) * 3, (8192, 8192, 512), (16384, 1024, 1024), (1280, 1280, 8192), (100, 100, 100000)]
if len(sys.argv) > 1 and sys.argv[1] == "kdeep":
    shapes = [(256, 4096, 4096), (4096, 256, 4096), (1024, 1024, 4096), (512, 512, 4096), (512, 1024, 8192), (1024, 1024, 2048), (768, 768, 3072), (512,) * 3, (1024,) * 3, (2048, 2048, 512)]
if len(sys.argv) > 1 and sys.argv[1] in ("swizzle", "waves"):
    shapes = [(512,) * 3, (640,) * 3, (1001, 1003, 1002)] + [(768,) * 3, (1000,) * 3, (1024,) * 3, (1280,) * 3, (2048,) * 3, (256, 4096, 4096), (4096, 256, 4096), (1024, 1024, 4096), (2048, 2048, 512), (4096, 4096, 256)]
if len(sys.argv) > 1 and sys.argv[1] == "short":
    shapes = [(768,) * 3, (1000,) * 3, (1024,) * 3, (1536,) * 3, (256, 4096, 4096), (4096, 4096, 256)]
NAMES = ["128x128", "128x64", "64x64"]
t = Timer()


def run(a, b, c, reps, warm_s=0.08):
    # the same product for warm_s first: a hundred 20-us launches after an idle moment run on a clock that has not come up
    # (up to 8 % slower; profiles/r04/gemm_kdeep_ab.log against gemm_plans2.log)
    import time
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


for (m, n, k) in shapes:
    A = synth.uniform((m, k), 31, -1.0, 1.0)
    B = synth.uniform((k, n), 32, -1.0, 1.0)
    a, b, c = D.DeviceArray.from_host(A), D.DeviceArray.from_host(B), D.DeviceArray((m, n))
    reps = max(5, min(100, int(4e10 / (2.0 * m * n * k))))
    flop = 2.0 * m * n * k
    check(lib.np_sgemm_set_variant(-999))
    # the two planners in alternation (the first measurement after an upload runs on a colder clock: 1024^3 measured 8 % slower
    # as "default" than as the same plan forced a moment later), medians of four rounds
    r03s, news = [], []
    for _ in range(4):
        check(lib.np_sgemm_set_variant(-14))      # the planner of round 3: no mid-size LDS-DMA tiles
        r03s.append(run(a, b, c, reps))
        check(lib.np_sgemm_set_variant(-15))
        news.append(run(a, b, c, reps))
    ms_r03, ms = float(np.median(r03s)), float(np.median(news))
    ref = c.to_host().astype(np.float64)
    scale = float(np.abs(A[:64]).astype(np.float64).sum(1).max()) * float(np.abs(B).max())   # a cheap bound on |A|.|B| per element
    print("%5d x %5d x %5d  default %7.1f us %6.1f TF   (round-3 planner %7.1f us %6.1f TF)" % (
        m, n, k, ms * 1e3, flop / ms / 1e9, ms_r03 * 1e3, flop / ms_r03 / 1e9), flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "plans":      # the planner's choice only
        for d in (a, b, c):
            d.free()
        continue
    if len(sys.argv) > 1 and sys.argv[1] == "kdeep":       # candidates for few-tile, deep-K products in alternation (medians of 4 rounds)
        cands = [("round-3 planner", -14, None), ("64x64 S1", -15, (2, 1)), ("128x64 S2", -15, (1, 2)), ("128x128 S4", -15, (0, 4)), ("64x64 S2", -15, (2, 2))]
        got = {name: [] for name, _, _ in cands}
        for _ in range(4):
            for name, planner, forced in cands:
                check(lib.np_sgemm_set_variant(-999))
                check(lib.np_sgemm_set_variant(planner))
                if forced:
                    check(lib.np_sgemm_set_variant(-(1000 + 100 * forced[0] + forced[1])))
                got[name].append(run(a, b, c, reps))
        check(lib.np_sgemm_set_variant(-999))
        check(lib.np_sgemm_set_variant(-15))
        print("      " + "   ".join("%s %.1f us (%.1f TF)" % (name, np.median(v) * 1e3, flop / np.median(v) / 1e9) for name, v in got.items()), flush=True)
        for d in (a, b, c):
            d.free()
        continue
    if len(sys.argv) > 1 and sys.argv[1] == "waves":       # 64x64 tiles: 4 waves (shape 2) against 8 (two per tile position, k-groups split: shape 5)
        for shape in (2,):
            for S in (1, 2, 4):
                line = "      %-8s S%d" % (NAMES[shape], S)
                for sh in (shape, shape + 3, shape, shape + 3):
                    check(lib.np_sgemm_set_variant(-(1000 + 100 * sh + S)))
                    D.fill(c, float("nan"))
                    ms = run(a, b, c, reps)
                    got = c.to_host().astype(np.float64)
                    err = float(np.abs(got - ref).max()) / scale if not np.isnan(got).any() else float("nan")
                    line += "  %s %6.1f us %5.1f TF (%.0e)" % ("4 waves" if sh == shape else "8 waves", ms * 1e3, flop / ms / 1e9, err)
                print(line, flush=True)
        check(lib.np_sgemm_set_variant(-999))
        for d in (a, b, c):
            d.free()
        continue
    if len(sys.argv) > 1 and sys.argv[1] == "swizzle":     # whole-K mid tiles: row-major tile order against XCD-aware bands
        for shape in range(3):
            line = "      %-8s" % NAMES[shape]
            for sw in (-16, -17, -16, -17):
                check(lib.np_sgemm_set_variant(sw))
                check(lib.np_sgemm_set_variant(-(1000 + 100 * shape + 1)))
                ms = run(a, b, c, reps)
                line += "  %s %6.1f us %5.1f TF" % ("row-major" if sw == -16 else "bands    ", ms * 1e3, flop / ms / 1e9)
            print(line, flush=True)
        check(lib.np_sgemm_set_variant(-17))
        check(lib.np_sgemm_set_variant(-999))
        for d in (a, b, c):
            d.free()
        continue
    best = (ms, "default")
    for shape in range(3):
        line = "      %-8s" % NAMES[shape]
        for S in (1, 2, 4, 8, 16):
            check(lib.np_sgemm_set_variant(-(1000 + 100 * shape + S)))
            D.fill(c, float("nan"))
            ms = run(a, b, c, reps)
            got = c.to_host().astype(np.float64)
            err = float(np.abs(got - ref).max()) / scale if not np.isnan(got).any() else float("nan")
            line += "  S%-2d %6.1f us %5.1f TF (%.0e)" % (S, ms * 1e3, flop / ms / 1e9, err)
            if ms < best[0] and err == err and err < 1e-5:
                best = (ms, "%s S=%d" % (NAMES[shape], S))
        print(line, flush=True)
    check(lib.np_sgemm_set_variant(-999))
    print("      best: %s  %.1f us  %.1f TF" % (best[1], best[0] * 1e3, flop / best[0] / 1e9), flush=True)
    for d in (a, b, c):
        d.free()
