"""Dev tool: what the GEMM planner would run, per shape, on a 256-CU device — host arithmetic only (np_sgemm_debug_plan with
cus = 256), so it runs in the build container.  Columns: the chosen plan and its modelled time, stream-K's model, the best
mid-size-tile plan, the best plan without them.  Compare with profiles/r04/gemm_mid_sweep_forced.log (measured, forced forms).
Usage: python tools/gemm_plan_model.py [M N K ...]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd._lib import check, load

lib = load()
CFG = {0: "256x128 dma", 1: "128x128 reg", 2: "64x64 reg", 3: "128x128 mid", 4: "128x64 mid", 5: "64x64 mid", -1: "-"}
for _i, (_tm, _tn) in enumerate([(3, 3), (2, 2), (4, 4), (3, 2), (4, 2), (4, 3), (5, 3)]):
    CFG[6 + _i] = "%dx%d kq" % (16 * _tm, 16 * _tn)
shapes = [(256,) * 3, (384,) * 3, (512,) * 3, (576,) * 3, (640,) * 3, (704,) * 3, (768,) * 3, (832,) * 3, (896,) * 3, (1000,) * 3, (1024,) * 3, (1152,) * 3, (1280,) * 3, (1536,) * 3, (2000,) * 3, (2048,) * 3,
          (2560,) * 3, (3072,) * 3, (4096,) * 3, (256, 4096, 4096), (4096, 4096, 256), (4096, 256, 4096), (1024, 1024, 4096), (2048, 2048, 512),
          (512, 512, 4096), (768, 768, 3072), (768, 768, 256), (512, 1024, 512), (8192, 8192, 512), (16384, 1024, 1024), (1280, 1280, 8192), (100, 100, 100000), (1001, 1003, 1002)]
if len(sys.argv) > 3:
    v = [int(x) for x in sys.argv[1:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v) - 2, 3)]
out = (C.c_double * 11)()
for (m, n, k) in shapes:
    check(lib.np_sgemm_debug_plan(m, n, k, 1, 256, out))
    o = list(out)
    f = lambda t: "%7.1f" % t if t < 1e299 else "      -"
    print("%5d x %5d x %6d  -> %-12s tail %3d S %3d %s us%s   stream-K %s   mid: %-12s S %2d %s   other: %-12s %s" % (
        m, n, k, CFG[int(o[0])], int(o[1]), int(o[2]), f(o[3]), "  (stream-K taken)" if o[4] else "", f(o[5]),
        CFG[int(o[6])], int(o[7]), f(o[8]), CFG[int(o[9])], f(o[10])))
