"""Which k-quartered tile shape and how many K-chunks for a deep-K product of a few tiles?  np_sgemm_set_variant(-(30000 + 1000 *
shape + S)) forces single products onto S chunks of shape `shape` (0: 48 x 48, 1: 32 x 32, 2: 64 x 64, 3: 48 x 32, 4: 64 x 32, 5: 64 x 48);
the fold (np_reduce_axis over the chunks) is timed alone beside it.  Usage: python tools/gemm_deep_k_sweep.py"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
t = Timer()
warm = D.DeviceArray.from_host(synth.uniform((2048, 2048), 1, -1, 1))
wc = D.DeviceArray((2048, 2048))
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.5:
    for _ in range(20):
        D.sgemm(warm, warm, out=wc)
    D.sync()
SHAPES = ((100, 100, 100000), (128, 128, 65536), (200, 200, 50000), (256, 256, 32768), (100, 100, 10000), (300, 300, 20000))
SS = (8, 16, 32, 64, 128, 256)
NAMES = {0: "48x48", 1: "32x32", 2: "64x64", 3: "48x32", 4: "64x32", 5: "64x48"}


def timed(fn, reps):
    for _ in range(5):
        fn()
    D.sync()
    best = 1e9
    for _ in range(3):
        t.start()
        for _ in range(reps):
            fn()
        t.stop()
        best = min(best, t.elapsed_ms() / reps * 1e3)
    return best


for (m, n, k) in SHAPES:
    a = D.DeviceArray.from_host(synth.uniform((m, k), 31, -1.0, 1.0))
    b = D.DeviceArray.from_host(synth.uniform((k, n), 32, -1.0, 1.0))
    c = D.DeviceArray((m, n))
    reps = max(10, min(100, int(5e10 / (2.0 * m * n * k))))
    check(lib.np_sgemm_set_variant(-30000))
    check(lib.np_sgemm_set_variant(-22))
    print("%d x %d x %d   old plan %.1f us" % (m, n, k, timed(lambda: D.sgemm(a, b, out=c), reps)), flush=True)
    check(lib.np_sgemm_set_variant(-23))
    w = D.DeviceArray((256, m * n))
    folds = []
    for S in SS:
        folds.append(timed(lambda: check(lib.np_reduce_axis(0, w.ptr, 1, S, m * n, c.ptr, 0)), reps))
    print("   fold alone " + "  ".join("S=%-3d %5.1f" % (S, f) for S, f in zip(SS, folds)))
    for shape in (2, 0, 1, 4, 5):
        for major in (-25, -26):   # tile-major / chunk-major over the XCDs
            check(lib.np_sgemm_set_variant(major))
            row = []
            for S in SS:
                check(lib.np_sgemm_set_variant(-(30000 + 1000 * shape + S)))
                row.append(timed(lambda: D.sgemm(a, b, out=c), reps))
            print("   %s %s " % (NAMES[shape], "tile " if major == -25 else "chunk") + "  ".join("S=%-3d %5.1f" % (S, u) for S, u in zip(SS, row)), flush=True)
    check(lib.np_sgemm_set_variant(-30000))
    check(lib.np_sgemm_set_variant(-26))
    for d in (a, b, c, w):
        d.free()
