"""Layout kernels only (transpose2d / permute / strided copy), for A/B of two builds on one box:
    python tools/layout_sweep.py                      the shipped library
    NP_HIP_LIB=build/ab/libnp_hip_r03layout.so python tools/layout_sweep.py     another build of libnp_hip.so
Every case is also checked against numpy (bit-exact) on a down-scaled shape where the full one is large."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load

D.init(0)
lib = load()
N = 100_000_000
big, out = D.DeviceArray((N,)), D.DeviceArray((N,))
D.fill(big, 1.5)
t = Timer()


def run(fn, reps=10):
    for _ in range(2):
        fn()
    D.sync()
    t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps


def line(name, ms, nbytes):
    print("  %-52s %8.3f ms  %6.0f GB/s" % (name, ms, nbytes / ms / 1e6), flush=True)


for rows, cols in ((8192, 8192), (8191, 8193), (4099, 4099), (100_000, 1000), (1000, 100_000), (65536, 1500), (12345, 6789), (10_000_000, 3)):
    n = rows * cols
    line("transpose2d %dx%d" % (rows, cols), run(lambda: check(lib.np_transpose2d(big.ptr, out.ptr, 1, rows, cols))), 8.0 * n)
CASES = (((64, 128, 1024, 8), (0, 2, 1, 3)), ((256, 512, 512), (2, 1, 0)), ((256, 512, 512), (1, 0, 2)), ((100, 100, 100, 100), (3, 2, 1, 0)),
         ((30, 3, 1024, 1024), (0, 2, 3, 1)), ((30, 1024, 1024, 3), (0, 3, 1, 2)), ((1000, 300, 300), (2, 0, 1)), ((50, 60, 70, 80), (1, 3, 0, 2)),
         ((4096, 4, 4096), (2, 1, 0)), ((128, 128, 128, 16), (2, 1, 0, 3)), ((200, 33, 77, 41), (3, 1, 2, 0)), ((1000, 1000, 5, 20), (3, 2, 1, 0)),
         ((300, 300, 300), (1, 2, 0)), ((16, 500, 500, 4), (0, 2, 1, 3)))
import os
if os.environ.get("NP_LAYOUT_VARIANT"):
    check(lib.np_layout_set_variant(int(os.environ["NP_LAYOUT_VARIANT"])))
for shape, perm in CASES:
    n = int(np.prod(shape))
    sh = (C.c_int * len(shape))(*shape)
    pm = (C.c_int * len(perm))(*perm)
    line("permute %s %s" % (shape, perm), run(lambda: check(lib.np_permute(big.ptr, out.ptr, len(shape), sh, pm))), 8.0 * n)
# correctness of the same permutations on small shapes of the same structure
for shape, perm in CASES + (((7, 9, 11), (2, 1, 0)), ((5, 6, 7, 3), (0, 2, 1, 3)), ((3, 130, 70), (2, 0, 1)), ((2, 3, 65, 67), (3, 2, 1, 0))):
    small = tuple(max(2, min(s, 37 + 3 * i)) for i, s in enumerate(shape))
    h = synth.uniform(small, 3, -1.0, 1.0)
    d = D.DeviceArray.from_host(h)
    o = D.DeviceArray((h.size,))
    sh = (C.c_int * len(small))(*small)
    pm = (C.c_int * len(perm))(*perm)
    check(lib.np_permute(d.ptr, o.ptr, len(small), sh, pm))
    ok = (o.to_host().reshape(-1) == np.ascontiguousarray(h.transpose(perm)).reshape(-1)).all()
    if not ok:
        print("  MISMATCH permute %s %s" % (small, perm), flush=True)
    d.free()
    o.free()
print("  permutations checked against numpy", flush=True)
