"""Device-side initializers at the reference's phpbench shapes and larger (benchmarks/initializers/*:
full / zeros / ones / identity / arange) — 4 B/elem of HBM writes.  Usage: python tools/init_sweep.py"""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import Timer, check, load
D.init(0); lib = load(); t = Timer()
out = D.DeviceArray((100_000_000,))


def run(name, fn, n):
    for _ in range(3): fn()
    D.sync(); t.start(); reps = 10
    for _ in range(reps): fn()
    t.stop(); ms = t.elapsed_ms() / reps
    D.sync(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    D.sync(); wall = (time.perf_counter() - t0) / reps * 1e3
    print("  %-34s %8.4f ms (wall %8.4f ms) %7.0f GB/s" % (name, ms, wall, 4.0 * n / ms / 1e6), flush=True)


for shape in ((100, 1, 1), (1000, 1, 1), (500, 1000, 1), (1000, 10000, 1), (10000, 10000, 1)):
    n = shape[0] * shape[1] * shape[2]
    run("full %s" % (shape,), lambda: check(lib.np_fill(out.ptr, 4.0, n)), n)
    run("zeros %s" % (shape,), lambda: check(lib.np_memset0(out.ptr, 4 * n)), n)
for size in (100, 1000, 10000):
    run("identity %d" % size, lambda: check(lib.np_identity(out.ptr, size)), size * size)
for n, start, step in ((1000, 0.0, 1.0), (1_000_000, 0.0, 1.0), (100_000_000, 0.0, 1.0), (100_000_000, -5.0, 1e-7), (50_000_000, 1.0, 0.1)):
    run("arange n=%d step=%g" % (n, step), lambda: check(lib.np_arange(out.ptr, start, step, n)), n)
