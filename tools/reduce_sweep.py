"""Throughput of the reductions across shapes / axes (HBM roofline view, 4 B/elem read).
Usage: python tools/reduce_sweep.py"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import Timer
D.init(0)
t = Timer()


def run(fn):
    for _ in range(2): fn()
    D.sync(); t.start()
    reps = 10
    for _ in range(reps): fn()
    t.stop()
    return t.elapsed_ms() / reps


print("full reductions, 10^8 elements")
x = D.DeviceArray((100_000_000,)); D.fill(x, 1.0)
for op in ("sum", "prod", "min", "max", "mean"):
    ms = run(lambda: D.reduce_all(op, x))
    print("  %-5s %7.3f ms %6.0f GB/s (incl. D2H of the result)" % (op, ms, 0.4 / ms * 1e3))
x.free()
print("axis reductions")
CASES = [((65536, 4096), 0), ((65536, 4096), 1), ((4096, 65536), 0), ((4096, 65536), 1), ((1000, 1000), 0), ((1000, 1000), 1),
         ((256, 1024, 1024), 1), ((256, 1024, 1024), 0), ((256, 1024, 1024), 2), ((30_000_000, 3), 0), ((30_000_000, 3), 1),
         ((3, 30_000_000), 0), ((3, 30_000_000), 1), ((100_000_000, 1), 0), ((16, 16, 16, 16, 16, 64), 3), ((10007, 10007), 0),
         ((10007, 10007), 1), ((30_000_000, 4), 0), ((12_000_000, 8), 0), ((100_000, 64, 16), 1), ((3, 10_000_000, 2), 1),
         ((1_000_000, 64), 0), ((400_000, 128), 0),
         # tall and skinny with a few hundred to a few thousand columns (feature means over samples): few column tiles, many chunks
         ((500_000, 200), 0), ((400_000, 256), 0), ((200_000, 512), 0), ((100_000, 1000), 0), ((65536, 1000), 0), ((50_000, 2000), 0), ((25_000, 4000), 0)]
for shape, axis in CASES:
    n = int(np.prod(shape))
    x = D.DeviceArray(shape); D.fill(x, 1.0)
    for op in ("sum", "max"):
        out_shape = tuple(s for i, s in enumerate(shape) if i != axis) or (1,)
        out = D.DeviceArray(out_shape)
        ms = run(lambda: D.reduce_axis(op, x, axis, out=out))
        print("  %-28s axis %d %-4s %7.3f ms %6.0f GB/s" % (shape, axis, op, ms, 4.0 * n / ms / 1e6), flush=True)
        out.free()
    x.free()
