import ctypes as C, sys
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import Timer, check, load
D.init(0); lib = load(); t = Timer()
N = 100_000_000
big, out = D.DeviceArray((N,)), D.DeviceArray((N,)); D.fill(big, 1.5)
def run(fn, reps=20):
    for _ in range(3): fn()
    D.sync(); t.start()
    for _ in range(reps): fn()
    t.stop(); return t.elapsed_ms() / reps
for rnd in range(2):
    for rows, cols in ((7, 10_000_000), (40, 2_000_000), (3, 30_000_000), (200, 400_000), (1000, 100_000)):
        row = D.DeviceArray((cols,)); D.fill(row, 2.0)
        line = "X + row %dx%d" % (rows, cols)
        for v in (0, 8000):
            check(lib.np_elementwise_set_variant(v))
            ms = run(lambda: check(lib.np_binary(0, big.ptr, 0, row.ptr, 2, out.ptr, rows, cols, 0, 0)))
            line += "   %s %.3f ms %5.0f GB/s" % ("column blocks" if v == 0 else "plain order", ms, 8.0 * rows * cols / ms / 1e6)
        check(lib.np_elementwise_set_variant(0))
        print(line, flush=True); row.free()
