import sys
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import ctypes as C
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import Timer, check, load
D.init(0); lib = load(); t = Timer()
N = 100_000_000
big, out = D.DeviceArray((N,)), D.DeviceArray((N,)); D.fill(big, 1.5)
def run(fn, reps=10):
    for _ in range(2): fn()
    D.sync(); t.start()
    for _ in range(reps): fn()
    t.stop(); return t.elapsed_ms() / reps
for variant in (1, 0, 1, 0):   # 1 = plain tiles, 0 = default (write-aligned form where output rows are off the line grid)
  check(lib.np_layout_set_variant(variant)); print("-- np_layout_set_variant(%d)" % variant)
  for rows, cols in ((8192, 8192), (8192, 8196), (8196, 8192), (8192, 8200), (8200, 8192), (8192, 8224), (8224, 8192), (8200, 8200), (8224, 8224), (8192, 8193), (8193, 8192), (8191, 8193), (8190, 8194), (8188, 8196), (4099, 4099), (12345, 6789), (100_000, 1000), (1000, 100_000), (65536, 1500), (1500, 65536)):
    ms = run(lambda: check(lib.np_transpose2d(big.ptr, out.ptr, 1, rows, cols)))
    print("  " + "transpose %5d x %5d   %.3f ms  %5.0f GB/s" % (rows, cols, ms, 8.0 * rows * cols / ms / 1e6), flush=True)
