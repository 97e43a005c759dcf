"""np_order_stat (radix select) at 10^8 elements on a few distributions: time per call, GB/s over
the 12 B/elem the three passes read.  Usage: python tools/select_sweep.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth, _lib
from numpower_amd._lib import Timer
D.init(0); lib = _lib.load(); t = Timer()
N = 100_000_000
out = (C.c_float * 2)()
for label, host in (("U[0,1)", lambda: synth.uniform((N,), 5, 0.0, 1.0)),
                    ("U[-3,5)", lambda: synth.uniform((N,), 6, -3.0, 5.0)),
                    ("10 distinct values", lambda: np.rint(synth.uniform((N,), 7, 0.0, 9.0)).astype(np.float32)),
                    ("relu(U[-1,1))", lambda: np.maximum(synth.uniform((N,), 8, -1.0, 1.0), np.float32(0.0))),
                    ("constant", lambda: np.full((N,), 2.5, np.float32))):
    x = D.DeviceArray.from_host(host())
    for k in (N // 2, (3 * N) // 4, N - 1):
        for _ in range(2): _lib.check(lib.np_order_stat(x.ptr, N, k, out))
        D.sync(); t.start()
        for _ in range(5): _lib.check(lib.np_order_stat(x.ptr, N, k, out))
        t.stop(); ms = t.elapsed_ms() / 5
        print("  %-20s k=%-10d %7.3f ms  %6.0f GB/s   -> %r, %r" % (label, k, ms, 12.0 * N / ms / 1e6, out[0], out[1]), flush=True)
    x.free()
