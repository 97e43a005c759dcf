"""Launch-bound sequences: 50 small elementwise ops issued one by one vs the same sequence captured
once (np_graph_begin / np_graph_end) and replayed with one graph launch.  Usage: python tools/graph_ab.py"""
import ctypes as C
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, check, load
D.init(0); lib = load()
for n in (1_000, 100_000, 1_000_000, 10_000_000):
    a, b, t1, t2 = (D.DeviceArray((n,)) for _ in range(4))
    D.fill(a, 0.5); D.fill(b, 0.25)

    def sequence():
        src = a
        for k in range(25):
            check(lib.np_binary(BINARY_OPS["add"], src.ptr, 0, b.ptr, 0, t1.ptr, 1, n, 0, 0))
            check(lib.np_unary(UNARY_OPS["tanh"], t1.ptr, t2.ptr, n, 0.0, 0.0))
            src = t2
    sequence(); D.sync()
    check(lib.np_graph_begin()); sequence()
    g = C.c_void_p(); check(lib.np_graph_end(C.byref(g)))
    res = {}
    for name, fn in (("eager", sequence), ("graph", lambda: check(lib.np_graph_launch(g)))):
        for _ in range(3): fn()
        D.sync(); t0 = time.perf_counter(); reps = 20
        for _ in range(reps): fn()
        D.sync(); res[name] = (time.perf_counter() - t0) / reps * 1e6
    print("n=%-9d 50 ops: eager %8.1f us (%.2f us/op)   graph %8.1f us (%.2f us/op)   x%.2f" %
          (n, res["eager"], res["eager"] / 50, res["graph"], res["graph"] / 50, res["eager"] / res["graph"]), flush=True)
    check(lib.np_graph_destroy(g))
    for x in (a, b, t1, t2): x.free()
