"""Dev tool: the 2-D broadcast kernels (binary_rows2d_kernel for X (op) row, cchain_tile2d_kernel for compiled chains with
broadcast operands) against the flat kernels they replace (np_elementwise_set_variant(8100)): time in alternation, GB/s over the
algorithmic bytes, and bit-identity of the stored values.
Usage: python tools/bcast2d_ab.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, Timer, check, load

D.init(0)
lib = load()
t = Timer()


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    D.sync()
    t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps


def ab(name, nbytes, fn, out):
    res, bits = {}, {}
    for rnd in range(3):
        for label, code in (("flat", 8100), ("2-D", 8101)):
            check(lib.np_elementwise_set_variant(code))
            res.setdefault(label, []).append(timed(fn))
            if rnd == 0:
                bits[label] = out.to_host().view(np.uint32).copy()
    check(lib.np_elementwise_set_variant(8101))
    f, d = float(np.median(res["flat"])), float(np.median(res["2-D"]))
    print("  %-44s flat %7.1f us %5.0f GB/s (%.3f)   2-D %7.1f us %5.0f GB/s (%.3f)   bit-identical: %s" % (
        name, f * 1e3, nbytes / f / 1e6, nbytes / f / 8e9, d * 1e3, nbytes / d / 1e6, nbytes / d / 8e9, bool((bits["flat"] == bits["2-D"]).all())), flush=True)


for rows, cols in ((25000, 4000), (100000, 1024), (4000, 25000), (1000, 100000), (40, 2000000), (25000, 4004)):
    n = rows * cols
    X = D.DeviceArray.from_host(synth.uniform((n,), 5))
    row = D.DeviceArray.from_host(synth.uniform((cols,), 6))
    col = D.DeviceArray.from_host(synth.uniform((rows,), 7))
    out = D.DeviceArray((n,))
    print("%d x %d" % (rows, cols))
    ab("X + row", 8.0 * n, lambda: D.binary("add", X, "full", row, "row", rows, cols, out=out), out)
    ab("row / X", 8.0 * n, lambda: D.binary("divide", row, "row", X, "full", rows, cols, out=out), out)
    for label, kinds, ptr2 in (("exp(X) + row", 2, row), ("exp(X) + col", 3, col)):
        prog = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0))
        ptrs = (C.c_void_p * 2)(X.ptr, ptr2.ptr)
        kd = (C.c_int * 2)(0, kinds)
        ab(label, 8.0 * n, lambda: check(lib.np_fused_chain(ptrs, kd, 2, prog, 2, out.ptr, rows, cols)), out)
    prog = (FusedOp * 3)(FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 3, 0, 0, 0, 0, 0))
    ptrs = (C.c_void_p * 4)(X.ptr, X.ptr, col.ptr, row.ptr)
    kd = (C.c_int * 4)(0, 0, 3, 2)
    ab("X * X + col + row", 12.0 * n, lambda: check(lib.np_fused_chain(ptrs, kd, 4, prog, 3, out.ptr, rows, cols)), out)
    for d in (X, row, col, out):
        d.free()
