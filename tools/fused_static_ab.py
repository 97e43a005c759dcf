"""A/B of the compiled chains (np_fused_static.hip) against the chain interpreter (np_elementwise_set_variant(7000)) on
bench.py's fused cases, interleaved rounds:
  cols   sum(exp(X), axis 0), 25000 x 4000       variants: 7000 interpreter | 0 compiled, 2 rows in flight | 7004: 4 rows |
                                                  12000 + k / 14000 + k: 2 / 4 rows in flight with k workgroups per CU
  flat   exp(a)*b+2 (store), sum(exp(a)*b+2), exp(X)+row, exp(X)+col, sum(exp(a)), 1e8 elements
Store chains must be BIT-identical between the two; sums are compared with an fp64 accumulation.
Usage: python tools/fused_static_ab.py [rounds]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, Timer, check, load

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
D.init(0)
lib = load()
R, Cc = 25000, 4000
N = R * Cc
a = synth.uniform((N,), 5, 0.0, 1.0)
b = synth.uniform((N,), 6, 0.0, 1.0)
row = synth.uniform((Cc,), 9, 0.0, 1.0)
col = synth.uniform((R,), 10, 0.0, 1.0)
da, db, do = D.DeviceArray.from_host(a), D.DeviceArray.from_host(b), D.DeviceArray((N,))
drow, dcol = D.DeviceArray.from_host(row), D.DeviceArray.from_host(col)
two = C.c_float(2.0)
t = Timer()
e64 = np.exp(a.astype(np.float64))


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    D.sync()
    t.start()
    for _ in range(iters):
        fn()
    t.stop()
    return t.elapsed_ms() / iters * 1e3


# ---- cols ----
prog1 = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
ptrs1 = (C.c_void_p * 1)(da.ptr)
kinds1 = (C.c_int * 1)(0)
out0 = D.DeviceArray((Cc,))
ref0 = e64.reshape(R, Cc).sum(0)


def cols():
    check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, R, Cc, 0, out0.ptr))


for rnd in range(rounds):
    print("-- cols round", rnd, flush=True)
    for v in (7000, 0, 7004, 12006, 12008, 12012, 14006, 14008, 12016):
        check(lib.np_elementwise_set_variant(v))
        us = timed(cols)
        err = float((np.abs(out0.to_host().astype(np.float64) - ref0) / ref0).max())
        print("   sum(exp(X),0) variant %5d   %6.1f us  %5.0f GB/s  frac %.3f  max rel err %.1e" % (
            v, us, 4.0 * N / us / 1e3, 4.0 * N / us / 1e3 / 8000, err), flush=True)
check(lib.np_elementwise_set_variant(0))

# ---- flat ----
prog3 = (FusedOp * 3)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0),
                      FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0),
                      FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0))
ptrs3 = (C.c_void_p * 3)(da.ptr, db.ptr, C.cast(C.pointer(two), C.c_void_p))
kinds3 = (C.c_int * 3)(0, 0, 4)
prog2 = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0),
                      FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0))
res = C.c_float(0.0)


def case(name, nbytes, fn, kind, ref=None):
    """kind 'store': out `do` must be bit-identical between interpreter and compiled; 'sum': res vs ref"""
    line = "   %-26s" % name
    keep = None
    for v in (7000, 0):
        check(lib.np_elementwise_set_variant(v))
        us = timed(fn)
        line += "  %s %6.1f us %5.0f GB/s (%.3f)" % ("interp" if v else "compiled", us, nbytes / us / 1e3, nbytes / us / 1e3 / 8000)
        if kind == "store":
            got = do.to_host().reshape(-1).view(np.uint32)
            if keep is None:
                keep = got
            else:
                line += "  bit-identical: %s" % bool((keep == got).all())
        elif kind == "sum":
            line += "  rel err %.1e" % (abs(res.value - ref) / abs(ref))
    check(lib.np_elementwise_set_variant(0))
    print(line, flush=True)


sum3 = float((e64 * b.astype(np.float64) + 2.0).sum())
for rnd in range(rounds):
    print("-- flat round", rnd, flush=True)
    case("exp(a)*b+2", 12.0 * N, lambda: check(lib.np_fused_chain(ptrs3, kinds3, 3, prog3, 3, do.ptr, 1, N)), "store")
    case("sum(exp(a)*b+2)", 8.0 * N, lambda: check(lib.np_fused_chain_reduce(ptrs3, kinds3, 3, prog3, 3, 0, 1, N, C.byref(res))), "sum", sum3)
    case("sum(exp(a))", 4.0 * N, lambda: check(lib.np_fused_chain_reduce(ptrs1, kinds1, 1, prog1, 1, 0, 1, N, C.byref(res))), "sum", float(e64.sum()))
    # the same chain as the PHP binding builds it: the multiply carries NP_QUIRK_AVX_BODY + the end of the reference's AVX2 body
    prog3q = (FusedOp * 3)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0),
                           FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 1, N - N % 8),
                           FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0))
    case("exp(a)*b+2 (binding: quirk)", 12.0 * N, lambda: check(lib.np_fused_chain(ptrs3, kinds3, 3, prog3q, 3, do.ptr, 1, N)), "store")
    case("sum(exp(a)*b+2) (quirk)", 8.0 * N, lambda: check(lib.np_fused_chain_reduce(ptrs3, kinds3, 3, prog3q, 3, 0, 1, N, C.byref(res))), "sum", sum3)
    dres = D.DeviceArray((1,))
    case("sum(exp(a)*b+2) dev result", 8.0 * N, lambda: check(lib.np_fused_chain_reduce_dev(ptrs3, kinds3, 3, prog3, 3, 0, 1, N, dres.ptr)), "none")
    case("sum(exp(a)) dev result", 4.0 * N, lambda: check(lib.np_fused_chain_reduce_dev(ptrs1, kinds1, 1, prog1, 1, 0, 1, N, dres.ptr)), "none")
    out1 = D.DeviceArray((R,))
    case("sum(exp(X), axis 1)", 4.0 * N, lambda: check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, R, Cc, 1, out1.ptr)), "none")
    for label, dvec, k in (("row", drow, 2), ("col", dcol, 3)):
        ptrs2 = (C.c_void_p * 2)(da.ptr, dvec.ptr)
        kinds2 = (C.c_int * 2)(0, k)
        case("exp(X)+%s" % label, 8.0 * N, lambda: check(lib.np_fused_chain(ptrs2, kinds2, 2, prog2, 2, do.ptr, R, Cc)), "store")
