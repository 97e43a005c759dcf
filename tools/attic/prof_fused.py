"""Profiling driver (dev tool): the fused-chain cases of bench.py, a few launches each, for rocprofv3 passes.
Usage: python tools/prof_fused.py [iters]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, check, load

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
only = sys.argv[2] if len(sys.argv) > 2 else ""      # "row": just exp(X) + row and the plain exp kernel
D.init(0)
lib = load()
N = 100_000_000
R, Cc = 25000, 4000
a = D.DeviceArray.from_host(synth.uniform((N,), 5))
b = D.DeviceArray.from_host(synth.uniform((N,), 6))
o = D.DeviceArray((N,))
row = D.DeviceArray.from_host(synth.uniform((Cc,), 9))
col = D.DeviceArray.from_host(synth.uniform((R,), 10))
prog = (FusedOp * 3)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 1, N // 8 * 8),
                     FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0))
two = C.c_float(2.0)
ptrs = (C.c_void_p * 3)(a.ptr, b.ptr, C.cast(C.pointer(two), C.c_void_p))
kinds = (C.c_int * 3)(0, 0, 4)
out = C.c_float(0.0)
for _ in range(0 if only else iters):
    check(lib.np_fused_chain(ptrs, kinds, 3, prog, 3, o.ptr, 1, N))
for _ in range(0 if only else iters):
    check(lib.np_fused_chain_reduce(ptrs, kinds, 3, prog, 3, 0, 1, N, C.byref(out)))
prog2 = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0))
for dvec, kind in (((row, 2),) if only == "row" else ((row, 2), (col, 3))):
    ptrs2 = (C.c_void_p * 2)(a.ptr, dvec.ptr)
    kinds2 = (C.c_int * 2)(0, kind)
    for _ in range(iters):
        check(lib.np_fused_chain(ptrs2, kinds2, 2, prog2, 2, o.ptr, R, Cc))
for _ in range(iters):
    D.unary("exp", a, out=o)
D.sync()
print("done")
