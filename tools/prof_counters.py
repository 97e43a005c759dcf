"""Profiling driver, round 2 (dev tool): the kernels VERDICT r01 asks counters for — the 4096^2 GEMM
(MFMA utilisation), pow at 1e8 (VALU-bound? clock?), add at 1e8 and the fused sum(exp(X), 0) — a few
clean launches each so rocprofv3 --pmc / --kernel-trace passes see one dispatch per launch.
Usage: python tools/prof_counters.py [iters] [which: gemm,pow,add,cols,rows]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from numpower_amd import device as D, synth
from numpower_amd._lib import UNARY_OPS, FusedOp, check, load

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = set((sys.argv[2] if len(sys.argv) > 2 else "gemm,pow,add,cols,rows").split(","))
D.init(0)
lib = load()
if "gemm" in which:
    n = 4096
    A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1))
    B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1))
    Cm = D.DeviceArray((n, n))
    for _ in range(iters):
        D.sgemm(A, B, out=Cm)
    D.sync()
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5))
b = D.DeviceArray.from_host(synth.uniform((N,), 6))
o = D.DeviceArray((N,))
if "add" in which:
    for _ in range(iters):
        D.binary("add", a, "full", b, "full", 1, N, out=o)
if "pow" in which:
    import os
    check(lib.np_elementwise_set_variant(int(os.environ.get("NP_PROF_POW_VARIANT", "0"))))   # 9000: the log2 table in LDS instead of registers
    for _ in range(iters):
        D.binary("pow", a, "full", b, "full", 1, N, out=o)
    check(lib.np_elementwise_set_variant(0))
R, Cc = 25000, 4000
prog1 = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
ptrs1 = (C.c_void_p * 1)(a.ptr)
kinds1 = (C.c_int * 1)(0)
# sum(exp(X), axis): the compiled chain (cchain_cols_kernel / cchain_rows_kernel) and, behind it, the chain interpreter it
# replaced for this chain (np_elementwise_set_variant(7000): fused_chain_cols_kernel / fused_chain_rows_kernel)
for axis, nout, key in ((0, Cc, "cols"), (1, R, "rows")):
    if key in which:
        dred = D.DeviceArray((nout,))
        for variant in (0, 7000):
            check(lib.np_elementwise_set_variant(variant))
            for _ in range(iters):
                check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, R, Cc, axis, dred.ptr))
        check(lib.np_elementwise_set_variant(0))
D.sync()
print("done")
