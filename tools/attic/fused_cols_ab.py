"""A/B of fused_chain_cols_kernel's launch shape: sum(exp(X), axis 0) on 25000 x 4000 (bench.py's
sum_exp_axis0_fused) under np_elementwise_set_variant(1000 + k) = k workgroups per CU with 64-slot column
blocks (four waves on four different rows) and (2000 + k) = 256-slot column blocks (the workgroup walks down
the rows together).  Interleaved rounds; the axis-1 (row) variant of the same chain as the yardstick.
Usage: python tools/fused_cols_ab.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import UNARY_OPS, FusedOp, Timer, check, load

D.init(0)
lib = load()
R, Cc = 25000, 4000
a = synth.uniform((R * Cc,), 5, 0.0, 1.0)
da = D.DeviceArray.from_host(a)
prog = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
ptrs = (C.c_void_p * 1)(da.ptr)
kinds = (C.c_int * 1)(0)
ref = np.exp(a.reshape(R, Cc).astype(np.float64))
out0, out1 = D.DeviceArray((Cc,)), D.DeviceArray((R,))
t = Timer()


def run(axis, out, iters=30):
    for _ in range(3):
        check(lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, prog, 1, 0, R, Cc, axis, out.ptr))
    D.sync()
    t.start()
    for _ in range(iters):
        check(lib.np_fused_chain_reduce_axis(ptrs, kinds, 1, prog, 1, 0, R, Cc, axis, out.ptr))
    t.stop()
    return t.elapsed_ms() / iters * 1e3


variants = [0, 1004, 1007, 1008, 1009, 1010, 1012, 0, 1008]   # 100k = k workgroups per CU; 600k = k rows in flight per lane (default 2: 6001 / 6003 / 6004 measured 97 / 93 / 86 us)
for rnd in range(3):
    print("-- round", rnd, flush=True)
    print("   axis 1 (rows kernel)            %6.1f us" % run(1, out1))
    for v in variants:
        check(lib.np_elementwise_set_variant(v))
        us = run(0, out0)
        err = float((np.abs(out0.to_host().astype(np.float64) - ref.sum(0)) / ref.sum(0)).max())
        print("   axis 0 variant %4d              %6.1f us   %5.0f GB/s   max rel err %.1e" % (v, us, 4.0 * R * Cc / us / 1e3, err), flush=True)
    check(lib.np_elementwise_set_variant(0))
