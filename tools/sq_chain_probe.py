"""Chains with a `** 2` step (squared differences, norms) next to a compiled chain of the same traffic, 1e8 elements / 25000 x 4000:
  store  (x - y) ** 2            12 B/elem      next to (x - y) * 0.5        (on the menu of np_fused_static.hip)
  mean   mean((x - y) ** 2)       8 B/elem      next to sum((x - y) * 0.5)
  norm   sum(x ** 2)              4 B/elem      next to sum(exp(x))
  axis   sum((x - y) ** 2, axis)  8 B/elem      next to sum((x - y) * 0.5, axis)
  twice  x * y + x and (x - y) * y  12 B/elem     an operand named twice: input 0 again, another array again
np_elementwise_set_variant(7000) forces the interpreter for the right-hand forms (what a chain off the menu runs on); 7001 lets the
compiled kernels take chains with a twice-named array (they stream it twice), 7002 is the interpreter without the plain loads for
twice-named arrays (round 5's behaviour).
Usage: python tools/sq_chain_probe.py [rounds]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, Timer, check, load

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
D.init(0)
lib = load()
R, Cc = 25000, 4000
N = R * Cc
x = synth.uniform((N,), 5, -1.0, 1.0)
y = synth.uniform((N,), 6, 0.0, 1.0)
dx, dy, do = D.DeviceArray.from_host(x), D.DeviceArray.from_host(y), D.DeviceArray((N,))
two = C.c_float(2.0)
t = Timer()
U, B = 0, 1
SUB, MUL, POW = BINARY_OPS["subtract"], BINARY_OPS["multiply"], BINARY_OPS["pow"]


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    D.sync()
    t.start()
    for _ in range(iters):
        fn()
    t.stop()
    return t.elapsed_ms() / iters * 1e3


def chain(steps, inputs, kinds):
    ops = (FusedOp * len(steps))(*[FusedOp(*s) for s in steps])
    ptrs = (C.c_void_p * len(inputs))(*inputs)
    k = (C.c_int * len(kinds))(*kinds)
    return ptrs, k, len(inputs), ops, len(steps)


HOST = 4   # NP_HOST_SCALAR
sq_diff = chain([(B, SUB, 1, 0, 0, 0, 0, 0), (B, POW, 2, 0, 0, 0, 0, 0)], [dx.ptr, dy.ptr, C.addressof(two)], [0, 0, HOST])
half = C.c_float(0.5)
mul_diff = chain([(B, SUB, 1, 0, 0, 0, 0, 0), (B, MUL, 2, 0, 0, 0, 0, 0)], [dx.ptr, dy.ptr, C.addressof(half)], [0, 0, HOST])
sq = chain([(B, POW, 1, 0, 0, 0, 0, 0)], [dx.ptr, C.addressof(two)], [0, HOST])
ex = chain([(U, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0)], [dx.ptr], [0])
ADD = BINARY_OPS["add"]
# an operand named twice: input 0 again (re-read where it is used) and another array again (streamed twice)
x_again = chain([(B, MUL, 1, 0, 0, 0, 0, 0), (B, ADD, 0, 0, 0, 0, 0, 0)], [dx.ptr, dy.ptr], [0, 0])
y_again = chain([(B, SUB, 1, 0, 0, 0, 0, 0), (B, MUL, 1, 0, 0, 0, 0, 0)], [dx.ptr, dy.ptr], [0, 0])
res = C.c_float()
out_r, out_c = D.DeviceArray((R,)), D.DeviceArray((Cc,))

cases = [
    ("store (x-y)**2", 12, lambda: check(lib.np_fused_chain(*sq_diff, do.ptr, 1, N))),
    ("store (x-y)*.5", 12, lambda: check(lib.np_fused_chain(*mul_diff, do.ptr, 1, N))),
    ("store x*y+x", 12, lambda: check(lib.np_fused_chain(*x_again, do.ptr, 1, N))),
    ("store (x-y)*y", 12, lambda: check(lib.np_fused_chain(*y_again, do.ptr, 1, N))),
    ("mean((x-y)**2)", 8, lambda: check(lib.np_fused_chain_reduce(*sq_diff, 4, 1, N, C.byref(res)))),
    ("sum((x-y)*.5)", 8, lambda: check(lib.np_fused_chain_reduce(*mul_diff, 0, 1, N, C.byref(res)))),
    ("sum(x**2)", 4, lambda: check(lib.np_fused_chain_reduce(*sq, 0, 1, N, C.byref(res)))),
    ("sum(exp(x))", 4, lambda: check(lib.np_fused_chain_reduce(*ex, 0, 1, N, C.byref(res)))),
    ("sum((x-y)**2, 0)", 8, lambda: check(lib.np_fused_chain_reduce_axis(*sq_diff, 0, R, Cc, 0, out_c.ptr))),
    ("sum((x-y)*.5, 0)", 8, lambda: check(lib.np_fused_chain_reduce_axis(*mul_diff, 0, R, Cc, 0, out_c.ptr))),
    ("sum((x-y)**2, 1)", 8, lambda: check(lib.np_fused_chain_reduce_axis(*sq_diff, 0, R, Cc, 1, out_r.ptr))),
    ("sum((x-y)*.5, 1)", 8, lambda: check(lib.np_fused_chain_reduce_axis(*mul_diff, 0, R, Cc, 1, out_r.ptr))),
]
for rnd in range(rounds):
    for v in (0, 7000, 7001, 7002):
        check(lib.np_elementwise_set_variant(v))
        for name, bpe, fn in cases:
            us = timed(fn)
            print("round %d variant %4d  %-18s %7.1f us  %5.2f TB/s" % (rnd, v, name, us, bpe * N / us / 1e6), flush=True)
check(lib.np_elementwise_set_variant(0))
d64 = (x.astype(np.float64) - y.astype(np.float64)) ** 2
check(lib.np_fused_chain_reduce(*sq_diff, 4, 1, N, C.byref(res)))
print("mse rel err %.2e" % (abs(res.value - d64.mean()) / d64.mean()))
