"""Profiling driver (dev tool): launches each headline kernel at BASELINE sizes so that rocprofv3 (--kernel-trace --stats, or
--pmc passes) sees clean per-kernel dispatches.
Usage: python tools/prof_kernels.py [iters] [warm_ms]
warm_ms > 0 (the kernel-trace pass: 250): each kernel keeps being launched, back to back, until warm_ms have passed — the
per-kernel AVERAGE rocprofv3 reports is then that of a kernel running on a clock that has come up, the state bench.py's timed
region measures (20 launches right after an upload average 7 % slower than the bench's ms_per_step).
The PMC passes keep warm_ms = 0: counters do not depend on the clock and every dispatch is serialised there."""
import sys
import time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D, synth

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
warm_s = (float(sys.argv[2]) if len(sys.argv) > 2 else 0.0) / 1e3
D.init(0)


class _Times:
    """`for _ in range(iters)` that, with warm_ms, keeps going until the time is up (synchronising every 16 launches so that
    the clock it reads is the device's, not the queue's)."""
    def __init__(self, n):
        self.n = n

    def __iter__(self):
        t0, k = time.perf_counter(), 0
        while k < self.n or time.perf_counter() - t0 < warm_s:
            yield k
            k += 1
            if warm_s and k % 16 == 0:
                D.sync()


_range = range


def range(n):   # noqa: A001 — every `for _ in range(iters)` below becomes time-bounded
    return _Times(n)


n = 4096
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1)); B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1)); Cm = D.DeviceArray((n, n))
for _ in range(iters): D.sgemm(A, B, out=Cm)
D.sync()
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5)); b = D.DeviceArray.from_host(synth.uniform((N,), 6)); o = D.DeviceArray((N,))
for _ in range(iters): D.binary("add", a, "full", b, "full", 1, N, out=o)
for _ in range(iters): D.binary("pow", a, "full", b, "full", 1, N, out=o)
for _ in range(iters): D.unary("exp", a, out=o)
for _ in range(iters): D.unary("log", b, out=o)
row = D.DeviceArray.from_host(synth.uniform((4000,), 9)); col = D.DeviceArray.from_host(synth.uniform((25000,), 10))
for _ in range(iters): D.binary("add", a, "full", row, "row", 25000, 4000, out=o)
for _ in range(iters): D.binary("add", a, "full", col, "col", 25000, 4000, out=o)
for _ in range(iters): D.reduce_all("sum", a)
# fused chain exp(a)*b+2
import ctypes as C
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, check, load
lib = load()
prog = (FusedOp * 3)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0),
                     FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0))
two = C.c_float(2.0)
ptrs = (C.c_void_p * 3)(a.ptr, b.ptr, C.cast(C.pointer(two), C.c_void_p)); kinds = (C.c_int * 3)(0, 0, 4)
for _ in range(iters): check(lib.np_fused_chain(ptrs, kinds, 3, prog, 3, o.ptr, 1, N))
# ... the same chain ending in a sum (device result), and sum(exp(X), axis) over 25000 x 4000: the compiled chains
dsum = D.DeviceArray((1,))
for _ in range(iters): check(lib.np_fused_chain_reduce_dev(ptrs, kinds, 3, prog, 3, 0, 1, N, dsum.ptr))
prog1 = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
ptrs1 = (C.c_void_p * 1)(a.ptr); kinds1 = (C.c_int * 1)(0)
red0 = D.DeviceArray((4000,)); red1 = D.DeviceArray((25000,))
for _ in range(iters): check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, 25000, 4000, 0, red0.ptr))
for _ in range(iters): check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, 25000, 4000, 1, red1.ptr))
D.sync()
a.free(); b.free(); o.free()
X = D.DeviceArray((65536, 4096)); D.fill(X, 0.5); out = D.DeviceArray((4096,)); out1 = D.DeviceArray((65536,))
for _ in range(iters): D.reduce_axis("sum", X, 0, out=out)
for _ in range(iters): D.reduce_axis("sum", X, 1, out=out1)
XT = D.DeviceArray((4096, 65536))
for _ in range(iters): check(lib.np_transpose2d(X.ptr, XT.ptr, 1, 65536, 4096))
# SURVEY 8(f) rows 2 and 4 as bench.py's extras run them (round 5): argmax over the last axis of 65536 x 1024 (one wave per
# row), argmax of 1e8 flat, variance's second pass, dot(matrix, vector) with ten long rows
idx = D.DeviceArray((65536,))
for _ in range(iters): check(lib.np_argreduce(1, X.ptr, 65536, 1024, 1, idx.ptr))
for _ in range(iters): check(lib.np_argreduce(1, X.ptr, 1, 100_000_000, 1, idx.ptr))
mean, m2 = C.c_float(), C.c_float()
for _ in range(iters): check(lib.np_moments(X.ptr, 100_000_000, C.byref(mean), C.byref(m2)))
for _ in range(iters): check(lib.np_sgemv(10, 10_000_000, X.ptr, XT.ptr, idx.ptr))
D.sync()
print("done")
