"""Profiling driver (dev tool): launches each headline kernel at BASELINE sizes so that rocprofv3 (--kernel-trace --stats, or
--pmc passes) sees clean per-kernel dispatches.
Usage: python tools/prof_kernels.py [iters] [warm_ms]
warm_ms > 0 (the kernel-trace pass: 250): each kernel keeps being launched, back to back, until warm_ms have passed — the
per-kernel AVERAGE rocprofv3 reports is then that of a kernel running on a clock that has come up, the state bench.py's timed
region measures (20 launches right after an upload average 7 % slower than the bench's ms_per_step).
The PMC passes keep warm_ms = 0: counters do not depend on the clock and every dispatch is serialised there.

Round 6: every workload is a SECTION named after its key in bench.py's `extras` (section("add_1e8") ...).  A section starts
with one MARKER launch (np_debug_raise_device_error(0): `raise_error_kernel`, which ORs nothing into the error word), so that
tools/pmc_traffic.py can cut the counter CSV — ordered by dispatch — at the markers and sum FETCH_SIZE / WRITE_SIZE over
EVERYTHING a call launched (np_moments' two kernels, the median's six, a split-K product's fold), then divide by the number of
calls: HBM bytes per call for every line of the bench, not only for single-kernel entries.  The section list (label, calls)
goes to the file named by NP_PROF_SECTIONS (default gpurun_out/prof_sections.json)."""
import json
import os
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
SECTIONS = []          # [label, calls]


def section(label):
    """Start the section `label`: a marker launch, then whatever the following `for _ in range(iters)` loops launch."""
    from numpower_amd._lib import check as _check, load as _load
    _check(_load().np_debug_raise_device_error(0))
    SECTIONS.append([label, 0])


def range(n):   # noqa: A001 — every `for _ in range(iters)` below becomes time-bounded, and counts its calls into the open section
    def counted():
        for k in _Times(n):
            if SECTIONS:
                SECTIONS[-1][1] += 1
            yield k
    return counted()


import ctypes as C
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, FusedOp, check, load
lib = load()

n = 4096
A = D.DeviceArray.from_host(synth.uniform((n, n), 3, -1, 1)); B = D.DeviceArray.from_host(synth.uniform((n, n), 4, -1, 1)); Cm = D.DeviceArray((n, n))
section("sgemm_dma_kernel")          # the headline: nd::matmul 4096^2
for _ in range(iters): D.sgemm(A, B, out=Cm)
D.sync()
section("_setup")                    # (labels that start with "_" are not workloads: uploads and fills between them launch kernels too)
for d in (A, B, Cm): d.free()
N = 100_000_000
a = D.DeviceArray.from_host(synth.uniform((N,), 5)); b = D.DeviceArray.from_host(synth.uniform((N,), 6)); o = D.DeviceArray((N,))
section("add_1e8")
for _ in range(iters): D.binary("add", a, "full", b, "full", 1, N, out=o)
section("greater_1e8")
for _ in range(iters): D.binary("greater", a, "full", b, "full", 1, N, out=o)
section("pow_1e8")
for _ in range(iters): D.binary("pow", a, "full", b, "full", 1, N, out=o)
section("exp_1e8")
for _ in range(iters): D.unary("exp", a, out=o)
section("log_1e8")
for _ in range(iters): D.unary("log", b, out=o)
row = D.DeviceArray.from_host(synth.uniform((4000,), 9)); col = D.DeviceArray.from_host(synth.uniform((25000,), 10))
section("add_row_broadcast")
for _ in range(iters): D.binary("add", a, "full", row, "row", 25000, 4000, out=o)
section("add_col_broadcast")
for _ in range(iters): D.binary("add", a, "full", col, "col", 25000, 4000, out=o)
# C3c as BASELINE.json words it: exp(X) + r in ONE pass through the fused chain (bench.py: exp_plus_row_fused / exp_plus_col_fused)
prog2 = (FusedOp * 2)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["add"], 1, 0, 0, 0, 0, 0))
for label, dvec, kind in (("row", row, 2), ("col", col, 3)):
    ptrs2 = (C.c_void_p * 2)(a.ptr, dvec.ptr); kinds2 = (C.c_int * 2)(0, kind)
    section("exp_plus_%s_fused" % label)
    for _ in range(iters): check(lib.np_fused_chain(ptrs2, kinds2, 2, prog2, 2, o.ptr, 25000, 4000))
section("sum_1e8")
for _ in range(iters): D.reduce_all("sum", a)
flag = C.c_int(0)
section("allclose_1e8")
for _ in range(iters): check(lib.np_count_mismatch(1, a.ptr, b.ptr, N, 1e-5, 1e-8, C.byref(flag)))
two_f = (C.c_float * 2)()
section("median_1e8")
for _ in range(iters): check(lib.np_order_stat(a.ptr, N, N // 2 - 1, two_f))
# fused chain exp(a)*b+2
prog = (FusedOp * 3)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0), FusedOp(1, BINARY_OPS["multiply"], 1, 0, 0, 0, 0, 0),
                     FusedOp(1, BINARY_OPS["add"], 2, 0, 0, 0, 0, 0))
two = C.c_float(2.0)
ptrs = (C.c_void_p * 3)(a.ptr, b.ptr, C.cast(C.pointer(two), C.c_void_p)); kinds = (C.c_int * 3)(0, 0, 4)
section("fused_chain_1e8")
for _ in range(iters): check(lib.np_fused_chain(ptrs, kinds, 3, prog, 3, o.ptr, 1, N))
# ... the same chain ending in a sum (device result), and sum(exp(X), axis) over 25000 x 4000: the compiled chains
dsum = D.DeviceArray((1,))
section("fused_chain_sum_1e8")
for _ in range(iters): check(lib.np_fused_chain_reduce_dev(ptrs, kinds, 3, prog, 3, 0, 1, N, dsum.ptr))
prog1 = (FusedOp * 1)(FusedOp(0, UNARY_OPS["exp"], 0, 0, 0, 0, 0, 0))
ptrs1 = (C.c_void_p * 1)(a.ptr); kinds1 = (C.c_int * 1)(0)
red0 = D.DeviceArray((4000,)); red1 = D.DeviceArray((25000,))
section("sum_exp_axis0_fused")
for _ in range(iters): check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, 25000, 4000, 0, red0.ptr))
section("sum_exp_axis1_fused")
for _ in range(iters): check(lib.np_fused_chain_reduce_axis(ptrs1, kinds1, 1, prog1, 1, 0, 25000, 4000, 1, red1.ptr))
# SURVEY 8(f) row 2 at 1e8: variance in one read, the weighted average's two sums, argmax of the flat array
mean, m2, saw, sw = C.c_float(), C.c_float(), C.c_float(), C.c_float()
section("moments_1e8")
for _ in range(iters): check(lib.np_moments(a.ptr, N, C.byref(mean), C.byref(m2)))
section("weighted_sums_1e8")
for _ in range(iters): check(lib.np_weighted_sums(a.ptr, b.ptr, N, C.byref(saw), C.byref(sw)))
idx = D.DeviceArray((65536,))
section("argmax_1e8")
for _ in range(iters): check(lib.np_argreduce(1, a.ptr, 1, N, 1, idx.ptr))
section("sgemv_10x1e7")
for _ in range(iters): check(lib.np_sgemv(10, 10_000_000, a.ptr, b.ptr, idx.ptr))
D.sync()
section("_setup")
a.free(); b.free(); o.free()
X = D.DeviceArray((65536, 4096)); D.fill(X, 0.5); out = D.DeviceArray((4096,)); out1 = D.DeviceArray((65536,))
section("sum_axis0")
for _ in range(iters): D.reduce_axis("sum", X, 0, out=out)
section("sum_axis1_65536x4096")
for _ in range(iters): D.reduce_axis("sum", X, 1, out=out1)
XT = D.DeviceArray((4096, 65536))
section("transpose_65536x4096")
for _ in range(iters): check(lib.np_transpose2d(X.ptr, XT.ptr, 1, 65536, 4096))
section("argmax_axis1_65536x1024")
for _ in range(iters): check(lib.np_argreduce(1, X.ptr, 65536, 1024, 1, idx.ptr))
# bench.py's mid sizes: a transpose whose rows are off the 128-byte grid, an NHWC-like permute, the small / deep-K products
section("transpose_8191x8193")
for _ in range(iters): check(lib.np_transpose2d(X.ptr, XT.ptr, 1, 8191, 8193))
shape, perm = (C.c_int * 4)(60, 128, 1024, 8), (C.c_int * 4)(0, 2, 1, 3)
section("permute_nhwc_like")
for _ in range(iters): check(lib.np_permute(X.ptr, XT.ptr, 4, shape, perm))
D.sync()
for (m_, n_, k_) in ((768,) * 3, (1000,) * 3, (1024,) * 3, (100, 100, 100000)):
    section("_setup")
    dA = D.DeviceArray.from_host(synth.uniform((m_, k_), 31, -1.0, 1.0)); dB = D.DeviceArray.from_host(synth.uniform((k_, n_), 32, -1.0, 1.0))
    dC = D.DeviceArray((m_, n_))
    section("matmul_%d" % n_ if m_ == n_ == k_ else "matmul_%dx%dx%d" % (m_, n_, k_))
    for _ in range(iters): D.sgemm(dA, dB, out=dC)
    D.sync()
    for d in (dA, dB, dC): d.free()
section("end")
D.sync()
path = Path(os.environ.get("NP_PROF_SECTIONS", str(Path(__file__).resolve().parent.parent / "gpurun_out" / "prof_sections.json")))
path.parent.mkdir(parents=True, exist_ok=True)
path.write_text(json.dumps(SECTIONS))
print("done", sum(1 for label, _ in SECTIONS if not label.startswith("_") and label != "end"), "workloads")
