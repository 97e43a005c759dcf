"""Throughput of the remaining entry points over awkward shapes (odd sizes, misaligned views, few
outputs): looking for paths that fall far off the HBM roofline.  Usage: python tools/misc_sweep.py"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
from numpower_amd import device as D
from numpower_amd._lib import BINARY_OPS, UNARY_OPS, Timer, check, load
D.init(0); lib = load(); t = Timer()


def run(fn, reps=10):
    for _ in range(2): fn()
    D.sync(); t.start()
    for _ in range(reps): fn()
    t.stop()
    return t.elapsed_ms() / reps


def line(name, ms, nbytes):
    print("  %-58s %8.3f ms %7.0f GB/s" % (name, ms, nbytes / ms / 1e6), flush=True)


N = 100_000_000
big = D.DeviceArray((N + 64,)); D.fill(big, 1.5)
big2 = D.DeviceArray((N + 64,)); D.fill(big2, 0.5)
out = D.DeviceArray((N + 64,))
print("elementwise on misaligned views (pointer + 4 bytes) and odd lengths")
for off_a, off_o, n in ((0, 0, N), (1, 0, N), (0, 1, N), (1, 1, N), (1, 2, N - 3)):
    line("add a+%d out+%d n=%d" % (off_a, off_o, n),
         run(lambda: check(lib.np_binary(0, big.ptr + 4 * off_a, 0, big2.ptr, 0, out.ptr + 4 * off_o, 1, n, 0, 0))), 12.0 * n)
    line("exp a+%d out+%d n=%d" % (off_a, off_o, n),
         run(lambda: check(lib.np_unary(UNARY_OPS["exp"], big.ptr + 4 * off_a, out.ptr + 4 * off_o, n, 0.0, 0.0))), 8.0 * n)
print("broadcast with odd row lengths")
for rows, cols in ((25000, 4000), (25000, 4001), (33333, 3001), (10_000_000, 7), (7, 10_000_000), (40, 2_000_000), (1_000_000, 100)):
    n = rows * cols
    row = D.DeviceArray((cols,)); D.fill(row, 2.0); col = D.DeviceArray((rows,)); D.fill(col, 3.0)
    line("X + row  %dx%d" % (rows, cols), run(lambda: check(lib.np_binary(0, big.ptr, 0, row.ptr, 2, out.ptr, rows, cols, 0, 0))), 8.0 * n)
    line("X + col  %dx%d" % (rows, cols), run(lambda: check(lib.np_binary(0, big.ptr, 0, col.ptr, 3, out.ptr, rows, cols, 0, 0))), 8.0 * n)
    if cols >= (1 << 20):   # a long row operand: the column-block work order (default) against the plain one
        check(lib.np_elementwise_set_variant(8000))
        line("X + row  %dx%d, plain order (variant 8000)" % (rows, cols), run(lambda: check(lib.np_binary(0, big.ptr, 0, row.ptr, 2, out.ptr, rows, cols, 0, 0))), 8.0 * n)
        check(lib.np_elementwise_set_variant(0))
    row.free(); col.free()
print("transpose / permute / strided copy")
for rows, cols in ((8192, 8192), (8191, 8193), (10_000_000, 3), (3, 10_000_000), (100_000, 1000), (1000, 100_000), (4099, 4099)):
    n = rows * cols
    line("transpose2d %dx%d" % (rows, cols), run(lambda: check(lib.np_transpose2d(big.ptr, out.ptr, 1, rows, cols))), 8.0 * n)
for shape, perm in (((64, 128, 1024, 8), (0, 2, 1, 3)), ((256, 512, 512), (2, 1, 0)), ((256, 512, 512), (1, 0, 2)), ((100, 100, 100, 100), (3, 2, 1, 0)),
                    ((30, 3, 1024, 1024), (0, 2, 3, 1)), ((30, 1024, 1024, 3), (0, 3, 1, 2))):
    n = int(np.prod(shape))
    sh = (C.c_int * len(shape))(*shape); pm = (C.c_int * len(perm))(*perm)
    line("permute %s %s" % (shape, perm), run(lambda: check(lib.np_permute(big.ptr, out.ptr, len(shape), sh, pm))), 8.0 * n)
for shape, strides, what in (((5000, 10000), (20000, 2), "every 2nd column"), ((10000, 5000), (10000, 1), "left half"),
                             ((10000,), (10001,), "diagonal of 10000^2"), ((50_000_000,), (2,), "stride 2 vector")):
    n = int(np.prod(shape))
    sh = (C.c_int * len(shape))(*shape); st = (C.c_longlong * len(strides))(*strides)
    line("strided_copy %s (%s)" % (shape, what), run(lambda: check(lib.np_strided_copy(big.ptr, out.ptr, len(shape), sh, st))), 8.0 * n)
print("argreduce")
for outer, L, inner in ((1, N, 1), (3, 30_000_000, 1), (65536, 1024, 1), (1, 30_000_000, 3), (1, 65536, 1024), (1000, 1000, 100), (1, 9973, 9973)):
    n = outer * L * inner
    line("argmax outer=%d len=%d inner=%d" % (outer, L, inner), run(lambda: check(lib.np_argreduce(1, big.ptr, outer, L, inner, out.ptr)), reps=3), 4.0 * n)
print("statistics / equality on misaligned views")
mean, m2 = C.c_float(), C.c_float(); flag = C.c_int()
line("moments aligned", run(lambda: check(lib.np_moments(big.ptr, N, C.byref(mean), C.byref(m2)))), 4.0 * N)   # one read since round 6
line("moments ptr+4", run(lambda: check(lib.np_moments(big.ptr + 4, N, C.byref(mean), C.byref(m2)))), 4.0 * N)   # one read since round 6
line("allclose ptr+4", run(lambda: check(lib.np_count_mismatch(1, big.ptr + 4, big2.ptr + 4, N, 1e-5, 1e-8, C.byref(flag)))), 8.0 * N)
line("reduce_all sum ptr+4", run(lambda: check(lib.np_reduce_all(0, big.ptr + 4, N, C.byref(mean)))), 4.0 * N)
print("sgemv")
for M, K in ((4096, 4096), (100_000, 1000), (1000, 100_000), (10, 10_000_000), (10_000_000, 10), (1, 100_000_000)):
    # bytes: the matrix, the vector (as long as a row: for 1 x 1e8 it weighs as much as the matrix) and the result
    line("sgemv %dx%d" % (M, K), run(lambda: check(lib.np_sgemv(M, K, big.ptr, big2.ptr, out.ptr))), 4.0 * (M * K + K + M))
