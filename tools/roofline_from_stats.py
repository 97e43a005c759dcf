"""Roofline fractions from a committed `rocprofv3 --kernel-trace --stats` summary of tools/prof_kernels.py
(profiles/rNN/prof_kernels_stats.csv): average duration of each workload's dominant kernel against its ALGORITHMIC bytes / flops
(the per-unit figures) — the numbers profiles/rNN/README.md quotes, recomputed.
Peaks: HBM 8 TB/s, dense fp32 MFMA 157.3 TFLOP/s (MI355X_MICROARCH.md).
Usage: python tools/roofline_from_stats.py [profiles/r06/prof_kernels_stats.csv]"""
import csv
import sys

HBM, MFMA = 8.0e12, 157.3e12
N, R, C = 10 ** 8, 25000, 4000
# (kernel name prefix after stripping "void " and the anonymous namespace, what it is, algorithmic bytes or flops, "hbm" | "mfma")
TABLE = [
    ("sgemm_dma_kernel<false, false, true>", "matmul 4096^3 (C2)", 2.0 * 4096 ** 3, "mfma"),
    ("binary_vec_kernel<0, 0, 0, false", "add 1e8 (C3a)", 12.0 * N, "hbm"),
    ("unary_vec_kernel<2,", "exp 1e8 (C3b)", 8.0 * N, "hbm"),
    ("unary_vec_kernel<5,", "log 1e8 (C3b)", 8.0 * N, "hbm"),
    ("binary_vec_kernel<0, 0, 2,", "X + row 25000x4000 (C3c)", 8.0 * N + 4.0 * C, "hbm"),
    ("binary_vec_kernel<0, 0, 3,", "X + col 25000x4000 (C3c)", 8.0 * N + 4.0 * R, "hbm"),
    ("reduce_axis_cols<0, false", "sum(axis 0) 65536x4096 (C4)", 4.0 * 65536 * 4096 + 4.0 * 4096, "hbm"),
    ("reduce_all_pass1<0", "sum 1e8", 4.0 * N, "hbm"),
    ("moments_pass1", "variance 1e8, the one read", 4.0 * N, "hbm"),
    ("weighted_sums_pass1", "weighted average 1e8", 8.0 * N, "hbm"),
    ("reduce_xform_pass1<4", "allclose 1e8", 8.0 * N, "hbm"),
    ("argreduce_rows_kernel<true>", "argmax 1e8", 4.0 * N, "hbm"),
    ("argreduce_rows_wave<true>", "argmax(axis 1) 65536x1024", 4.0 * 65536 * 1024 + 4.0 * 65536, "hbm"),
    ("sgemv_fewrows_chunks_kernel", "10 x 1e7 . 1e7", 4.0 * (10 * 10 ** 7 + 10 ** 7), "hbm"),
    ("transpose_tile_kernel<128, 128, true>", "transpose 65536x4096", 8.0 * 65536 * 4096, "hbm"),
    ("transpose_walign_kernel", "transpose 8191x8193", 8.0 * 8191 * 8193, "hbm"),
    ("permute_plane_kernel<4>", "permute (60, 128, 1024, 8) -> (0, 2, 1, 3)", 8.0 * 60 * 128 * 1024 * 8, "hbm"),
    ("cchain_flat_kernel<CChain<2, 258, 512>, -1>", "exp(a)*b+2 stored (compiled chain)", 12.0 * N, "hbm"),
    ("cchain_flat_kernel<CChain<2, 258, 512>, 0>", "sum(exp(a)*b+2) (compiled chain)", 8.0 * N, "hbm"),
    ("cchain_cols_kernel", "sum(exp(X), 0) (compiled chain)", 4.0 * N, "hbm"),
    ("cchain_rows_kernel", "sum(exp(X), 1) (compiled chain)", 4.0 * N, "hbm"),
]


def main(path):
    rows = list(csv.DictReader(open(path)))
    print("%-46s %-38s %10s %10s %7s" % ("kernel", "workload", "avg us", "rate", "frac"))
    for prefix, what, work, bound in TABLE:
        if work is None:
            continue
        for r in rows:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
            if name.startswith(prefix):
                us = float(r["AverageNs"]) / 1e3
                rate = work / (us * 1e-6)
                peak = HBM if bound == "hbm" else MFMA
                unit = "TB/s" if bound == "hbm" else "TFLOP/s"
                print("%-46s %-38s %10.1f %7.2f %-7s %.3f" % (prefix[:46], what, us, rate / 1e12, unit, rate / peak))
                break
        else:
            print("%-46s %-38s      (not in this summary)" % (prefix[:46], what))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r06/prof_kernels_stats.csv")
