"""Copy what a `tools/gpu_lease.sh evidence <tag>; tools/gpu_lease.sh after <tag>` call left under gpurun_out/<tag>/ into
profiles/rNN/ under the names that directory uses, and write `roofline_from_stats.txt` (tools/roofline_from_stats.py over the
copied kernel stats) next to them.  NOT a measurement; it only moves files.
Usage: python tools/adopt_lease.py gpurun_out/r06g profiles/r06 r06"""
import contextlib
import io
import os
import shutil
import sys

RENAME = {
    "bench.json": "bench_{r}.json",
    "bench_driver_args.json": "bench_{r}_driver_args.json",
    "bench_under_rocprof.json": "bench_{r}_under_rocprof.json",
    "bench_world1_abi.json": "bench_{r}_world1_abi.json",
    "bench_world1_torch.json": "bench_{r}_world1_torch.json",
    "cleanbuild.log": "cleanbuild_on_gpu_box.log",
    "gemm_mid_plans.log": "gemm_plans.log",
}
SAME = [
    "bench_two_ranks_one_gpu_abi.json", "bench_kernel_stats.csv", "prof_kernels_stats.csv", "prof_sections.json", "pmc_traffic.json",
    "pmc_summary.txt", "pmc_sq.txt", "gemm_pmc.json", "stamp.json", "pytest_gpu.log", "pytest_lazy_binding.log", "pytest_peers.log",
    "lazy_bodies.log", "sq_chain_probe.log", "fuzz_parity.log", "gemm_mid_fuzz.log", "fused_static_fuzz.log", "gemm_deep_k_fuzz.log",
    "arg_sweep.log", "layout_sweep.log", "misc_sweep.log", "gemm_sweep.log", "gemm_deep_k_ab.log", "gemm_deep_k_sweep.log",
    "gemm_thin_k_ab.log", "gemm_thin_fill_ab.log", "reduce_small_ab.log", "fused_static_ab.log",
]


def main(src, dst, r):
    os.makedirs(dst, exist_ok=True)
    done = []
    for name in SAME + list(RENAME):
        p = os.path.join(src, name)
        if not os.path.exists(p):
            print("missing in the lease:", name)
            continue
        out = os.path.join(dst, RENAME.get(name, name).format(r=r))
        shutil.copyfile(p, out)
        done.append(out)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import roofline_from_stats
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        roofline_from_stats.main(os.path.join(dst, "prof_kernels_stats.csv"))
    open(os.path.join(dst, "roofline_from_stats.txt"), "w").write(buf.getvalue())
    print("%d files -> %s (+ roofline_from_stats.txt)" % (len(done), dst))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "r06")
