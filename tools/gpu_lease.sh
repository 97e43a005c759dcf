#!/bin/bash
# What runs on a GPU lease, in one place (rounds 2 and 3 kept one script per lease; those left the tree in round 6).
#   gpurun --timeout 1500 -- 'bash tools/gpu_lease.sh <recipe> [tag]'
# Recipes (outputs under gpurun_out/<tag>/, tag defaults to the recipe name; copy what should be judged to profiles/rNN/):
#   tests      the -m gpu suite (the gate)
#   bench      bench.py twice: default flags and the driver's (--steps 20 --warmup 5)
#   world1     bench.py's N > 1 code paths on one GPU (torch.distributed plumbing, then the C-ABI communicator)
#   peers      bench.py --gpus 2 with both ranks on this GPU over tests/loopback_rccl (NP_COMM=abi): the N > 1 path with a peer
#   profile    rocprofv3 --kernel-trace --stats of bench.py -> bench_kernel_stats.csv
#   counters   the PMC passes (their own runs, as the guide prescribes): HBM traffic (FETCH_SIZE / WRITE_SIZE ->
#              pmc_traffic.json), MFMA / SQ counters of the headline kernels (-> gemm_pmc.json, pmc_sq.txt)
#   sweeps     GEMM shape sweeps, the layout and misc sweeps, the compiled-chain A/B
#   evidence   all of the above, in that order (the round's evidence run)
#   fuzz       the long random-shape parity sweeps (tests/test_gpu_fuzz.py x 400, GEMM tiles / splits, compiled chains)
#   binding    the GPU suite with NP_LAZY_BINDING=1, lazy_bodies, the square-step chains, the peer-process tests (verbose)
#   after      cleanbuild + fuzz + binding: what follows the evidence lease on the same kernel stamp
#   cleanbuild a copy of the SOURCES (no prebuilt library, no object files) built from scratch with the box's own hipcc, then
#              smoke() and a slice of the GPU suite against THAT build (the leases otherwise run the .so files pushed from the
#              build container: VERDICT r03 #15)
R=${GRAFT_REPO_ROOT:-/root/repo}
RECIPE=${1:-tests}
TAG=${2:-$RECIPE}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd "$R" || exit 1
python tools/source_stamp.py --json > "$O/stamp.json"

r_tests() {
    timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > "$O/pytest_gpu.log"
    tail -4 "$O/pytest_gpu.log"
}
r_bench() {
    python bench.py > "$O/bench.json" 2> "$O/bench.err"; echo "bench rc=$?"
    python bench.py --steps 20 --warmup 5 > "$O/bench_driver_args.json" 2> "$O/bench_driver_args.err"; echo "bench (driver args) rc=$?"
    python - "$O" <<'PY'
import json, sys
for name in ("bench.json", "bench_driver_args.json"):
    try:
        j = json.load(open(sys.argv[1] + "/" + name))
    except Exception as e:
        print(name, "unreadable", e); continue
    print(name, round(j["value"]), "GFLOP/s  frac", round(j["roofline"]["frac"], 3), " add", j["roofline"].get("secondary_frac"))
    print("   ", json.dumps(j.get("summary", {}))[:1500])
PY
}
r_world1() {
    NP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 > "$O/bench_world1_torch.json" 2> "$O/w1t.err"; echo "world1 torch rc=$?"
    NP_BENCH_FORCE_DIST=1 NP_COMM=abi timeout 600 python bench.py --steps 20 --warmup 5 > "$O/bench_world1_abi.json" 2> "$O/w1a.err"; echo "world1 abi rc=$?"
}
r_peers() {
    # the N > 1 bench with two ranks that SHARE this GPU, over the tests' stand-in for librccl.so.1 (tests/loopback_rccl): every
    # leg of config 5 through np_comm_* with a real peer process; "valid": false in the line - a code-path run, not a measurement
    LD_LIBRARY_PATH="$R/tests/loopback_rccl/lib:$LD_LIBRARY_PATH" NP_COMM=abi NP_BENCH_SHARE_DEVICE=1 timeout 700 python bench.py --gpus 2 --steps 10 --warmup 3 \
        > "$O/bench_two_ranks_one_gpu_abi.json" 2> "$O/peers.err"; echo "peers (2 ranks, one GPU, loopback) rc=$?"
}
r_profile() {
    (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$O/kt" -o bench --output-format csv -- python "$R/bench.py" > "$O/bench_under_rocprof.json" 2> "$O/bench_under_rocprof.err")
    cp "$O"/kt/*kernel_stats.csv "$O/bench_kernel_stats.csv" 2>/dev/null
    head -16 "$O/bench_kernel_stats.csv" | cut -c1-200
    # the same kernels launched at the BASELINE sizes ONLY (bench.py also runs add / sum at 1000 x 1000 for config 1, which share
    # kernel names with the 1e8 launches and pull their average down): per-kernel averages that reproduce the bench fractions
    # ... each kept running for >= 250 ms, so the averages are those of warm kernels (tools/prof_kernels.py)
    (cd /tmp && export TMPDIR=/tmp && NP_PROF_SECTIONS="$O/prof_sections_kt.json" timeout 900 rocprofv3 --kernel-trace --stats -d "$O/kt2" -o k --output-format csv -- python "$R/tools/prof_kernels.py" 20 250 > "$O/prof_kernels.log" 2>&1)
    cp "$O"/kt2/*kernel_stats.csv "$O/prof_kernels_stats.csv" 2>/dev/null
    head -20 "$O/prof_kernels_stats.csv" | cut -c1-200
}
r_counters() {
    (cd /tmp && export TMPDIR=/tmp
     NP_PROF_SECTIONS="$O/prof_sections.json" timeout 400 rocprofv3 --pmc FETCH_SIZE -d "$O/fetch" -o f --output-format csv -- python "$R/tools/prof_kernels.py" 3 > "$O/fetch.log" 2>&1
     NP_PROF_SECTIONS="$O/prof_sections_w.json" timeout 400 rocprofv3 --pmc WRITE_SIZE -d "$O/write" -o w --output-format csv -- python "$R/tools/prof_kernels.py" 3 > "$O/write.log" 2>&1
     timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE -d "$O/p1" -o p1 --output-format csv -- python "$R/tools/prof_counters.py" 5 gemm,pow,add,cols,rows > "$O/p1.log" 2>&1
     timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d "$O/p2" -o p2 --output-format csv -- python "$R/tools/prof_counters.py" 5 gemm,pow,add,cols,rows > "$O/p2.log" 2>&1)
    python tools/pmc_summary.py "$O"/p1/*counter_collection.csv "$O"/p2/*counter_collection.csv > "$O/pmc_sq.txt" 2>&1
    python tools/pmc_summary.py "$O"/fetch/*counter_collection.csv "$O"/write/*counter_collection.csv > "$O/pmc_summary.txt" 2>&1
    # HBM bytes per CALL of every workload of the bench: the CSVs cut at prof_kernels.py's marker launches (tools/pmc_traffic.py)
    cmp -s "$O/prof_sections.json" "$O/prof_sections_w.json" || echo "WARNING: the two PMC passes ran different sections"
    python tools/pmc_traffic.py "$O"/fetch/*counter_collection.csv "$O"/write/*counter_collection.csv "$O/pmc_traffic.json" "$O/prof_sections.json"
    python tools/gemm_pmc_json.py "$O/gemm_pmc.json" "$O"/p1/*counter_collection.csv "$O"/p2/*counter_collection.csv
    # both carry the stamp of the sources they were measured on (tools/source_stamp.py; the same lease's kernel stats sit next to them)
    python - "$O" <<'PY'
import json, sys
sys.path.insert(0, "tools")
from source_stamp import stamp
st = stamp()
for name in ("pmc_traffic.json", "gemm_pmc.json"):
    path = sys.argv[1] + "/" + name
    try:
        j = json.load(open(path))
    except Exception as e:
        print(name, "unreadable", e); continue
    j["source_sha16"], j["kernel_sha16"] = st["source_sha16"], st["kernel_sha16"]
    json.dump(j, open(path, "w"), indent=1)
PY
    cat "$O/pmc_sq.txt" | cut -c1-260
}
r_sweeps() {
    timeout 600 python tools/gemm_sweep.py > "$O/gemm_sweep.log" 2>&1; tail -30 "$O/gemm_sweep.log"
    timeout 300 python tools/gemm_mid_sweep.py plans > "$O/gemm_mid_plans.log" 2>&1; cat "$O/gemm_mid_plans.log"
    timeout 300 python tools/misc_sweep.py > "$O/misc_sweep.log" 2>&1
    timeout 200 python tools/layout_sweep.py > "$O/layout_sweep.log" 2>&1; cat "$O/layout_sweep.log"
    timeout 200 python tools/fused_static_ab.py 2 > "$O/fused_static_ab.log" 2>&1; tail -14 "$O/fused_static_ab.log"
    timeout 200 python tools/arg_sweep.py > "$O/arg_sweep.log" 2>&1; cat "$O/arg_sweep.log"
    timeout 200 python tools/gemm_thin_k_ab.py > "$O/gemm_thin_k_ab.log" 2>&1; cat "$O/gemm_thin_k_ab.log"
    timeout 200 python tools/reduce_small_ab.py > "$O/reduce_small_ab.log" 2>&1; cut -c1-300 "$O/reduce_small_ab.log"
    timeout 300 python tools/gemm_deep_k_ab.py > "$O/gemm_deep_k_ab.log" 2>&1; cut -c1-200 "$O/gemm_deep_k_ab.log"
    timeout 300 python tools/gemm_deep_k_sweep.py > "$O/gemm_deep_k_sweep.log" 2>&1; tail -5 "$O/gemm_deep_k_sweep.log"
    timeout 300 python tools/gemm_thin_fill_ab.py > "$O/gemm_thin_fill_ab.log" 2>&1; cut -c1-200 "$O/gemm_thin_fill_ab.log"
}

r_cleanbuild() {
    C=/tmp/np_clean_build
    rm -rf "$C"; mkdir -p "$C"
    tar -C "$R" --exclude='./gpurun_out' --exclude='./build' --exclude='./profiles' --exclude='*.so' --exclude='*.o' --exclude='__pycache__' \
        --exclude='./numpower_amd/lib' --exclude='./oracle/lib' -cf - . | tar -C "$C" -xf -
    cd "$C" || exit 1
    ls numpower_amd/lib 2>/dev/null && echo "UNEXPECTED: prebuilt libraries in the clean copy"
    start=$(date +%s)
    python __graft_entry__.py --smoke > "$O/cleanbuild.log" 2>&1; echo "clean build + smoke rc=$? in $(( $(date +%s) - start )) s"
    tail -3 "$O/cleanbuild.log"
    timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fusion.py tests/test_gpu_gemm_mid.py -x -q 2>&1 | tail -3 | tee -a "$O/cleanbuild.log"
    python - <<'PY' | tee -a "$O/cleanbuild.log"
import sys
sys.path.insert(0, ".")
from numpower_amd import _lib
print("library under test:", _lib.lib_path())
PY
    cd "$R"
}

r_fuzz() {
    # the long random-shape sweeps on the library as shipped (cases per family / seed offsets differ from the test tier's)
    timeout 900 python tools/fuzz_parity.py 400 6 2>&1 | tail -3 > "$O/fuzz_parity.log"; cat "$O/fuzz_parity.log"
    timeout 400 python tools/gemm_mid_fuzz.py 400 2 2>&1 | tail -3 > "$O/gemm_mid_fuzz.log"; cat "$O/gemm_mid_fuzz.log"
    timeout 400 python tools/fused_static_fuzz.py 400 2 2>&1 | tail -3 > "$O/fused_static_fuzz.log"; cat "$O/fused_static_fuzz.log"
    timeout 400 python tools/gemm_deep_k_fuzz.py 200 3 2>&1 | tail -4 > "$O/gemm_deep_k_fuzz.log"; cat "$O/gemm_deep_k_fuzz.log"
}

r_binding() {
    # what the round built behind the binding and around the communicator, on the final library: the GPU suite with the Python stand-in
    # appending to pending chains, the inserted text as a program (with its random programs), the chains with a
    # square step, and the peer-process tests one by one
    NP_LAZY_BINDING=1 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > "$O/pytest_lazy_binding.log"; tail -2 "$O/pytest_lazy_binding.log"
    timeout 300 numpower_amd/lib/lazy_bodies gpu /tmp/lazy_bodies.bin > "$O/lazy_bodies.log" 2>&1; tail -3 "$O/lazy_bodies.log"
    timeout 300 python tools/sq_chain_probe.py 2 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > "$O/sq_chain_probe.log"; tail -3 "$O/sq_chain_probe.log"
    timeout 600 python -m pytest tests/test_gpu_comm_loopback_peers.py tests/test_gpu_comm_multi.py -v --durations=0 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -40 > "$O/pytest_peers.log"; tail -3 "$O/pytest_peers.log"
}

case "$RECIPE" in
    tests) r_tests ;;
    bench) r_bench ;;
    world1) r_world1 ;;
    peers) r_peers ;;
    profile) r_profile ;;
    counters) r_counters ;;
    sweeps) r_sweeps ;;
    evidence) r_tests; r_bench; r_world1; r_peers; r_profile; r_counters; r_sweeps ;;
    cleanbuild) r_cleanbuild ;;
    fuzz) r_fuzz ;;
    binding) r_binding ;;
    after) r_cleanbuild; r_fuzz; r_binding ;;
    *) echo "unknown recipe $RECIPE (tests | bench | world1 | peers | profile | counters | sweeps | evidence | cleanbuild | fuzz | binding | after)"; exit 2 ;;
esac
