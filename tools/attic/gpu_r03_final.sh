#!/bin/bash
# round 3, evidence run on the final library: the whole -m gpu suite, the bench line, kernel-trace stats of the same
# command, PMC passes (HBM traffic, MFMA / SQ counters), the GEMM shape sweep, the N > 1 code paths at world size 1
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03final
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/pytest_gpu.log; tail -4 $O/pytest_gpu.log
python bench.py > $O/bench_r03.json 2> $O/bench_r03.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 > $O/bench_r03_driver_args.json 2> $O/bench_r03_driver_args.err; echo "bench(driver args) rc=$?"
NP_BENCH_FORCE_DIST=1 timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_r03_world1_torch.json 2> $O/w1t.err; echo "world1 torch rc=$?"
NP_BENCH_FORCE_DIST=1 NP_COMM=abi timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_r03_world1_abi.json 2> $O/w1a.err; echo "world1 abi rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/kt -o bench --output-format csv -- python $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python $R/tools/prof_kernels.py 3 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python $R/tools/prof_kernels.py 3 > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_WAVES GRBM_GUI_ACTIVE -d $O/p1 -o p1 --output-format csv -- python $R/tools/prof_r02.py 5 gemm,pow,add,cols,rows > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -o p2 --output-format csv -- python $R/tools/prof_r02.py 5 gemm,pow,add,cols,rows > $O/p2.log 2>&1
cd $R
python tools/pmc_summary.py $O/p1/*counter_collection.csv $O/p2/*counter_collection.csv > $O/pmc_sq_final.txt 2>&1
python tools/pmc_summary.py $O/fetch/*counter_collection.csv $O/write/*counter_collection.csv > $O/pmc_summary.txt 2>&1
python tools/pmc_traffic.py $O/fetch/*counter_collection.csv $O/write/*counter_collection.csv $O/pmc_traffic.json
python tools/gemm_pmc_json.py $O/gemm_pmc.json $O/p1/*counter_collection.csv $O/p2/*counter_collection.csv
cp $O/kt/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
head -14 $O/bench_kernel_stats.csv | cut -c1-200
timeout 900 python tools/gemm_sweep.py > $O/gemm_sweep_final.log 2>&1; cat $O/gemm_sweep_final.log
python - <<'PY'
import json
for name in ("bench_r03.json", "bench_r03_driver_args.json", "bench_r03_world1_torch.json", "bench_r03_world1_abi.json"):
    try:
        j = json.load(open('gpurun_out/r03final/' + name))
    except Exception as e:
        print(name, "unreadable", e); continue
    print(name, round(j['value']), round(j['roofline']['frac'], 3), j['roofline'].get('launch_ms', {}).get('median'))
    for k, v in j.get('extras', {}).items():
        if isinstance(v, dict) and 'roofline' in v:
            print("   ", k, round(v['ms_per_launch'], 4), round(v['roofline']['frac'], 3), v.get('parity_ok'))
        elif isinstance(v, dict) and 'ms_per_step' in v:
            print("   ", k, {kk: round(vv, 3) for kk, vv in v['ms_per_step'].items()}, v.get('parity_ok'))
        else:
            print("   ", k, str(v)[:300])
PY
