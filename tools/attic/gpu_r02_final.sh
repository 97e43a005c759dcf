#!/bin/bash
# round 2, final evidence run: bench line, kernel-trace stats of the same command, PMC passes (HBM traffic, MFMA / SQ counters)
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02final
mkdir -p $O
cd $R
python bench.py > $O/bench_r02.json 2> $O/bench_r02.err; echo "bench rc=$?"
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
cp $O/kt/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
head -30 $O/bench_kernel_stats.csv | cut -c1-200
grep -E "sgemm|binary_vec_kernel<5" $O/pmc_sq_final.txt | cut -c1-400
python -c "
import json; j=json.load(open('gpurun_out/r02final/bench_r02.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'), j['roofline'].get('mfma_busy'))
print(json.dumps(j['secondary']['roofline']))
for k,v in j['extras'].items():
    if isinstance(v,dict) and 'roofline' in v: print(k, round(v['ms_per_launch'],4), round(v['roofline']['frac'],3), v.get('parity_ok'))
"
