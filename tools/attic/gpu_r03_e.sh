#!/bin/bash
# round 3: pow (replicated LDS table) and fused column sink (in-kernel chunk fold): tests, bench, SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_libm_domain.py tests/test_gpu_statistics.py -m gpu -q 2>&1 | tail -15 > $O/pytest.log; tail -6 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p2 -o p2 --output-format csv -- python $R/tools/prof_r02.py 5 pow,cols,rows > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d $O/p1 -o p1 --output-format csv -- python $R/tools/prof_r02.py 5 pow,cols,rows > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/tools/prof_r02.py 10 pow,cols,rows > $O/kt.log 2>&1
cd $R
python tools/pmc_summary.py $O/p1/*counter_collection.csv $O/p2/*counter_collection.csv > $O/pmc_sq.txt 2>&1
grep -E "binary_vec_kernel<5|fused_chain_cols|fused_chain_rows" $O/pmc_sq.txt | cut -c1-420
head -12 $O/kt/*kernel_stats.csv | cut -c1-160
python - <<'PY'
import json
j = json.load(open('gpurun_out/r03e/bench.json'))
print(j['value'], j['roofline']['frac'])
for k, v in j.get('extras', {}).items():
    if isinstance(v, dict) and 'roofline' in v:
        print(k, round(v['ms_per_launch'], 4), round(v['roofline']['frac'], 3), v.get('parity_ok'))
PY
