#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite on the new library, the bench line, the overlap trace, the GEMM sweep
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03a
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_fuzz.py 2>&1 | tail -25 > $O/pytest_gpu.log; echo "pytest rc=${PIPESTATUS[0]}" >> $O/pytest_gpu.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -40 > $O/pytest_fuzz.log
tail -5 $O/pytest_gpu.log; tail -5 $O/pytest_fuzz.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/overlap -o ov --output-format csv -- python $R/tools/overlap_trace.py 5 4 > $O/overlap_trace.log 2>&1
cd $R
python tools/overlap_trace_report.py $O/overlap/*kernel_trace.csv > $O/overlap_report.txt 2>&1
cat $O/overlap_trace.log | tail -6; cat $O/overlap_report.txt
timeout 600 python tools/gemm_sweep.py > $O/gemm_sweep_planner.log 2>&1; tail -30 $O/gemm_sweep_planner.log
python - <<'PY'
import json
j = json.load(open('gpurun_out/r03a/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
for k, v in j.get('extras', {}).items():
    if isinstance(v, dict) and 'roofline' in v:
        print(k, round(v['ms_per_launch'], 4), round(v['roofline']['frac'], 3), v.get('parity_ok'))
    elif isinstance(v, dict):
        print(k, json.dumps(v)[:1500])
PY
