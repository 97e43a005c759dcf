#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02i
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
NP_BENCH_DIAG=1 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; grep diag $O/bench.err
python tools/gemm_prio_ab.py > $O/gemm_prio_ab.log 2>&1; cat $O/gemm_prio_ab.log
python tools/add_ramp.py 2>&1 | head -3
python -c "
import json; j=json.load(open('gpurun_out/r02i/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
print(json.dumps(j['secondary']['roofline']))
for k,v in j['extras'].items():
    if isinstance(v,dict) and 'roofline' in v: print(k, round(v['ms_per_launch'],4), round(v['roofline']['frac'],3), v.get('parity_ok'))
"
