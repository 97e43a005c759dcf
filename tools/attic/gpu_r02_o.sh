#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02v
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/fused_cols_ab.py 2>&1 | grep -E "round 2|rows kernel|variant 1004" | tail -3 
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('gpurun_out/r02v/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
print(json.dumps(j['secondary']['roofline']))
for k,v in j['extras'].items():
    if isinstance(v,dict) and 'roofline' in v: print(k, round(v['ms_per_launch'],4), round(v['roofline']['frac'],3), v.get('parity_ok'), v.get('parity_max_rel_err'))
"
