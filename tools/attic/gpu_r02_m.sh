#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02m
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('gpurun_out/r02m/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
print(json.dumps(j['secondary']['roofline']))
for k,v in j['extras'].items():
    if isinstance(v,dict) and 'roofline' in v: print(k, round(v['ms_per_launch'],4), round(v['roofline']['frac'],3), v.get('parity_ok'))
"
python tools/reduce_sweep.py > $O/reduce_sweep.log 2>&1; tail -30 $O/reduce_sweep.log
