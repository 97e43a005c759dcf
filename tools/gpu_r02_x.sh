#!/bin/bash
# round 2, call x: full GPU suite after the last-workgroup folds + bracket select, then a bench line
mkdir -p gpurun_out/r02x
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02x/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02x/pytest.log
tail -4 gpurun_out/r02x/pytest.log
python bench.py > gpurun_out/r02x/bench.json 2> gpurun_out/r02x/bench.err; echo "bench rc=$?"
python -c "
import json; j=json.load(open('gpurun_out/r02x/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
print(json.dumps(j['secondary']['roofline']))
for k,v in j['extras'].items():
    if isinstance(v,dict) and 'roofline' in v: print(k, round(v['ms_per_launch'],4), round(v['roofline']['frac'],3), v.get('parity_ok'))
print(json.dumps(j['extras']['median_1e8']))
"
