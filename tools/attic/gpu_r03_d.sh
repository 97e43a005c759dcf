#!/bin/bash
# round 3: full -m gpu suite + bench line on the library with the device-side pipeline
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03d
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/pytest_gpu.log; tail -8 $O/pytest_gpu.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open('gpurun_out/r03d/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
for k, v in j.get('extras', {}).items():
    if isinstance(v, dict) and 'roofline' in v:
        print(k, round(v['ms_per_launch'], 4), round(v['roofline']['frac'], 3), v.get('parity_ok'))
    elif isinstance(v, dict):
        print(k, json.dumps(v)[:1800])
PY
