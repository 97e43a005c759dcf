#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03q
mkdir -p $O
cd $R
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open('gpurun_out/r03q/bench.json'))
print(round(j['value']), round(j['roofline']['frac'], 3), j['roofline'].get('launch_ms'))
for k, v in j.get('extras', {}).items():
    if isinstance(v, dict) and 'roofline' in v:
        print("   %-24s timed %.4f ms frac %.3f | median launch %.4f ms frac %.3f  %s" % (k, v['ms_per_launch'], v['roofline']['frac'], v['launch_ms']['median'], v['roofline']['frac_median_launch'], v.get('parity_ok')))
PY
