#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02t
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2 3; do echo "== process $i"; python tools/add_slab_lib.py 2>&1 | head -3; done
for i in 1 2; do python bench.py > $O/bench_$i.json 2> $O/bench_$i.err; python -c "
import json; j=json.load(open('gpurun_out/r02t/bench_$i.json'))
print(j['value'], j['roofline']['frac'])
print(json.dumps(j['secondary']['roofline']['ceiling']))
print(' '.join('%s %.3f' % (k, v['roofline']['frac']) for k,v in j['extras'].items() if isinstance(v,dict) and 'roofline' in v))
"; done
