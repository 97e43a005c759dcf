#!/bin/bash
# round 2, GPU call G: why does bench.py's add read 3 % below the same kernel in a fresh process?
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02g
mkdir -p $O
cd $R
NP_BENCH_DIAG=1 python bench.py > $O/bench_diag.json 2> $O/bench_diag.err; grep diag $O/bench_diag.err
NP_BENCH_DIAG=1 NP_BENCH_DIAG_NOCPU=1 python bench.py > $O/bench_diag_nocpu.json 2> $O/bench_diag_nocpu.err; grep diag $O/bench_diag_nocpu.err
python tools/fused_cols_ab.py > $O/fused_cols_ab.log 2>&1; grep -E "round|rows kernel|variant    0|variant 1004|variant 1006|variant 1008" $O/fused_cols_ab.log
NP_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 > $O/bench_torch_world1.json 2> $O/bench_torch_world1.err; python -c "
import json; j=json.load(open('$O/bench_torch_world1.json')); print(json.dumps(j.get('extras'))[:1200])"
