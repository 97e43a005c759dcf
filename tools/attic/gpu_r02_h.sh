#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02h
mkdir -p $O
cd $R
python tools/gemm_prio_ab.py > $O/gemm_prio_ab.log 2>&1; cat $O/gemm_prio_ab.log
timeout 200 tools/explore/add_bw offsets x > $O/add_offsets.log 2>&1; cat $O/add_offsets.log
python tools/fused_cols_ab.py > $O/fused_cols_ab.log 2>&1; grep -E "round|rows kernel|variant    0|variant 1004|variant 1008" $O/fused_cols_ab.log
