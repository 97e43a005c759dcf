#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03m
mkdir -p $O
cd $R
true
NP_PROBE_VARIANTS=5000,0 timeout 600 python tools/fused_ragged_rows_probe.py > $O/fused_rows_u4_ab.log 2>&1; cat $O/fused_rows_u4_ab.log
