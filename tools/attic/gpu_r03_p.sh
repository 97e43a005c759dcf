#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03p
mkdir -p $O
cd $R
timeout 300 python tools/fused_cols_placement_probe.py > $O/fused_cols_placement.log 2>&1; cat $O/fused_cols_placement.log
