#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03o
mkdir -p $O
cd $R
timeout 300 python tools/fused_cols_ab.py > $O/fused_cols_rows_in_flight_ab.log 2>&1; cat $O/fused_cols_rows_in_flight_ab.log
