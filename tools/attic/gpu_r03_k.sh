#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03k
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log; tail -3 $O/pytest.log
NP_PROBE_VARIANTS=4001,0,4000 timeout 600 python tools/fused_ragged_rows_probe.py > $O/fused_mid_rows_light_ab.log 2>&1; cat $O/fused_mid_rows_light_ab.log
timeout 300 python tools/fused_short_rows_ab.py > $O/fused_short_rows.log 2>&1; tail -12 $O/fused_short_rows.log
