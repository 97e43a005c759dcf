#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03j
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fuzz.py -m gpu -q -x 2>&1 | tail -6 > $O/pytest.log; tail -3 $O/pytest.log
NP_PROBE_VARIANTS=4000,0,4000,0 timeout 600 python tools/fused_ragged_rows_probe.py > $O/fused_mid_rows_ab.log 2>&1; cat $O/fused_mid_rows_ab.log
