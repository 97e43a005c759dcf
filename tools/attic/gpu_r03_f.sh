#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03f
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_streamk.py -m gpu -q -x 2>&1 | tail -25 > $O/pytest_streamk.log; tail -12 $O/pytest_streamk.log
timeout 600 python tools/gemm_sweep.py > $O/gemm_sweep.log 2>&1; cat $O/gemm_sweep.log
timeout 300 python tools/pow_grid_ab.py > $O/pow_ab.log 2>&1; tail -30 $O/pow_ab.log
timeout 300 python tools/fused_cols_ab.py > $O/fused_cols_ab.log 2>&1; tail -14 $O/fused_cols_ab.log
