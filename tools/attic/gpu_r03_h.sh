#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03h
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_streamk.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest_streamk.log; tail -4 $O/pytest_streamk.log
timeout 900 python tools/gemm_sweep.py > $O/gemm_sweep.log 2>&1; cat $O/gemm_sweep.log
