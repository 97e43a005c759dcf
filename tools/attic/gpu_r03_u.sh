#!/bin/bash
# stream-K fold through LDS-DMA: parity, then the short shape sweep
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03u; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_streamk.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py -x -q -m gpu -k "matmul or streamk or sgemm" 2>&1 | tail -15 > $O/pytest_matmul.log; tail -5 $O/pytest_matmul.log
NP_SWEEP_SHORT=1 timeout 900 python tools/gemm_sweep.py > $O/gemm_sweep_short.log 2>&1; cat $O/gemm_sweep_short.log
