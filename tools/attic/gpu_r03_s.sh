#!/bin/bash
# unaligned operands straight through the LDS-DMA kernel: parity tests, then odd-shape sweep against the pad-copy path
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03s; mkdir -p $O
cd $R
timeout 120 tools/explore/dma_unaligned > $O/dma_unaligned.log 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streamk.py tests/test_gpu_fuzz.py -x -q -m gpu -k "matmul or streamk or sgemm" 2>&1 | tail -15 > $O/pytest_matmul.log; tail -5 $O/pytest_matmul.log
NP_SWEEP_ODD=1 timeout 900 python tools/gemm_sweep.py > $O/gemm_unaligned.log 2>&1; cat $O/gemm_unaligned.log
