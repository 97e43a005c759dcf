#!/bin/bash
# peeling thin ragged edges: parity tests, then whole / peeled / default across "one past" shapes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03t; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_streamk.py tests/test_gpu_fuzz.py -x -q -m gpu -k "matmul or streamk or sgemm or few_rows or sgemv or outer or dot" 2>&1 | tail -15 > $O/pytest_matmul.log; tail -5 $O/pytest_matmul.log
timeout 300 python tools/gemm_fringe_probe.py > $O/gemm_fringe_probe.log 2>&1
NP_SWEEP_PEEL=1 timeout 900 python tools/gemm_sweep.py > $O/gemm_peel.log 2>&1; cat $O/gemm_peel.log
