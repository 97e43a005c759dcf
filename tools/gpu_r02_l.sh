#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests/test_gpu_order_stats.py tests/test_gpu_fuzz.py -x -q -m gpu -k "order or stat" 2>&1 | tail -4
timeout 1500 python tools/select_large_fuzz.py 12 1 > gpurun_out/r02l/select_large_fuzz.log 2>&1; echo rc=$?
cat gpurun_out/r02l/select_large_fuzz.log
SELECT_AB_CASES=uniform01,signed_wide,sorted SELECT_AB_VARIANTS=0,1 timeout 300 python tools/select_ab.py 2>&1 | tail -9
