#!/bin/bash
mkdir -p gpurun_out/r02l
for v in old new old new; do echo "== $v"; NP_HIP_LIB=$PWD/build/ab/libnp_hip_$v.so timeout 600 python tools/ragged_reduce_ab.py; done > gpurun_out/r02l/ragged_reduce_ab.log 2>&1
cat gpurun_out/r02l/ragged_reduce_ab.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_statistics.py tests/test_gpu_fusion.py -x -q -m gpu 2>&1 | tail -3
