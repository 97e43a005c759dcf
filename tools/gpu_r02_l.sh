#!/bin/bash
mkdir -p gpurun_out/r02l
for v in new; do echo "== $v"; NP_HIP_LIB=$PWD/build/ab/libnp_hip_$v.so timeout 600 python tools/fused_short_rows_ab.py; done > gpurun_out/r02l/fused_short_rows_ab2.log 2>&1
cat gpurun_out/r02l/fused_short_rows_ab2.log
timeout 900 python -m pytest tests/test_gpu_fusion.py -x -q -m gpu 2>&1 | tail -2
