#!/bin/bash
for v in old new; do echo "== $v"; NP_HIP_LIB=$PWD/build/ab/libnp_hip_$v.so timeout 600 python tools/fused_ragged_rows_probe.py 2>&1 | tail -7; done
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_fuzz.py -x -q -m gpu -k "fus or chain" 2>&1 | tail -2
