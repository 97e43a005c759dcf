#!/bin/bash
for v in old new; do echo "== $v"; NP_HIP_LIB=$PWD/build/ab/libnp_hip_$v.so timeout 600 python tools/mid_rows_reduce_ab.py 2>&1 | tail -12; done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_statistics.py -x -q -m gpu -k "axis or reduc or stat" 2>&1 | tail -2
