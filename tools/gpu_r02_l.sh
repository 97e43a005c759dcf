#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_edge_cases.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -4
NP_FUZZ_CASES=800 NP_FUZZ_SEED=21 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k broadcast 2>&1 | tail -2
timeout 300 python tools/ragged_ab.py 2>&1 | tail -9
