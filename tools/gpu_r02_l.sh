#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_statistics.py tests/test_gpu_argreduce.py -x -q -m gpu 2>&1 | tail -3
NP_FUZZ_CASES=500 NP_FUZZ_SEED=41 timeout 900 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "axis" 2>&1 | tail -2
timeout 300 python tools/short_rows_reduce_ab.py 2>&1 | tail -8
