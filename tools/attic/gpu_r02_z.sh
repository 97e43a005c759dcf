#!/bin/bash
mkdir -p gpurun_out/r02z
NP_FUZZ_CASES=300 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "order_statistics" > gpurun_out/r02z/fuzz.log 2>&1
tail -5 gpurun_out/r02z/fuzz.log
NP_FUZZ_CASES=300 NP_FUZZ_SEED=7 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -k "order_statistics" >> gpurun_out/r02z/fuzz.log 2>&1
tail -3 gpurun_out/r02z/fuzz.log
