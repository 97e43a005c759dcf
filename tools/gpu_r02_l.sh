#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 600 python -m pytest tests/test_gpu_result_wait.py -x -q -m gpu 2>&1 | tail -15
