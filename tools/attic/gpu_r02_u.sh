#!/bin/bash
# round 2, call u: method_bodies (C program over the host boundary) + bracket-path order statistics
mkdir -p gpurun_out/r02u
timeout 900 python -m pytest tests/test_gpu_method_bodies.py tests/test_gpu_order_stats.py tests/test_gpu_statistics.py -x -q -m gpu > gpurun_out/r02u/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r02u/pytest.log
timeout 600 python tools/select_ab.py > gpurun_out/r02u/select_ab.log 2>&1
tail -5 gpurun_out/r02u/pytest.log
cat gpurun_out/r02u/select_ab.log
