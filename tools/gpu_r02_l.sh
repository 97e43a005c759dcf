#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 900 python -m pytest tests/test_gpu_order_stats.py tests/test_gpu_statistics.py tests/test_gpu_fuzz.py -x -q -m gpu > gpurun_out/r02l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02l/pytest.log
tail -15 gpurun_out/r02l/pytest.log
timeout 600 python tools/latency_ab.py > gpurun_out/r02l/latency_ab.log 2>&1
cat gpurun_out/r02l/latency_ab.log
