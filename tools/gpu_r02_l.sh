#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 600 python tools/latency_ab.py > gpurun_out/r02l/latency_ab.log 2>&1
cat gpurun_out/r02l/latency_ab.log
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/r02l/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02l/pytest.log
tail -3 gpurun_out/r02l/pytest.log
