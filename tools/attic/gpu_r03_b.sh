#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03b
mkdir -p $O
cd $R
timeout 300 python tools/chunk_overhead.py > $O/chunk_overhead.log 2>&1; cat $O/chunk_overhead.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
timeout 1500 python -m pytest tests -m gpu -q --deselect tests/test_gpu_fuzz.py 2>&1 | tail -40 > $O/pytest_gpu.log; tail -15 $O/pytest_gpu.log
timeout 900 python -m pytest tests/test_gpu_fuzz.py -m gpu -q 2>&1 | tail -40 > $O/pytest_fuzz.log; tail -8 $O/pytest_fuzz.log
