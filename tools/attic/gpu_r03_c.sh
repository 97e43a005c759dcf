#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03c
mkdir -p $O
cd $R
timeout 300 python -m pytest tests/test_gpu_comm.py tests/test_gpu_fusion.py tests/test_gpu_method_bodies.py -m gpu -q -x 2>&1 | tail -30 > $O/pytest_a.log; tail -12 $O/pytest_a.log
timeout 300 python tools/chunk_overhead.py > $O/chunk_overhead.log 2>&1; cat $O/chunk_overhead.log | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/overlap -o ov --output-format csv -- python $R/tools/overlap_trace.py 5 4 > $O/overlap_trace.log 2>&1
cd $R
python tools/overlap_trace_report.py $O/overlap/*kernel_trace.csv > $O/overlap_report.txt 2>&1
grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|rocprofv3" $O/overlap_trace.log | tail -6; cat $O/overlap_report.txt
