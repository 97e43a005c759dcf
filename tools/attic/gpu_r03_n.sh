#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03n
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_streamk.py -m gpu -q -x 2>&1 | tail -8 > $O/pytest.log; tail -4 $O/pytest.log
