#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03g
mkdir -p $O
cd $R
NP_SWEEP_SHORT=1 timeout 600 python tools/gemm_sweep.py > $O/gemm_sweep.log 2>&1; cat $O/gemm_sweep.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sk_one.py <<'PY'
import sys
sys.path.insert(0, "/root/repo")
from numpower_amd import device as D
from numpower_amd._lib import load, check
D.init(0); lib = load()
for n in (3000, 2048):
    a = D.DeviceArray((n, n)); b = D.DeviceArray((n, n)); c = D.DeviceArray((n, n))
    D.fill(a, 0.5); D.fill(b, 0.25); D.unary("sin", a, out=a); D.unary("cos", b, out=b)
    for v in (-5, -4):
        check(lib.np_sgemm_set_variant(v))
        for _ in range(10): D.sgemm(a, b, out=c)
        D.sync()
PY
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python /tmp/sk_one.py > $O/kt.log 2>&1
head -8 $O/kt/*kernel_stats.csv | cut -c1-230
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/p1 -o p1 --output-format csv -- python /tmp/sk_one.py > $O/p1.log 2>&1
cd $R; python tools/pmc_summary.py $O/p1/*counter_collection.csv 2>&1 | grep -i "sgemm" | cut -c1-420
