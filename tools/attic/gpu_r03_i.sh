#!/bin/bash
# round 3: long random-shape parity sweep + awkward-shape throughput sweep on the final library + stream-K SQ counters
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03i
mkdir -p $O
cd $R
timeout 1500 python tools/fuzz_parity.py 400 3 > $O/fuzz_parity.log 2>&1; tail -15 $O/fuzz_parity.log
timeout 900 python tools/misc_sweep.py > $O/misc_sweep.log 2>&1; tail -5 $O/misc_sweep.log
cd /tmp && export TMPDIR=/tmp
cat > /tmp/sk_pmc.py <<'PY'
import sys
sys.path.insert(0, "/root/repo")
from numpower_amd import device as D
from numpower_amd._lib import load, check
D.init(0); lib = load()
for n in (3072, 3000):
    a = D.DeviceArray((n, n)); b = D.DeviceArray((n, n)); c = D.DeviceArray((n, n))
    D.fill(a, 0.5); D.fill(b, 0.25); D.unary("sin", a, out=a); D.unary("cos", b, out=b)
    for v in (-5, -4):
        check(lib.np_sgemm_set_variant(v))
        for _ in range(6): D.sgemm(a, b, out=c)
        D.sync()
PY
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/p1 -o p1 --output-format csv -- python /tmp/sk_pmc.py > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python /tmp/sk_pmc.py > $O/f.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python /tmp/sk_pmc.py > $O/w.log 2>&1
cd $R; python tools/pmc_summary.py $O/p1/*counter_collection.csv $O/f/*counter_collection.csv $O/w/*counter_collection.csv > $O/pmc_streamk.txt 2>&1; grep -i "sgemm\|reduce_axis\|==" $O/pmc_streamk.txt | cut -c1-330
