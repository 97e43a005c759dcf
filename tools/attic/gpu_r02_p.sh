#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d $O/p1 -o p1 --output-format csv -- python $R/tools/prof_fused.py 10 row > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD -d $O/p2 -o p2 --output-format csv -- python $R/tools/prof_fused.py 10 row > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/kt -o kt --output-format csv -- python $R/tools/prof_fused.py 20 row > $O/kt.log 2>&1
cd $R
python tools/pmc_summary.py $O/p1/*counter_collection.csv $O/p2/*counter_collection.csv 2>&1 | grep -v copyBuffer
python - <<'PY'
import csv,collections,statistics as st,os
d=collections.defaultdict(list)
for r in csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r02p/kt/kt_kernel_trace.csv')):
    d[r['Kernel_Name'][28:90]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print("%-64s n=%3d min %.1f med %.1f max %.1f us"%(k,len(v),min(v),st.median(v),max(v)))
PY
