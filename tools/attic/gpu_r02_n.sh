#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02n
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/tools/prof_fused.py 20 > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_WAVES -d $O/p1 -o p1 --output-format csv -- python $R/tools/prof_fused.py 5 > $O/p1.log 2>&1
cd $R
python tools/pmc_summary.py $O/p1/*counter_collection.csv > $O/pmc_fused.txt 2>&1; cat $O/pmc_fused.txt
python - <<'PY'
import csv,collections,statistics as st,os
d=collections.defaultdict(list); info={}
for r in csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r02n/kt/kt_kernel_trace.csv')):
    k=r['Kernel_Name'][28:100]
    d[k].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3); info[k]=(r['VGPR_Count'],r['SGPR_Count'],r['Grid_Size_X'],r['LDS_Block_Size'])
for k,v in d.items(): print("%-74s n=%3d min %.1f med %.1f max %.1f us  vgpr/sgpr/grid/lds %s"%(k,len(v),min(v),st.median(v),max(v),info[k]))
PY
