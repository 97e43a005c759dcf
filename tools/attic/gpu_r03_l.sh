#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r03l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS -d $O/p2 -o p2 --output-format csv -- python $R/tools/rows_pmc_probe.py > $O/p2.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS -d $O/p1 -o p1 --output-format csv -- python $R/tools/rows_pmc_probe.py > $O/p1.log 2>&1
cd $R
python - <<'PY'
import csv, collections, glob
for path in sorted(glob.glob('gpurun_out/r03l/p*/*counter_collection.csv')):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        if 'fused_chain_rows' not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:60], r.get('Grid_Size', ''))
        d[key][r['Counter_Name']].append(float(r['Counter_Value']))
    print('==', path)
    for k, cs in d.items():
        print(' ', k, ' '.join('%s=%.4g' % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
PY
