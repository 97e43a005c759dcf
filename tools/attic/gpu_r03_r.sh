#!/bin/bash
# pow with the log2 table in registers (ds_bpermute) against the per-wave LDS copy: bit check, timing, SQ counters
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03r; mkdir -p $O
cd $R
timeout 600 python tools/pow_grid_ab.py > $O/pow_regtab_ab.log 2>&1
for v in 0 9000; do
  NP_PROF_POW_VARIANT=$v timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $O/p$v -o p --output-format csv -- python $R/tools/prof_r02.py 5 pow > $O/p$v.log 2>&1
  NP_PROF_POW_VARIANT=$v timeout 300 rocprofv3 --pmc SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM -d $O/q$v -o q --output-format csv -- python $R/tools/prof_r02.py 5 pow > $O/q$v.log 2>&1
  python tools/pmc_summary.py $O/p$v/*counter_collection.csv $O/q$v/*counter_collection.csv > $O/pmc_sq_pow_variant$v.txt 2>&1
done
tail -20 $O/pow_regtab_ab.log
