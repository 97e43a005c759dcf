#!/bin/bash
# round 2, GPU call A: counters for the GEMM / pow / add / fused-cols kernels, forced-dist vs plain bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02a
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_list.txt 2>&1
run_pmc() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" -d $O/$name -o $name --output-format csv -- python $R/tools/prof_r02.py 5 gemm,pow,add,cols,rows > $O/$name.log 2>&1
  echo "pass $name rc=$?"
}
run_pmc p1 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE
run_pmc p2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run_pmc p3 SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/tools/prof_r02.py 20 gemm,pow,add,cols,rows > $O/kt.log 2>&1
cd $R
for i in 1 2; do
  python bench.py --no-extras --steps 20 --warmup 5 > $O/bench_plain_$i.json 2> $O/bench_plain_$i.err
  NP_BENCH_FORCE_DIST=1 python bench.py --no-extras --steps 20 --warmup 5 > $O/bench_dist_$i.json 2> $O/bench_dist_$i.err
done
python tools/pmc_summary.py $O/p*/*counter_collection.csv > $O/pmc_summary.txt 2>&1
tail -n 40 $O/pmc_summary.txt
grep -h value $O/bench_*.json | cut -c1-200
