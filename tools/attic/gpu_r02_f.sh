#!/bin/bash
# round 2, GPU call F: LDS-staged pow table, add ramp experiment, explorer back-to-back A/B, np_comm, abi bench
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r02f
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
python tools/pow_ab.py > $O/pow_ab.log 2>&1; cat $O/pow_ab.log | grep base
python tools/add_ramp.py > $O/add_ramp.log 2>&1; cat $O/add_ramp.log
timeout 200 tools/explore/add_bw lds > $O/add_lds_dma_ab.log 2>&1; grep -E "round|grid  mode0 U(1|2|4) |ldsdma U(1|2|4) T256 nts1 aux2|mode[123]" $O/add_lds_dma_ab.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
NP_BENCH_FORCE_DIST=1 NP_COMM=abi python bench.py --steps 20 --warmup 5 > $O/bench_abi_world1.json 2> $O/bench_abi_world1.err; echo "bench abi rc=$?"; cut -c1-300 $O/bench_abi_world1.json; tail -3 $O/bench_abi_world1.err
NP_BENCH_FORCE_DIST=1 python bench.py --steps 20 --warmup 5 > $O/bench_torch_world1.json 2> $O/bench_torch_world1.err; echo "bench torch world1 rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python $R/tools/prof_r02.py 30 pow,add > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $O/p1 -o p1 --output-format csv -- python $R/tools/prof_r02.py 5 pow,add > $O/p1.log 2>&1
cd $R
python tools/pmc_summary.py $O/p1/*counter_collection.csv > $O/pmc_pow.txt 2>&1; cat $O/pmc_pow.txt
python - <<'PY'
import csv,collections,statistics as st,os
d=collections.defaultdict(list)
for r in csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/r02f/kt/kt_kernel_trace.csv')):
    d[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items(): print("%-62s n=%3d min %.1f med %.1f max %.1f mean %.1f sd %.1f us"%(k,len(v),min(v),st.median(v),max(v),st.mean(v),st.pstdev(v)))
PY
python -c "
import json; j=json.load(open('gpurun_out/r02f/bench.json'))
print(j['value'], j['roofline']['frac'], j['roofline'].get('launch_ms'))
print(json.dumps(j['secondary']['roofline']))
for k,v in j['extras'].items():
    if isinstance(v,dict) and 'roofline' in v: print(k, round(v['ms_per_launch'],4), round(v['roofline']['frac'],3), v.get('parity_ok'))
for f in ('bench_abi_world1','bench_torch_world1'):
    j=json.load(open('gpurun_out/r02f/%s.json'%f)); print(f, j['value'], json.dumps(j.get('extras'))[:900])
"
