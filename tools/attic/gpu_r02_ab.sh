#!/bin/bash
# same-box A/B of two builds of libnp_hip.so through bench.py's extras
mkdir -p gpurun_out/r02ab
for rnd in 1 2; do for v in old new; do
NP_BENCH_DIAG_NOCPU=1 NP_HIP_LIB=$PWD/build/ab/libnp_hip_$v.so python bench.py > gpurun_out/r02ab/bench_${v}_$rnd.json 2>/dev/null
python - $v $rnd <<'PY'
import json,sys
v,r=sys.argv[1],sys.argv[2]
j=json.load(open('gpurun_out/r02ab/bench_%s_%s.json'%(v,r)))
keys=["exp_1e8","fused_chain_1e8","fused_chain_sum_1e8","exp_plus_row_fused","exp_plus_col_fused","sum_exp_axis1_fused","sum_exp_axis0_fused","add_1e8"]
print(v,r," ".join("%s=%.4f"%(k.replace('_fused','').replace('_1e8',''),j['extras'][k]['ms_per_launch']) for k in keys),flush=True)
PY
done; done
