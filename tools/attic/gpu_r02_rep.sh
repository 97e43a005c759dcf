#!/bin/bash
# one bench.py run (driver's arguments) on whatever box this call lands on; headline numbers only
mkdir -p gpurun_out/r02rep
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02rep/bench.json 2>/dev/null
python - <<'PY'
import json,socket
j=json.load(open('gpurun_out/r02rep/bench.json'))
s=j['secondary']['roofline']; e=j['extras']
print(json.dumps({"host": socket.gethostname(), "gemm_TF": round(j['value']/1e3,1), "gemm_frac": round(j['roofline']['frac'],3),
  "launch_ms_median": round(j['roofline']['launch_ms']['median'],4), "add_GBps": round(s['achieved']), "add_frac": round(s['frac'],3),
  "copy_GBps": round(s['ceiling']['copy_GBps']), "read_GBps": round(s['ceiling']['read_GBps']), "write_GBps": round(s['ceiling']['write_GBps']),
  "pow_frac": round(e['pow_1e8']['roofline']['frac'],3), "sum_axis0_frac": round(e['sum_axis0']['roofline']['frac'],3),
  "median_ms": round(e['median_1e8']['ms_per_launch'],4)}))
PY
