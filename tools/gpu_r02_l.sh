#!/bin/bash
mkdir -p gpurun_out/r02l
for seed in 11 12; do
timeout 1400 python tools/fuzz_parity.py 1200 $seed > gpurun_out/r02l/fuzz_$seed.log 2>&1; echo "seed $seed rc=$?"; tail -2 gpurun_out/r02l/fuzz_$seed.log
done
