#!/bin/bash
mkdir -p gpurun_out/r02y
FUSED_AB_REDUCE=1 FUSED_AB_RBPCS=8,16,32,64,128,200 timeout 900 python tools/fused_ab.py > gpurun_out/r02y/fused_reduce_ab.log 2>&1
cat gpurun_out/r02y/fused_reduce_ab.log
