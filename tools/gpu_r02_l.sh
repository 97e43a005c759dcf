#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 600 python tools/op_rate.py > gpurun_out/r02l/op_rate.log 2>&1
cat gpurun_out/r02l/op_rate.log
