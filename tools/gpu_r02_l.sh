#!/bin/bash
mkdir -p gpurun_out/r02l
timeout 900 python tools/misc_sweep.py > gpurun_out/r02l/misc_sweep.log 2>&1; echo rc=$?
cat gpurun_out/r02l/misc_sweep.log
