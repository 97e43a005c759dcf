#!/bin/bash
mkdir -p gpurun_out/r02w
SELECT_AB_VARIANTS=0,1 timeout 600 python tools/select_ab.py > gpurun_out/r02w/grid.log 2>&1
cat gpurun_out/r02w/grid.log
