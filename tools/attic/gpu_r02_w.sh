#!/bin/bash
mkdir -p gpurun_out/r02w
SELECT_AB_CASES=uniform01,signed_wide SELECT_AB_VARIANTS=1,2,1,2 timeout 600 python tools/select_ab.py > gpurun_out/r02w/grid.log 2>&1
cat gpurun_out/r02w/grid.log
