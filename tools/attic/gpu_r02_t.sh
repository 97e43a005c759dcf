#!/bin/bash
mkdir -p gpurun_out/r02t
timeout 900 python tools/transpose_ab.py 128,256064,64256,128064,64128,256032,32256,64,128 > gpurun_out/r02t/transpose_rect_ab.log 2>&1
cat gpurun_out/r02t/transpose_rect_ab.log
