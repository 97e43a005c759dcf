#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "sgemv" 2>&1 | tail -3
timeout 300 python tools/sgemv_ab.py 2>&1 | tail -10
