#!/bin/bash
NP_FUZZ_CASES=800 NP_FUZZ_SEED=61 timeout 1200 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -3
