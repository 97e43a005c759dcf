#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "beyond" 2>&1 | tail -15
