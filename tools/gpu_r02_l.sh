#!/bin/bash
timeout 600 python tools/fused_ragged_rows_probe.py 2>&1 | tail -11
