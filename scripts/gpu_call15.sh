#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "h64" 2>&1 | tail -3
HID=64 timeout 600 python scripts/bench_h64.py 2>&1 | tail -1
