#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short ${1:-} 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.txt
