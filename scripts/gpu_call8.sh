#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -15
bash scripts/gpu_dbg.sh "0 2"
