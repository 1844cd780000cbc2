#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
LP_MFMA_DEBUG=${1:-0} timeout 300 python scripts/phase_timing.py 2>&1 | tail -12 | tee gpurun_out/phases.txt
