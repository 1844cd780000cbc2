#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for d in ${1:-0 16}; do echo "== dbg=$d"; LP_MFMA_DEBUG=$d timeout 300 python scripts/phase_timing.py 2>&1 | tail -11; done | tee gpurun_out/phases.txt
