#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "splatter" 2>&1 | tail -4
LP_SPLAT_RPW=32 timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "splatter" 2>&1 | tail -2
timeout 600 python scripts/bench_extra.py cfg3 2>&1 | tail -1
LP_SPLAT_RPW=32 timeout 600 python scripts/bench_extra.py cfg3 2>&1 | tail -1
