#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "splatter" 2>&1 | tail -3
LP_SPLAT_RPW=32 timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "splatter" 2>&1 | tail -1
timeout 600 python scripts/bench_extra.py cfg3 2>&1 | tail -1
LP_SPLAT_RPW=32 timeout 600 python scripts/bench_extra.py cfg3 2>&1 | tail -1
