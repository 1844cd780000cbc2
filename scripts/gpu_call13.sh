#!/bin/bash
export TMPDIR=/tmp
for d in 0 1 2 3; do echo "dbg=$d $(LP_SPLAT_DEBUG=$d timeout 600 python scripts/bench_extra.py cfg3 2>&1 | tail -1 | cut -c1-120)"; done
