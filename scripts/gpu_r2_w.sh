#!/bin/bash
# segmented Splatter march: parity (all splatter tests) + small-batch kernel times with / without segments
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "splat" > gpurun_out/w_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/w_tests.log
tail -3 gpurun_out/w_tests.log
(timeout 300 python scripts/bench_small_batch.py --splatter --reps 10; LP_SPLAT_SEGMENTS=1 timeout 300 python scripts/bench_small_batch.py --splatter --reps 10) 2>&1 | grep -v "Warn\|warn\|amdgpu" | tee gpurun_out/w_splat.log
