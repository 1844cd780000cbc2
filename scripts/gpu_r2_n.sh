#!/bin/bash
# segment-parallel backward: parity tests, small-batch bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_coherent.py -q -k "segmented" > gpurun_out/n_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/n_tests.log
tail -15 gpurun_out/n_tests.log
timeout 600 python scripts/bench_small_batch.py > gpurun_out/n_small.log 2>&1; cat gpurun_out/n_small.log | tail -12
