#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r2_pytest_gpu_e.txt 2>&1
grep -n "AssertionError:\|Error\b.*:\|passed\|failed" gpurun_out/r2_pytest_gpu_e.txt | tail -40
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r2_bench_e.txt 2>&1
tail -1 gpurun_out/r2_bench_e.txt | cut -c1-1200
