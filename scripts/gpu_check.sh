#!/bin/bash
# One gpurun call: GPU parity tests + short bench (+ optional extras). Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -60 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --steps ${BENCH_STEPS:-10} --warmup 2 ${BENCH_ARGS:-} > gpurun_out/bench.txt 2>&1
tail -3 gpurun_out/bench.txt
