#!/bin/bash
# One gpurun call: GPU parity tests + the headline bench + the other configurations.  Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -30 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/bench_default.txt 2>&1; tail -1 gpurun_out/bench_default.txt
timeout 900 python scripts/bench_extra.py 2>&1 | grep '^{' | tee gpurun_out/bench_extra.txt
HID=64 timeout 300 python scripts/bench_h64.py 2>&1 | tail -1 | tee -a gpurun_out/bench_extra.txt
