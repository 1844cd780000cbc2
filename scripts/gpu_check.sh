#!/bin/bash
# One gpurun call: atomics probe + GPU parity tests + short bench. Outputs under gpurun_out/.
set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -m2 -E "Marketing Name|gfx" > gpurun_out/gpu.txt 2>&1
hipcc --offload-arch=gfx950 -O3 scripts/atomics_probe.hip -o /tmp/atomics_probe && timeout 120 /tmp/atomics_probe > gpurun_out/atomics_probe.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 > gpurun_out/bench.txt 2>&1
tail -5 gpurun_out/bench.txt
cat gpurun_out/atomics_probe.txt
