#!/bin/bash
# session-2 call 1: parity tests, bench, ablation, atomics probe
mkdir -p gpurun_out
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -15 ) > gpurun_out/pytest_gpu.txt 2>&1
cat gpurun_out/pytest_gpu.txt
timeout 300 python bench.py --steps 10 --warmup 2 > gpurun_out/bench_default.txt 2>&1; tail -1 gpurun_out/bench_default.txt
LP_MFMA_BWD_PIPE=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_pipe.txt 2>&1; tail -1 gpurun_out/bench_pipe.txt
timeout 300 python scripts/ablate_bwd.py 2>&1 | tail -1 | tee gpurun_out/ablate.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/atomics_probe.hip -o /tmp/atomics_probe && timeout 120 /tmp/atomics_probe | tee gpurun_out/atomics_probe.txt
