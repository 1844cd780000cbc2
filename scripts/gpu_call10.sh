#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -5
LIGHTPLANE_AMD_GRAD_REPLICAS=7 timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "renderer" 2>&1 | tail -3
timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench_default.txt
