#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -3
for v in 3 4; do LP_MFMA_FWD_VARIANT=$v timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "renderer or cfg2" 2>&1 | tail -1; done
timeout 300 python scripts/fwd_variants.py 2>&1 | tail -1
timeout 600 python scripts/bench_extra.py cfg4s 2>&1 | tail -1
