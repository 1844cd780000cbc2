#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -15
for occ in 1 2; do for d in 0 1 2; do
  echo "occ=$occ dbg=$d: $(LP_MFMA_BWD_OCC=$occ LP_MFMA_DEBUG=$d timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done; done | tee gpurun_out/dbg.txt
