#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -25
for d in 0 2; do
  echo "v2 dbg=$d: $(LP_MFMA_DEBUG=$d timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done | tee gpurun_out/v2.txt
timeout 300 python scripts/ablate_bwd.py 2>&1 | tail -1 | tee -a gpurun_out/v2.txt
