#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
LP_MFMA_DEBUG=32 timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "renderer or cfg2" 2>&1 | tail -5
for round in 1 2 3; do for d in 0 32; do
  echo "dbg=$d: $(LP_MFMA_DEBUG=$d timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done; done | tee gpurun_out/ab.txt
