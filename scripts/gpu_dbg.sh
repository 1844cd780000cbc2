#!/bin/bash
# usage: gpu_dbg.sh "0 2 4 6"   -> bench with LP_MFMA_DEBUG=<each>
mkdir -p gpurun_out
export TMPDIR=/tmp
for d in $1; do
  echo "dbg=$d: $(LP_MFMA_DEBUG=$d timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done | tee gpurun_out/dbg.txt
