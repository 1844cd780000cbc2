#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
for round in 1 2; do for v in A B; do for gg in 0 1; do
  if [ $gg = 1 ]; then export LP_MFMA_GENERIC_GRIDS=1; else unset LP_MFMA_GENERIC_GRIDS; fi
  echo "$v generic=$gg: $(LIGHTPLANE_AMD_LIB=$PWD/ab/lib$v.so timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done; done; done | tee gpurun_out/ab.txt
