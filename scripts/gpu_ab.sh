#!/bin/bash
# A/B timing of two library builds in ONE box: ab/libA.so vs ab/libB.so (3 alternating rounds)
export TMPDIR=/tmp
mkdir -p gpurun_out
for round in 1 2 3; do for v in A B; do
  echo "$v: $(LIGHTPLANE_AMD_LIB=$PWD/ab/lib$v.so LP_MFMA_DEBUG=${1:-0} timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done; done | tee gpurun_out/ab.txt
