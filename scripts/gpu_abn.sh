#!/bin/bash
# timing of every ab/lib*.so in ONE box (2 alternating rounds)
export TMPDIR=/tmp
mkdir -p gpurun_out
for round in 1 2; do for f in ab/lib*.so; do
  echo "$(basename $f): $(LIGHTPLANE_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])' 2>&1 | tail -1)"
done; done | tee gpurun_out/abn.txt
