#!/bin/bash
# Tuning-knob matrix for the MFMA backward (no parity tests). Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
for occ in 1 2; do for pipe in 0 1; do
  if [ $pipe = 1 ]; then export LP_MFMA_BWD_PIPE=1; else unset LP_MFMA_BWD_PIPE; fi
  LP_MFMA_BWD_OCC=$occ timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_o${occ}p${pipe}.txt 2>&1
  echo "occ=$occ pipe=$pipe: $(tail -1 gpurun_out/bench_o${occ}p${pipe}.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done; done
