#!/bin/bash
# One gpurun call: GPU parity tests + short bench (+ optional extras). Outputs under gpurun_out/.
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -30 > gpurun_out/pytest_gpu.txt
cat gpurun_out/pytest_gpu.txt
for occ in 1 2; do
  LP_MFMA_BWD_OCC=$occ timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_occ$occ.txt 2>&1
  echo "occ=$occ: $(tail -1 gpurun_out/bench_occ$occ.txt | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done
LP_MFMA_BWD_OCC=2 python scripts/ablate_bwd.py 2>&1 | tail -1
