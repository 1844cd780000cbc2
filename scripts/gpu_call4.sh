#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --tb=line 2>&1 | tail -5
for r in 0 1 3 7 15 31; do
  echo "replicas=$r: $(LIGHTPLANE_AMD_GRAD_REPLICAS=$r timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
done | tee gpurun_out/replicas.txt
for r in 0 31; do echo "occ1 replicas=$r: $(LP_MFMA_BWD_OCC=1 LIGHTPLANE_AMD_GRAD_REPLICAS=$r timeout 300 python scripts/ablate_bwd.py 2>&1 | tail -1)"; done | tee -a gpurun_out/replicas.txt
