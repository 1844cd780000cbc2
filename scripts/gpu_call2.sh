#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for occ in 1 2; do for pipe in 0 1; do
  if [ $pipe = 1 ]; then export LP_MFMA_BWD_PIPE=1; else unset LP_MFMA_BWD_PIPE; fi
  echo "occ=$occ pipe=$pipe: $(LP_MFMA_BWD_OCC=$occ timeout 300 python scripts/ablate_bwd.py 2>&1 | tail -1)"
done; done | tee gpurun_out/ablate_matrix.txt
