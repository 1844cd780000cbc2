#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
( for f in bench_flex bench_two_grid bench_h64 bench_mlp_splatter bench_voxel bench_early_stop; do echo "== scripts/$f.py"; timeout 600 python scripts/$f.py 2>&1 | grep -v "amdgpu.ids\|UserWarning\|warnings.warn" | tail -12; done
  echo "== scripts/bench_extra.py cfg5"; timeout 900 python scripts/bench_extra.py cfg5 2>&1 | grep -v amdgpu.ids | tail -3 ) > gpurun_out/r2_other_workloads.txt 2>&1
cat gpurun_out/r2_other_workloads.txt
