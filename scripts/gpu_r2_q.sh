#!/bin/bash
# phase cycles of the backward (library built with -DLP_PHASE_TIMING)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for v in ${1:-new}; do
  echo "== $v" | tee -a gpurun_out/q_phases.log
  LIGHTPLANE_AMD_LIB=$PWD/lightplane_amd/liblightplane_hip_pt_$v.so timeout 300 python scripts/phase_timing.py 2>&1 | tail -12 | tee -a gpurun_out/q_phases.log
done
