#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export LIGHTPLANE_AMD_LIB=$GRAFT_REPO_ROOT/ab/libPT.so
( timeout 300 python scripts/phase_timing.py
  LP_BF3_BWD=1 timeout 300 python scripts/phase_timing.py ) 2>&1 | grep -v amdgpu.ids > gpurun_out/r2_phases.txt
cat gpurun_out/r2_phases.txt
