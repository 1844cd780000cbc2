#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I lightplane_amd/csrc scripts/scatter_plane_test.hip -o /tmp/spt 2>/dev/null
/tmp/spt > gpurun_out/r2_scatter_plane_test.txt 2>&1
cat gpurun_out/r2_scatter_plane_test.txt
