#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -I lightplane_amd/csrc scripts/scatter_plane_test.hip -o /tmp/spt 2>/dev/null
/tmp/spt > gpurun_out/r2_scatter_plane_test_fixed.txt 2>&1
tail -3 gpurun_out/r2_scatter_plane_test_fixed.txt
timeout 1500 python -m pytest tests/test_gpu_coherent.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_pytest_coherent2.txt 2>&1
grep -n "AssertionError:\|passed\|failed" gpurun_out/r2_pytest_coherent2.txt | tail -40
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --deselect tests/test_gpu_coherent.py > gpurun_out/r2_pytest_gpu_rest.txt 2>&1
tail -5 gpurun_out/r2_pytest_gpu_rest.txt
