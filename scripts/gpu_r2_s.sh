#!/bin/bash
# other kernel families under two libraries: hidden-64 Renderer, flex shapes, two-grid decoder, MLP-Splatter
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/s_bench.log
for lib in "$1" "$2"; do
  if [ -n "$lib" ]; then export LIGHTPLANE_AMD_LIB=$PWD/lightplane_amd/liblightplane_hip_$lib.so; else unset LIGHTPLANE_AMD_LIB; fi
  echo "== ${lib:-default}" | tee -a gpurun_out/s_bench.log
  for sc in bench_h64.py bench_flex.py bench_two_grid.py bench_mlp_splatter.py; do
    echo "-- $sc" | tee -a gpurun_out/s_bench.log
    timeout 300 python scripts/$sc 2>&1 | grep -v Warning | tail -6 | tee -a gpurun_out/s_bench.log
  done
done
