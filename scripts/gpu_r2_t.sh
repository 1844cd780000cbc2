#!/bin/bash
# cfg2 bench over several builds of the library: default + liblightplane_hip_<suffix>.so for every suffix given
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/t_bench.log
for rep in 1 2; do
for lib in "" "$@"; do
  if [ -n "$lib" ]; then export LIGHTPLANE_AMD_LIB=$PWD/lightplane_amd/liblightplane_hip_$lib.so; else unset LIGHTPLANE_AMD_LIB; fi
  echo -n "${lib:-default} " | tee -a gpurun_out/t_bench.log
  timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['fwd_ms'], d['bwd_ms'], d['roofline']['frac'])" | tee -a gpurun_out/t_bench.log
done
done
