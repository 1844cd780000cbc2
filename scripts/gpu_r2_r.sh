#!/bin/bash
# A/B of two libraries over the bench's extras (all workloads): $1 / $2 = library suffixes ("" = the default library)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/r_bench.log
for lib in "$1" "$2"; do
  if [ -n "$lib" ]; then export LIGHTPLANE_AMD_LIB=$PWD/lightplane_amd/liblightplane_hip_$lib.so; else unset LIGHTPLANE_AMD_LIB; fi
  echo "== ${lib:-default}" | tee -a gpurun_out/r_bench.log
  timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readlines()[-1])
print('cfg2', d['value'], d['fwd_ms'], d['bwd_ms'], d['roofline']['frac'])
for k,v in d['extras'].items(): print(k, v.get('fwd_ms'), v.get('bwd_ms'), v.get('Mrays_per_s_fwd_bwd'), v.get('bwd_ms_one_sweep_per_ray'))
" | tee -a gpurun_out/r_bench.log
done
