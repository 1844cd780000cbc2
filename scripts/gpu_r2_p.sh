#!/bin/bash
# A/B bench against the previous library (lightplane_amd/liblightplane_hip_prev.so) + a parity subset on the new one
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; rm -f gpurun_out/p_bench.log
for lib in new prev new prev; do
  if [ $lib = prev ]; then export LIGHTPLANE_AMD_LIB=$PWD/lightplane_amd/liblightplane_hip_prev.so; else unset LIGHTPLANE_AMD_LIB; fi
  echo "== $lib" | tee -a gpurun_out/p_bench.log
  timeout 300 python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['fwd_ms'], d['bwd_ms'], d['roofline']['frac'])" | tee -a gpurun_out/p_bench.log
done
unset LIGHTPLANE_AMD_LIB
if [ "${1:-}" = tests ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_coherent.py -q -x -k "renderer or cfg or segmented" > gpurun_out/p_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/p_tests.log
tail -3 gpurun_out/p_tests.log
fi
