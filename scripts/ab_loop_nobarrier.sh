#!/bin/bash
# timing experiment: the looped Renderer backwards that run ONE wave per SIMD (deep 4/4/4 x 32, two-block 2/2/2 x 64) with every
# s_barrier of their layer phases compiled out (ab/libnobar.so: -DLP_EXPERIMENTS -DLP_X_LOOP_NO_BARRIER, wrong weight gradients) against
# the product library -- how much of their time is the barrier coupling of the four waves of a workgroup
cd ${GRAFT_REPO_ROOT:-$PWD}
for lib in product nobar; do
  echo "== $lib"
  if [ $lib = nobar ]; then export LIGHTPLANE_AMD_LIB=$PWD/ab/libnobar.so LIGHTPLANE_AMD_ALLOW_EXPERIMENTAL=1; fi
  SHAPES="4/4/4" python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-180
  SHAPESET=h64 SHAPES="2/2/2" python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-180
done
