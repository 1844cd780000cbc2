#!/bin/bash
# two-block (hidden 64) looped kernels against the wide fp32-MFMA family (default selection until 0.2.3) on 2/2/2 x 64:
# eight-wave forward workgroups (one copy of the 97 KB images per CU, two waves per SIMD) / four-wave (LP_LOOP_FWD_NW4=1);
# 256x256 rays (march kernels) and 64x64 rays (segment-parallel small-batch kernels)
cd ${GRAFT_REPO_ROOT:-$PWD}
for npix in 256 64; do
echo "== ${npix}x${npix} rays: default selection"; NPIX=$npix SHAPESET=h64 python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-200
echo "== ${npix}x${npix} rays: LP_LOOP=1 (looped, eight-wave forward)"; NPIX=$npix LP_LOOP=1 SHAPESET=h64 python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-200
done
echo "== 256x256 rays: LP_LOOP=1 LP_LOOP_FWD_NW4=1 (looped, four-wave forward)"; LP_LOOP=1 LP_LOOP_FWD_NW4=1 SHAPESET=h64 python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-200
