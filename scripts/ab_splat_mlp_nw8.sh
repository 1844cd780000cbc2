#!/bin/bash
# MLP-Splatter forward of hidden-64 MLPs with three / four layers: eight-wave workgroups (one copy of the 72 / 96 KB of limb images
# per CU, two waves per SIMD) against four-wave workgroups (LP_LOOP_FWD_NW4=1, one wave per SIMD)
cd ${GRAFT_REPO_ROOT:-$PWD}
echo "== default (eight-wave forward where the images exclude a second workgroup)"; python scripts/bench_shapes.py splatter 2>&1 | grep "^{" | cut -c1-200
echo "== LP_LOOP_FWD_NW4=1 (four-wave forward)"; LP_LOOP_FWD_NW4=1 python scripts/bench_shapes.py splatter 2>&1 | grep "^{" | cut -c1-200
