#!/bin/bash
# shallow decoders: family 1 (tuned bf16x3 default shape, fp32-MFMA flex / two-grid kernels) against the layer-looped family's
# shallow two-waves-per-SIMD backward (LP_LOOP=1), and that against its own deep instantiation (LP_LOOP_NO_SHALLOW=1)
cd ${GRAFT_REPO_ROOT:-$PWD}
echo "== default selection"; SHAPESET=shallow python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-230
echo "== LP_LOOP=1 (shallow looped backward)"; LP_LOOP=1 SHAPESET=shallow python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-230
echo "== LP_LOOP=1 LP_LOOP_NO_SHALLOW=1 (deep one-wave instantiation)"; LP_LOOP=1 LP_LOOP_NO_SHALLOW=1 SHAPES="2/2/2,0/2/2,1/1/1" SHAPESET=shallow python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-230
echo "== MLP-Splatter, two-layer MLPs: default selection"; SHAPESET=shallow python scripts/bench_shapes.py splatter 2>&1 | grep "^{" | cut -c1-230
echo "== LP_LOOP=1 (looped, two-waves-per-SIMD backward)"; LP_LOOP=1 SHAPESET=shallow python scripts/bench_shapes.py splatter 2>&1 | grep "^{" | cut -c1-230
echo "== LP_LOOP=1 LP_LOOP_NO_SHALLOW=1 (looped, deep instantiation)"; LP_LOOP=1 LP_LOOP_NO_SHALLOW=1 SHAPESET=shallow python scripts/bench_shapes.py splatter 2>&1 | grep "^{" | cut -c1-230
