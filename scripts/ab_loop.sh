#!/bin/bash
# A/B of library builds (ab/lib*.so) on the layer-looped family: LP_LOOP=1 bench.py cfg2 + bench_shapes (renderer), two alternating rounds
cd ${GRAFT_REPO_ROOT:-$PWD}
for round in 1 2; do for f in ab/lib*.so; do
  echo "== $(basename $f) round $round"
  LP_LOOP=1 LIGHTPLANE_AMD_LIB=$PWD/$f python bench.py --workload cfg2 --steps 50 --warmup 5 --no-cpu-baseline --no-extras 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("LP_LOOP=1 cfg2:", d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])'
  if [ $round = 1 ]; then
    LP_LOOP=1 LIGHTPLANE_AMD_LIB=$PWD/$f SHAPES="2/2/2,4/4/4,1/1/1,0/2/2,2/4/2" python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-230
    LIGHTPLANE_AMD_LIB=$PWD/$f SHAPES="1/1/1,0/2/2" python scripts/bench_shapes.py renderer 2>&1 | grep "^{" | cut -c1-230
  fi
done; done
