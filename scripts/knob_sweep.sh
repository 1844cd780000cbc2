#!/bin/bash
# env-knob sweeps on the BASELINE workloads (no rebuild): one line per setting
cd ${GRAFT_REPO_ROOT:-$PWD}
one() {  # label, workload, env...
  local label=$1 w=$2; shift 2
  echo "$label $(env "$@" python bench.py --workload $w --no-extras --no-cpu-baseline --steps 60 --warmup 5 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "fwd", d["fwd_ms"], "bwd", d["bwd_ms"])')"
}
for s in 1 2 3 4 6; do one "cfg3 LP_SPLAT_SEGMENTS=$s" cfg3 LP_SPLAT_SEGMENTS=$s; done
one "cfg3 default" cfg3 LP_DUMMY=1
for o in 2 3 4; do one "cfg2 LP_BF3_OCC=$o" cfg2 LP_BF3_OCC=$o; done
one "cfg2 default" cfg2 LP_DUMMY=1
for o in 2 3 4; do one "1080p_s128 LP_BF3_OCC=$o" 1080p_s128 LP_BF3_OCC=$o; done
