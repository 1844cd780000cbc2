#!/bin/bash
# ONE gpurun call = a sequence of named steps (replaces the per-call scripts of rounds 1-2).
#
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash scripts/gpu_run.sh TAG step [step ...]'
#
# Every step writes gpurun_out/<TAG>_<step>.log (merged back into the build container) and prints its tail.  Steps:
#   tests[=<pytest -k expression>]     pytest -m gpu (-x), optionally narrowed
#   testsall[=<-k expression>]         the same without -x (every failure is listed)
#   pytest=<arguments>                 pytest -m gpu with these arguments (files, -k ..., -x ...); eval'ed: -k \"a or b\" works
#   testfile=<tests/file.py[::test]>   one test file / test
#   bench[=<bench.py arguments>]       python bench.py <arguments>            (default: the driver's command, no flags)
#   ab[=<bench.py arguments>]          every ab/lib*.so through bench.py, two alternating rounds (A/B of library builds)
#   profile[=<w1,w2,..>]               scripts/gpu_profile.sh for the workloads (default cfg2,cfg3,cfg4) + summarize_profiles.py
#   py=<script.py[ args]>              python scripts/<script.py> args        (benches, probes written in Python)
#   hip=<file.hip>                     hipcc scripts/<file.hip> && run it     (micro-benchmarks)
#   env:<VAR=value>                    export for the following steps; env:-VAR unsets
# A step's "=" argument may contain spaces when the whole step is quoted.
TAG=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
cd "$R"; mkdir -p gpurun_out
i=0
for step in "$@"; do
  i=$((i + 1))
  name=${step%%=*}; arg=""
  [[ "$step" == *=* ]] && arg=${step#*=}
  log=gpurun_out/${TAG}_${i}_${name%%:*}.log
  case "$name" in
    env:*) v=${step#env:}; if [[ "$v" == -* ]]; then unset "${v#-}"; else export "$v"; fi; echo "== env $v"; continue ;;
    tests) if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "$arg" > "$log" 2>&1;
           else timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > "$log" 2>&1; fi; echo "rc=$?" >> "$log" ;;
    testsall) timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${arg:+-k "$arg"} > "$log" 2>&1; echo "rc=$?" >> "$log" ;;
    pytest) eval "timeout 2400 python -m pytest $arg -m gpu -q --tb=short -p no:cacheprovider" > "$log" 2>&1; echo "rc=$?" >> "$log" ;;
    testfile) timeout 1500 python -m pytest $arg -m gpu -q -x --tb=short -p no:cacheprovider > "$log" 2>&1; echo "rc=$?" >> "$log" ;;
    bench) timeout 900 python bench.py $arg > "$log" 2>&1; echo "rc=$?" >> "$log" ;;
    ab) for round in 1 2; do for f in ab/lib*.so; do
          echo "$(basename $f): $(LIGHTPLANE_AMD_LIB=$PWD/$f timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras $arg 2>&1 | tail -1 |
            python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])' 2>&1 | tail -1)"
        done; done > "$log" 2>&1 ;;
    profile) bash scripts/gpu_profile.sh ${arg//,/ } > "$log" 2>&1; python scripts/summarize_profiles.py ${LP_ROUND:-r06} ${arg:-cfg2,cfg3,cfg4} >> "$log" 2>&1 ;;  # (only the workloads profiled here: the round's other entries stay)
    py) timeout 900 python scripts/$arg > "$log" 2>&1; echo "rc=$?" >> "$log" ;;
    hip) /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/$arg -o /tmp/probe_bin > "$log" 2>&1 && timeout 300 /tmp/probe_bin >> "$log" 2>&1 ;;
    *) echo "unknown step $step"; continue ;;
  esac
  echo "== $step -> $log"; grep -v "amdgpu.ids\|Warning\|warn" "$log" | tail -${LP_TAIL:-12} | cut -c1-${LP_COLS:-400}
done
