#!/bin/bash
# bf16x3 backward for C = 32 by default: parity (all Renderer / config-scale / module tests) + cfg4 / cfg2 bench
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x -k "renderer or cfg or segmented or module or sweep or golden or fit" > gpurun_out/u_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/u_tests.log
tail -4 gpurun_out/u_tests.log
for w in cfg4 cfg2; do timeout 300 python bench.py --workload $w --no-cpu-baseline --no-extras 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$w', d['value'], d['fwd_ms'], d['bwd_ms'], d['roofline']['frac'])"; done
