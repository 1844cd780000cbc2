#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -x 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("cfg2", d["value"], "Mrays/s fwd", d["fwd_ms"], "bwd", d["bwd_ms"])'
timeout 600 python scripts/bench_extra.py cfg4s 2>&1 | tail -1
HID=64 timeout 300 python scripts/bench_h64.py 2>&1 | tail -1
timeout 300 python scripts/bench_mlp_splatter.py 2>&1 | tail -1
