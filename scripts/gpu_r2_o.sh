#!/bin/bash
# full GPU suite + smoke + default bench on the final library
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/o_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/o_tests.log
tail -8 gpurun_out/o_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" > gpurun_out/o_smoke.log 2>&1; tail -2 gpurun_out/o_smoke.log
timeout 900 python bench.py > gpurun_out/o_bench.log 2>&1; tail -1 gpurun_out/o_bench.log
