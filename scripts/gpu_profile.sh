#!/bin/bash
# rocprofv3 kernel trace of bench.py + backward ablation. Outputs under gpurun_out/.
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python scripts/ablate_bwd.py > gpurun_out/ablate.txt 2>&1; tail -2 gpurun_out/ablate.txt
cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.txt 2>&1
cd $GRAFT_REPO_ROOT; tail -2 gpurun_out/prof_bench.txt; ls -R gpurun_out/prof | head -20
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); echo "== $f"; head -12 "$f"
