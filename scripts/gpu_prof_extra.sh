#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_extra -o extra -- python $R/scripts/bench_extra.py $@ > $R/gpurun_out/prof_extra.txt 2>&1
f=$(find $R/gpurun_out/prof_extra -name "*kernel_stats.csv" | head -1); head -12 "$f" | cut -c1-220
