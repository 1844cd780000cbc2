#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python scripts/bench_extra.py $@ 2>&1 | tail -4 | tee gpurun_out/bench_extra.txt
