#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/tr_b16_probe.hip -o /tmp/trp && /tmp/trp > gpurun_out/r2_tr_b16_probe.txt 2>&1
head -40 gpurun_out/r2_tr_b16_probe.txt
