#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/$1.hip -o /tmp/probe_bin && /tmp/probe_bin | tee gpurun_out/$1.txt
