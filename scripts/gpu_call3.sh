#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/atomics_probe.hip -o /tmp/atomics_probe && timeout 300 /tmp/atomics_probe | tee gpurun_out/atomics_probe2.txt
