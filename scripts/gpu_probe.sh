#!/bin/bash
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_probe.hip -o /tmp/lap && /tmp/lap | tee gpurun_out/lds_atomic_probe.txt
