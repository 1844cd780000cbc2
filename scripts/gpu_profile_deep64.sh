#!/bin/bash
# rocprofv3 of the shape-generic Renderer kernels on deep hidden-64 decoders (SHAPESET=deep64 scripts/bench_shapes.py, 147 456 rays):
# one kernel trace + one counter pass -> gpurun_out/prof_deep64/, gpurun_out/pmc1_deep64/ (condensed into profiles/r06_kernel_stats_deep64_generic.csv
# and profiles/r06_generic_kernels.txt by hand).
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp; export TMPDIR=/tmp
export SHAPESET=deep64 NPIX=${NPIX:-384}
P="python $R/scripts/bench_shapes.py renderer"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_deep64 -o bench -- $P > $R/gpurun_out/prof_deep64.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc1_deep64 -o pmc -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3_deep64 -o pmc -- $P > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc4_deep64 -o pmc -- $P > /dev/null 2>&1
find $R/gpurun_out/prof_deep64 -name "*kernel_stats.csv" | head -1 | xargs head -8 | cut -c1-200
# keep the merge small: the per-dispatch traces of the counter passes are large
find $R/gpurun_out/pmc1_deep64 $R/gpurun_out/pmc3_deep64 $R/gpurun_out/pmc4_deep64 -name "*kernel_trace.csv" -delete
