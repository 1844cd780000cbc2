#!/bin/bash
# rocprofv3 of bench.py: per workload (cfg2 = the headline command without the `extras` runs, cfg3, cfg4) one kernel trace
# and PMC passes (own runs: --pmc is never combined with other trace domains).  Outputs under gpurun_out/prof_<w>, pmcN_<w>;
# scripts/summarize_profiles.py condenses them into profiles/.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
WL=${@:-cfg2 cfg3 cfg4}   # workloads to (re-)profile
for w in $WL; do rm -rf $R/gpurun_out/prof_$w $R/gpurun_out/prof_$w.txt $R/gpurun_out/pmc[1-4]_$w; done
unset LP_LOOP
run() {  # workload, steps (trace), steps (pmc)
  w=$1
  B="python $R/bench.py --workload $w --no-cpu-baseline --no-extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$w -o bench -- $B --steps $2 --warmup 3 > $R/gpurun_out/prof_$w.txt 2>&1
  tail -1 $R/gpurun_out/prof_$w.txt | cut -c1-300
  P="$B --steps $3 --warmup 1"
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc1_$w -o pmc -- $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2_$w -o pmc -- $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3_$w -o pmc -- $P > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc4_$w -o pmc -- $P > /dev/null 2>&1
}
for w in $WL; do
  case $w in
    cfg2) run cfg2 200 3 ;;
    cfg3) run cfg3 50 3 ;;
    cfg4) run cfg4 5 1 ;;
    small) run small 200 3 ;;
    1080p_s128) run 1080p_s128 5 1 ;;
    h64_222) run h64_222 20 1 ;;             # 2/2/2 x 64 on the two-block looped kernels (eight-wave forward)
    h64_example_112) run h64_example_112 20 1 ;;
    cfg5) run cfg5 3 1 ;;                    # BASELINE configs[4] at its per-GPU size (a step takes ~0.27 s): kernel trace + counters of the
                                             # render / splat kernels on the 256^3 x 32 grid (2.15 GB: the grid does NOT fit the caches)
    refbench256) run refbench256 20 2 ;;
    cfg5_render) run cfg5_render 5 1 ;;      # the render leg of cfg 5 alone (voxel 256^3 x 32 ch)     # the reference benchmark's 256^2 row: 65 536 RANDOM rays (incoherent scatter)
    loop)  # the headline workload through the layer-looped family (LP_LOOP=1: shallow two-waves-per-SIMD backward)
      export LP_LOOP=1
      rm -rf $R/gpurun_out/prof_loop $R/gpurun_out/pmc[1-4]_loop
      B="python $R/bench.py --workload cfg2 --no-cpu-baseline --no-extras"
      rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_loop -o bench -- $B --steps 200 --warmup 3 > $R/gpurun_out/prof_loop.txt 2>&1
      tail -1 $R/gpurun_out/prof_loop.txt | cut -c1-300
      P="$B --steps 3 --warmup 1"
      rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc1_loop -o pmc -- $P > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/pmc2_loop -o pmc -- $P > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3_loop -o pmc -- $P > /dev/null 2>&1
      rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/pmc4_loop -o pmc -- $P > /dev/null 2>&1
      unset LP_LOOP ;;
  esac
done
find $R/gpurun_out/prof_* $R/gpurun_out/pmc[1-4]_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" -size +512k -delete
ls $R/gpurun_out/prof_cfg2 $R/gpurun_out/pmc1_cfg2
