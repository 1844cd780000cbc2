#!/bin/bash
# round 2, call A: coherent / config-scale parity tests, MFMA-vs-VALU overlap probe, bench line with extras,
# WRITE_SIZE A/B of the Renderer backward (atomics on / off) at cfg 2 and cfg 4.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
timeout 1500 python -m pytest tests/test_gpu_coherent.py -m gpu -q -p no:cacheprovider > gpurun_out/r2_pytest_coherent.txt 2>&1
tail -40 gpurun_out/r2_pytest_coherent.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o /tmp/ovl && timeout 120 /tmp/ovl > gpurun_out/r2_mfma_valu_overlap.txt 2>&1
cat gpurun_out/r2_mfma_valu_overlap.txt
timeout 600 python bench.py > gpurun_out/r2_bench_default.txt 2>&1
tail -1 gpurun_out/r2_bench_default.txt
cd /tmp
for dbg in 0 1; do
  LP_MFMA_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2_pmc_w_cfg2_dbg$dbg -o pmc -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $R/gpurun_out/r2_pmc_w_cfg2_dbg$dbg.txt 2>&1
  LP_MFMA_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/r2_pmc_w_cfg4_dbg$dbg -o pmc -- python $R/bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_pmc_w_cfg4_dbg$dbg.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/r2_pmc_f_cfg4 -o pmc -- python $R/bench.py --workload cfg4 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_pmc_f_cfg4.txt 2>&1
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/r2_pmc_*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "renderer" in r["Kernel_Name"]:
                acc[(r["Kernel_Name"][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(d, k, "avg", sum(v) / len(v), "n", len(v))
PY
# keep the merged output small
find gpurun_out -name "*.csv" -size +4M -delete
