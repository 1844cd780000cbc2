#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_coherent.py -m gpu -q -p no:cacheprovider -k "other_kernel_families" > gpurun_out/r2_pytest_gpu_i.txt 2>&1
grep -n "AssertionError:\|Error\b.*:\|passed\|failed" gpurun_out/r2_pytest_gpu_i.txt | tail -10
rm -f gpurun_out/r2_bench_i.txt
for v in "LP_DUMMY=1" "LP_BF3_BWD=1"; do
  echo "== $v" >> gpurun_out/r2_bench_i.txt
  env $v timeout 600 python bench.py --no-cpu-baseline >> gpurun_out/r2_bench_i.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench_i.txt"):
    if l.startswith("=="): print(l.strip())
    elif l.startswith("{"):
        d = json.loads(l)
        print(" cfg2 fwd %.4f bwd %.4f Mrays %.2f | 1080p fwd %.2f bwd %.2f | cfg4 fwd %.2f bwd %.2f" % (d["fwd_ms"], d["bwd_ms"], d["value"],
              d["extras"]["renderer_1080p_s128"]["fwd_ms"], d["extras"]["renderer_1080p_s128"]["bwd_ms"],
              d["extras"]["renderer_cfg4_shard"]["fwd_ms"], d["extras"]["renderer_cfg4_shard"]["bwd_ms"]))
PY
