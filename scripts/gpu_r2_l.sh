#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1800 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "renderer or cfg or module or fit or graph or early or sweep" > gpurun_out/r2_pytest_gpu_l.txt 2>&1
grep -n "AssertionError:\|Error\b.*:\|passed\|failed" gpurun_out/r2_pytest_gpu_l.txt | tail -12
rm -f gpurun_out/r2_bench_l.txt
for v in "LP_BF3_C32=1" "LP_BF3_C32=0"; do
  echo "== $v" >> gpurun_out/r2_bench_l.txt
  env $v timeout 600 python bench.py --no-cpu-baseline >> gpurun_out/r2_bench_l.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r2_bench_l.txt"):
    if l.startswith("=="): print(l.strip())
    elif l.startswith("{"):
        d = json.loads(l)
        print(" cfg2 fwd %.4f bwd %.4f Mrays %.2f | 1080p fwd %.2f bwd %.2f | cfg4 fwd %.2f bwd %.2f" % (d["fwd_ms"], d["bwd_ms"], d["value"],
              d["extras"]["renderer_1080p_s128"]["fwd_ms"], d["extras"]["renderer_1080p_s128"]["bwd_ms"],
              d["extras"]["renderer_cfg4_shard"]["fwd_ms"], d["extras"]["renderer_cfg4_shard"]["bwd_ms"]))
    elif "Error" in l or "error" in l: print(l.strip()[:300])
PY
