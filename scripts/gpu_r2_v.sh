#!/bin/bash
# bench.py's N > 1 code path with two ranks on ONE GPU (gloo): the headline + the 1080p legs + the watchdog
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo > gpurun_out/v_bench2.log 2>&1; echo "rc=$?" >> gpurun_out/v_bench2.log
tail -3 gpurun_out/v_bench2.log | cut -c1-1500
LP_BENCH_EXTRAS_TIMEOUT=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --steps 20 --warmup 3 --backend gloo > gpurun_out/v_bench2_wd.log 2>&1; echo "rc=$?" >> gpurun_out/v_bench2_wd.log
tail -3 gpurun_out/v_bench2_wd.log | cut -c1-600
