"""Small Renderer batches (fewer rays than wave slots): kernel times with / without the segment-parallel backward
(LpRendererArgs.seg_prefix, config.segment_backward).  Kernel durations come from torch.profiler (device timestamps of
the lp:: kernels), not from events around Python calls: at these sizes the host side of a call is as long as the kernel.

    python scripts/bench_small_batch.py [--reps 20]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
import lightplane_amd as lp

CASES = [  # (H, W, S)
    (64, 64, 64), (64, 64, 128), (64, 64, 256), (128, 128, 128), (128, 256, 128), (192, 256, 128),
]


kernel_ms = bench.kernel_times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lp.config.check_inputs = False
    print(f"{'rays':>8s} {'S':>4s} {'segments':>8s} | {'fwd ms':>8s} {'bwd ms':>8s} | {'fwd ms':>8s} {'bwd ms':>8s} (one sweep per ray) | bwd speed-up")
    for H, W, S in CASES:
        name = f"small_{H}x{W}_s{S}"
        bench.RENDER_CFGS[name] = (H, W, S, 16, 64, name)
        wl = bench.RendererWorkload(name, 0, dev, None, lp._lib.LP_KERNEL_AUTO)
        n_seg = lp.backward_segments(wl.rays, None, wl.dec, num_samples=S, grid_sizes=wl.sizes)
        lp.config.segment_backward = lp.config.segment_forward = True
        f1, b1 = kernel_ms(wl, args.reps)
        lp.config.segment_backward = lp.config.segment_forward = False
        f0, b0 = kernel_ms(wl, args.reps)
        lp.config.segment_backward = lp.config.segment_forward = True
        print(f"{H * W:8d} {S:4d} {n_seg:8d} | {f1:8.3f} {b1:8.3f} | {f0:8.3f} {b0:8.3f} | {b0 / b1:5.2f}x  fwd {f0 / f1:5.2f}x", flush=True)


if __name__ == "__main__":
    main()
