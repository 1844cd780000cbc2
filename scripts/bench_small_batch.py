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


def splat_kernel_ms(wl, reps):
    """Median device time of the splat walk kernels of one step (forward walk, backward walk)."""
    from torch.profiler import ProfilerActivity, profile
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            wl.step()
        torch.cuda.synchronize()
    t = {"splat_fwd": [], "splat_bwd": []}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA:
            for k in t:
                if "lp::" + k in e.name:
                    t[k].append(e.device_time_total)
    return tuple(sorted(v)[len(v) // 2] / 1e3 for v in t.values())


def splatter_main(reps):
    """cfg 3's Splatter (32 ch -> 128^3 x 32 voxel grid, 256 samples) on small images.  The march segmentation is decided
    inside the library (lp_splatter.hip, splat_segments); LP_SPLAT_SEGMENTS=1 in the environment switches it off."""
    dev = torch.device("cuda:0")
    lp.config.check_inputs = False
    print(f"LP_SPLAT_SEGMENTS={os.environ.get('LP_SPLAT_SEGMENTS', '(auto)')}")
    print(f"{'rays':>8s} | {'fwd walk ms':>12s} {'bwd walk ms':>12s}")
    for H, W in ((64, 64), (128, 128), (128, 256), (256, 256)):
        wl = bench.SplatterWorkload(0, dev, None, image=(H, W))
        f, b = splat_kernel_ms(wl, reps)
        print(f"{H * W:8d} | {f:12.3f} {b:12.3f}", flush=True)
        del wl
        torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--splatter", action="store_true", help="the Splatter's small batches instead of the Renderer's")
    args = ap.parse_args()
    if args.splatter:
        return splatter_main(args.reps)
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
