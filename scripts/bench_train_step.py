"""The reference example's TRAINING STEP (examples/fit_single_scene.py + examples/config/synthetic_overfit.json of the reference): a
batch of `--n_rays` (default 4096) RANDOM rays per iteration through a triplane 128^2 x 32 ch, S = 128, decoder 2/2/2 x hidden
(64 in the example's config, 32 = the tuned family), optionally a scaffold.  Reports, per (hidden, n_rays): the wall time of one
forward + backward as a training loop issues them back to back (events around 50 steps: host-side cost shows where the GPU would idle),
and the device time of the lp:: kernels alone (torch.profiler).

    python scripts/bench_train_step.py [--reps 50]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import lightplane_amd as lp
from tests.synth import grid_sizes_for, random_decoder, random_grids


def random_rays(n, enc_dim, dev, gen):
    o = torch.randn(n, 3, generator=gen) * 0.1 + torch.tensor([0.0, 0.0, 2.7])
    tgt = torch.rand(n, 3, generator=gen) * 2 - 1
    d = torch.nn.functional.normalize(tgt - o, dim=-1)
    return lp.Rays(directions=d.to(dev), origins=o.to(dev), grid_idx=torch.zeros(n, dtype=torch.int32, device=dev),
                   near=torch.full((n,), 1.0, device=dev), far=torch.full((n,), 4.4, device=dev),
                   encoding=torch.randn(n, enc_dim, generator=gen).to(dev).requires_grad_(True))


def lp_kernel_ms(step, reps):
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
    tot, names = 0.0, {}
    for e in prof.events():
        if e.device_type == torch.autograd.DeviceType.CUDA and "lp::" in e.name:
            tot += e.device_time_total
            k = e.name.split("<")[0].replace("void ", "")
            names[k] = names.get(k, 0.0) + e.device_time_total
    return tot / reps / 1e3, {k: round(v / reps / 1e3, 4) for k, v in names.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--sizes", default="1024,4096,8192,16384")
    ap.add_argument("--hidden", default="64,32")
    ap.add_argument("--scaffold", type=int, default=1)
    ap.add_argument("--march", default="samples", help="march_order of the call: random rays are what 'auto' (with config.check_inputs) sends "
                                                       "to the transposed march; 'rays' = rays per wavefront")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lp.config.check_inputs = False
    C, G, S = 32, 128, 128
    print(f"march_order={args.march} LP_MFMA_DEBUG={os.environ.get('LP_MFMA_DEBUG', '')}")
    print(f"{'hidden':>6s} {'rays':>6s} {'family':>6s} | {'step wall ms':>12s} {'lp kernels ms':>13s} | {'Mrays/s (wall)':>14s}  kernels")
    for hidden in [int(v) for v in args.hidden.split(",")]:
        gen = torch.Generator().manual_seed(0)
        dec0 = random_decoder(gen, 2, 2, 2, C, hidden, 3, std=0.15)
        params = dec0.mlp_params.to(dev).requires_grad_(True)
        dec = lp.DecoderParams(params, dec0.n_hidden_trunk, dec0.n_hidden_opacity, dec0.n_hidden_color, 3)
        sizes = grid_sizes_for((1, G, G, G, C), True)
        grids = [g.to(dev).requires_grad_(True) for g in random_grids(gen, sizes)]
        scaffold = (torch.rand(1, G, G, G, generator=gen) < 0.35).float().to(dev) if args.scaffold else None
        for n in [int(v) for v in args.sizes.split(",")]:
            rays = random_rays(n, hidden, dev, gen)
            target = torch.rand(n, 3, generator=gen).to(dev)
            fam = lp.kernel_family(rays, grids, dec)

            def step():
                params.grad = rays.encoding.grad = None
                for g in grids:
                    g.grad = None
                _, _, feat = lp.lightplane_renderer(rays, grids, dec, num_samples=S, gain=1.0, scaffold=scaffold,
                                                    mask_out_of_bounds_samples=True, march_order=args.march)
                ((feat - target) ** 2).mean().backward()

            for _ in range(10):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                step()
            e1.record()
            torch.cuda.synchronize()
            wall = e0.elapsed_time(e1) / args.reps
            kms, names = lp_kernel_ms(step, 10)
            print(f"{hidden:6d} {n:6d} {fam:6d} | {wall:12.3f} {kms:13.3f} | {n / wall / 1e3:14.2f}  {names}", flush=True)


if __name__ == "__main__":
    main()
