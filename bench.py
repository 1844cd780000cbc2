#!/usr/bin/env python
"""bench.py -- Mrays/s fwd+bwd of the Lightplane Renderer hot path on MI355X.

Workload (BASELINE.json configs[1], "cfg 2"): 256x256 rays per GPU (pinhole camera looking at the
[-1,1]^3 cube), triplane 3 x [1, 64(1), 64(1), 64(1), 16], 128 samples/ray, trunk/opacity/colour
MLPs with 2 layers x 32 hidden, 3 colour channels, 32-wide ray encoding; synthetic N(0,1) grid,
random-init decoder.  One step = forward + backward (gradients w.r.t. grid, MLP parameters and
ray encoding), inputs resident in HBM.  With N>1 GPUs every rank renders its own 256x256 camera
(weak scaling), the grid / parameters are replicated and their gradients are summed with an RCCL
all-reduce inside the timed step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (renderer backward):
algorithmic bytes per launch (S*K*C*4 re-gather + S*K*C*4 atomic payload + ray I/O)
divided by its mean launch time measured with HIP events on the launch stream.  `cpu_baseline` is the
CPU oracle (oracle/lightplane_oracle.py, the PyTorch restatement of the reference's naive renderer)
timed on a bounded ray subset of the same workload on the host cores of rank 0.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import lightplane_amd as lp  # noqa: E402
from lightplane_amd import _lib, parallel  # noqa: E402
from lightplane_amd import renderer as R  # noqa: E402

H = W = 256
S = 128
C = 16
GRID = 64
HIDDEN = 32
COLOR = 3


def make_workload(rank: int, dev):
    from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids

    gen = torch.Generator().manual_seed(0)
    sizes = grid_sizes_for((1, GRID, GRID, GRID, C), True)
    grids = random_grids(gen, sizes)
    dec = random_decoder(gen, 2, 2, 2, C, HIDDEN, COLOR, std=0.15)
    gen_r = torch.Generator().manual_seed(100 + rank)
    rays = pinhole_rays(H, W, enc_dim=HIDDEN, gen=gen_r, azimuth_deg=45.0 * rank, elevation_deg=0.0 if rank == 0 else 30.0)
    up = (torch.randn(H * W, generator=gen_r), torch.randn(H * W, generator=gen_r), torch.randn(H * W, COLOR, generator=gen_r))
    return rays, grids, dec, sizes, up


def algorithmic_bytes(n_rays):
    k = 12  # triplane: 3 planes x 4 corners
    per_sample = k * C * 4
    fwd = n_rays * (S * per_sample + 184)
    bwd = n_rays * (S * per_sample * 2 + 316)
    return fwd, bwd


def pmc_traffic():
    """HBM bytes per backward launch from the committed rocprofv3 PMC passes (profiles/r*_pmc_summary.json:
    (FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc runs of this same command), or None."""
    import glob

    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary.json")))
    if not files:
        return None
    try:
        d = json.load(open(files[-1]))
        for k, v in d.items():
            if "renderer_bwd_mfma" in k and "hbm_bytes_per_launch" in v:
                return int(v["hbm_bytes_per_launch"])
    except Exception:
        return None
    return None


def cpu_baseline(rays, grids, dec, n_sub=1024):
    """Oracle (kind 'port': our PyTorch restatement of the reference's naive path) fwd+bwd on CPU."""
    import copy

    from oracle import lightplane_oracle as O

    idx = torch.arange(0, rays.n_rays, rays.n_rays // n_sub)[:n_sub]
    r = rays[idx]
    torch.set_num_threads(min(16, os.cpu_count() or 1))  # more threads only thrash on this tiny problem

    def one():
        rr = copy.copy(r)
        rr.encoding = r.encoding.clone().requires_grad_(True)
        d = copy.copy(dec)
        d.mlp_params = dec.mlp_params.clone().requires_grad_(True)
        gs = [g.clone().requires_grad_(True) for g in grids]
        out = O.lightplane_renderer_naive(rr, gs, d, num_samples=S, gain=1.0)
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()

    one()
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < 10.0 and reps < 50):
        one()
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    return {"value": round(n_sub / dt / 1e6, 6), "unit": "Mrays/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_sub} rays (every {rays.n_rays // n_sub}-th) of the same workload, fwd+bwd, {reps} reps, "
                      f"oracle/lightplane_oracle.py on CPU"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--kernel", type=int, default=_lib.LP_KERNEL_AUTO)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the "
                                                       "multi-rank code path with several ranks on one GPU)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    dev = torch.device(f"cuda:{local_rank % torch.cuda.device_count()}")
    torch.cuda.set_device(dev)
    pg = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        pg = dist.group.WORLD
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    lp.config.check_inputs = False  # the grid_idx range check is a host sync, not part of the op
    rays_c, grids_c, dec_c, sizes, up_c = make_workload(rank, dev)
    rays = rays_c.to(dev)
    flat, _ = lp.flatten_grid([g.to(dev) for g in grids_c])
    flat.requires_grad_(True)
    params = dec_c.mlp_params.to(dev).requires_grad_(True)
    rays.encoding.requires_grad_(True)
    dec = lp.DecoderParams(params, dec_c.n_hidden_trunk, dec_c.n_hidden_opacity, dec_c.n_hidden_color, COLOR)
    up = [u.to(dev) for u in up_c]
    n_rays = rays.n_rays

    def step():
        flat.grad = params.grad = rays.encoding.grad = None
        g, p = parallel.replicate_with_grad_allreduce([flat, params], pg)
        d = lp.DecoderParams(p, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, COLOR)
        out = lp.lightplane_renderer(rays, g, d, num_samples=S, gain=1.0, grid_sizes=sizes, kernel=args.kernel)
        loss = (out[0] * up[0]).sum() + (out[1] * up[1]).sum() + (out[2] * up[2]).sum()
        loss.backward()
        return out

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    torch.cuda.reset_peak_memory_stats(dev)
    mem0 = torch.cuda.memory_allocated(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    dt = time.perf_counter() - t0
    peak_mb = (torch.cuda.max_memory_allocated(dev) - mem0) / 2**20
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = n_rays * world / (ms_per_step * 1e-3) / 1e6

    # --- per-kernel timing with HIP events on the launch stream (forward / backward separately) ---
    with torch.no_grad():
        out = lp.lightplane_renderer(rays, flat.detach(), dec, num_samples=S, gain=1.0, grid_sizes=sizes, kernel=args.kernel)
    cfg_args = dict(num_samples=S, gain=1.0, grid_sizes=sizes, kernel=args.kernel)
    # No host sync inside the loop: the host runs ahead of the GPU, so the events bracket the kernels back to back
    # on the stream instead of the host's launch latency after an idle GPU.
    reps = max(5, min(args.steps, 20))
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps + 2)]
    for ev in evs:
        flat.grad = params.grad = rays.encoding.grad = None
        ev[0].record()
        o = lp.lightplane_renderer(rays, flat, dec, **cfg_args)
        ev[1].record()
        loss = (o[0] * up[0]).sum() + (o[1] * up[1]).sum() + (o[2] * up[2]).sum()
        ev[2].record()
        loss.backward()
        ev[3].record()
    torch.cuda.synchronize(dev)
    fwd_ms = sum(ev[0].elapsed_time(ev[1]) for ev in evs[2:]) / reps  # the first two iterations fill the queue
    bwd_ms = sum(ev[2].elapsed_time(ev[3]) for ev in evs[2:]) / reps
    fwd_b, bwd_b = algorithmic_bytes(n_rays)
    achieved = bwd_b / (bwd_ms * 1e-3) / 1e9
    mlp_mac = C * HIDDEN + HIDDEN * HIDDEN + HIDDEN * HIDDEN + HIDDEN + HIDDEN * HIDDEN + HIDDEN * COLOR
    flops_fwdbwd = 2 * mlp_mac * S * 4 * n_rays

    if rank == 0:
        res = {
            "metric": "Mrays/sec fwd+bwd, 64^3x16ch triplane @128 samples; peak bwd mem (MB)",
            "value": round(value, 4), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "cfg2: Renderer fwd+bwd, 256x256 rays/GPU, triplane 64^2x16ch, 128 samples, "
                                   "2-layer/32-hidden trunk/opacity/color MLP, 3 colour ch, 32-ch ray encoding",
                       "rays_per_gpu": n_rays, "parallelism": f"ray-shard dp{world}, grid replicated, grad all-reduce"},
            "peak_bwd_mem_mb": round(peak_mb, 2),
            "fwd_ms": round(fwd_ms, 4), "bwd_ms": round(bwd_ms, 4),
            "roofline": {"bound": "hbm", "kernel": "renderer backward", "achieved": round(achieved, 2), "peak": 8000.0,
                         "unit": "GB/s", "frac": round(achieved / 8000.0, 5), "traffic": pmc_traffic(),
                         "algorithmic_bytes_per_launch": bwd_b,
                         "note": "effective bandwidth: the 786 KB grid is L2 resident, compulsory HBM bytes are ~0.5 KB/ray"},
            "mlp_fp32_frac_of_peak": round(flops_fwdbwd / ((fwd_ms + bwd_ms) * 1e-3) / 157.3e12, 5),
        }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N=1 only
            res["cpu_baseline"] = cpu_baseline(rays_c, grids_c, dec_c)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
