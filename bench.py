#!/usr/bin/env python
"""bench.py -- Mrays/s fwd+bwd of the Lightplane Renderer / Splatter hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg4|1080p_s128|cfg3]

Workloads (BASELINE.json `configs`); one step = forward + backward, inputs resident in HBM:
  cfg2        (default, the headline metric) Renderer, 256x256 rays per GPU (pinhole camera looking at the [-1,1]^3
              cube), triplane 3 x 64^2 x 16 ch, 128 samples/ray, trunk/opacity/colour MLPs 2 layers x 32 hidden,
              3 colour channels, 32-wide ray encoding.
  cfg4        Renderer, ONE 1920x1080 camera per GPU (cameras on a ring, elevation 30 deg, azimuth 45 deg x rank),
              triplane 3 x 128^2 x 32 ch, 256 samples -- the 8-GPU ray-shard configuration of BASELINE.json.
  1080p_s128  Renderer, one 1920x1080 camera per GPU on the cfg-2 scene (triplane 64^2 x 16 ch, 128 samples):
              the batch north_star asks the Mrays/s report for.
  cfg3        Splatter, 256x256 rays x 32 ch per GPU -> voxel grid 128^3 x 32 ch, 256 samples.
With N > 1 GPUs (one process per GPU, RCCL) every rank works on its own camera (weak scaling), the grid / parameters
are replicated and the Renderer's grid + MLP gradients (the Splatter's un-normalised output + weight grid) are summed
with an all-reduce INSIDE the timed step.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel of the workload (Renderer: the backward
kernel; Splatter: the forward walk): SURVEY.md 8(d)'s algorithmic bytes per launch divided by that kernel's mean
launch time, measured with HIP events on the launch stream (torch's current stream is the stream the C ABI launches on).
At N = 1 with the default workload the line also carries `extras`: the other configurations measured the same way
(short runs), so that every number quoted in DESIGN.md / README.md can be recomputed from the driver's BENCH file (at
N > 1 `extras` holds north_star's reporting batches instead: 1920x1080 rays per GPU at S = 128 and the cfg 4 shard,
through the same sharded step with its RCCL all-reduce, bounded by a watchdog), and
`cpu_baseline`: the CPU oracle (oracle/lightplane_oracle.py, the PyTorch restatement of the reference's naive path)
timed on the host cores of rank 0, on BASELINE configs[0] (cfg 1) and on a ray subsample of the benchmarked workload.
"""
import argparse
import glob
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import lightplane_amd as lp  # noqa: E402
from lightplane_amd import _lib, parallel  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec
FP32_PEAK = 157.3e12    # fp32 vector = fp32 MFMA peak


# ----------------------------------------------------------------------------------------------------------------
# workloads
# ----------------------------------------------------------------------------------------------------------------

RENDER_CFGS = {
    #              H     W     S    C   grid  description
    "cfg2": (256, 256, 128, 16, 64, "cfg2: Renderer fwd+bwd, 256x256 rays/GPU, triplane 64^2x16ch, 128 samples, "
                                    "2-layer/32-hidden trunk/opacity/color MLP, 3 colour ch, 32-ch ray encoding"),
    "cfg4": (1080, 1920, 256, 32, 128, "cfg4 shard: Renderer fwd+bwd, 1920x1080 rays/GPU, triplane 128^2x32ch, 256 samples, "
                                       "2-layer/32-hidden MLPs, 3 colour ch, grid replicated + grad all-reduce"),
    "1080p_s128": (1080, 1920, 128, 16, 64, "1080p_s128: Renderer fwd+bwd, 1920x1080 rays/GPU, triplane 64^2x16ch, "
                                            "128 samples, 2-layer/32-hidden MLPs, 3 colour ch"),
    "small": (64, 64, 128, 16, 64, "small batch: Renderer fwd+bwd, 64x64 rays, triplane 64^2x16ch, 128 samples "
                                   "(segment-parallel march, DESIGN.md 4.9)"),
}
HIDDEN, COLOR = 32, 3


def camera_pose(name, rank):
    if name == "cfg2":  # rank 0: the axis-aligned view of round 1's headline; other ranks from the ring
        return 45.0 * rank, (0.0 if rank == 0 else 30.0)
    return 45.0 * rank, 30.0  # 8 cameras on a ring, elevation 30 deg


class RendererWorkload:
    def __init__(self, name, rank, dev, pg, kernel):
        from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids

        self.name, self.pg, self.kernel = name, pg, kernel
        H, W, S, C, G, self.desc = RENDER_CFGS[name]
        self.S, self.C = S, C
        gen = torch.Generator().manual_seed(0)
        self.sizes = grid_sizes_for((1, G, G, G, C), True)
        self.grids_c = random_grids(gen, self.sizes)
        self.dec_c = random_decoder(gen, 2, 2, 2, C, HIDDEN, COLOR, std=0.15)
        gen_r = torch.Generator().manual_seed(100 + rank)
        az, el = camera_pose(name, rank)
        self.rays_c = pinhole_rays(H, W, enc_dim=HIDDEN, gen=gen_r, azimuth_deg=az, elevation_deg=el)
        n = H * W
        up = (torch.randn(n, generator=gen_r), torch.randn(n, generator=gen_r), torch.randn(n, COLOR, generator=gen_r))
        self.n_rays = n
        self.rays = self.rays_c.to(dev)
        self.flat, _ = lp.flatten_grid([g.to(dev) for g in self.grids_c])
        self.flat.requires_grad_(True)
        self.params = self.dec_c.mlp_params.to(dev).requires_grad_(True)
        self.rays.encoding.requires_grad_(True)
        self.dec = lp.DecoderParams(self.params, self.dec_c.n_hidden_trunk, self.dec_c.n_hidden_opacity,
                                    self.dec_c.n_hidden_color, COLOR)
        self.up = [u.to(dev) for u in up]

    # bytes/ray = S*K*C*4 gathered (fwd) ; the same re-gathered + the same as atomic payload (bwd) ; + ray I/O
    def algorithmic_bytes(self):
        per_sample = 12 * self.C * 4  # triplane: 3 planes x 4 corners
        return self.n_rays * (self.S * per_sample + 184), self.n_rays * (self.S * per_sample * 2 + 316)

    def mlp_flops_fwdbwd(self):
        mac = self.C * HIDDEN + HIDDEN * HIDDEN + HIDDEN * HIDDEN + HIDDEN + HIDDEN * HIDDEN + HIDDEN * COLOR
        return 2 * mac * self.S * 4 * self.n_rays  # forward + (recompute + dX + dW)

    def zero_grads(self):
        self.flat.grad = self.params.grad = self.rays.encoding.grad = None

    def forward(self, replicated=True):
        g, p = (self.flat, self.params)
        if replicated:
            g, p = parallel.replicate_with_grad_allreduce([self.flat, self.params], self.pg)
        d = lp.DecoderParams(p, self.dec.n_hidden_trunk, self.dec.n_hidden_opacity, self.dec.n_hidden_color, COLOR)
        return lp.lightplane_renderer(self.rays, g, d, num_samples=self.S, gain=1.0, grid_sizes=self.sizes, kernel=self.kernel)

    def loss(self, out):
        return (out[0] * self.up[0]).sum() + (out[1] * self.up[1]).sum() + (out[2] * self.up[2]).sum()

    def step(self):
        self.zero_grads()
        self.loss(self.forward()).backward()

    roofline_kernel = "renderer backward"

    def roofline(self, fwd_ms, bwd_ms):
        fwd_b, bwd_b = self.algorithmic_bytes()
        ach = bwd_b / (bwd_ms * 1e-3) / 1e9
        both = (fwd_b + bwd_b) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": self.roofline_kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": bwd_b,
                "frac_fwd_plus_bwd": round(both / HBM_PEAK_GBS, 5)}


class SplatterWorkload:
    """cfg 3: 256x256 rays x 32 ch -> voxel 128^3 x 32 ch, S = 256 (atomic scatter-add path)."""

    name = "cfg3"
    desc = "cfg3: Splatter fwd+bwd, 256x256 rays/GPU x 32ch encoding -> 128^3x32ch voxel grid, 256 samples"
    roofline_kernel = "splatter forward walk"

    def __init__(self, rank, dev, pg, image=(256, 256)):
        from tests.synth import pinhole_rays

        self.pg, self.S, self.C, self.G = pg, 256, 32, 128
        gen = torch.Generator().manual_seed(100 + rank)
        az, el = camera_pose("cfg2", rank)
        self.rays_c = pinhole_rays(image[0], image[1], gen=gen, azimuth_deg=az, elevation_deg=el)
        self.rays_c.encoding = torch.rand(self.rays_c.n_rays, self.C, generator=gen)
        self.rays = self.rays_c.to(dev)
        self.rays.encoding.requires_grad_(True)
        self.n_rays = self.rays.n_rays
        self.sizes = [[1, self.G, self.G, self.G, self.C]]
        self.up = torch.randn(self.G ** 3, self.C, generator=gen).to(dev)

    # S*(K*C*4 + K*4) atomics forward, S*K*C*4 gathered backward (K = 8 corners)
    def algorithmic_bytes(self):
        return self.n_rays * self.S * (8 * self.C * 4 + 8 * 4), self.n_rays * self.S * 8 * self.C * 4

    def zero_grads(self):
        self.rays.encoding.grad = None

    def forward(self, replicated=True):
        return lp.lightplane_splatter(self.rays, self.sizes, num_samples=self.S, return_list=False,
                                      process_group=self.pg if replicated else None)

    def loss(self, out):
        return (out * self.up).sum()

    def step(self):
        self.zero_grads()
        self.loss(self.forward()).backward()

    def roofline(self, fwd_ms, bwd_ms):
        fwd_b, bwd_b = self.algorithmic_bytes()
        ach = fwd_b / (fwd_ms * 1e-3) / 1e9
        both = (fwd_b + bwd_b) / ((fwd_ms + bwd_ms) * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": self.roofline_kernel, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_launch": fwd_b,
                "frac_fwd_plus_bwd": round(both / HBM_PEAK_GBS, 5),
                "note": "fwd = splat walk + normalise + torch zero-fill of the 268 MB grid; SURVEY 8(d)'s 532 936 B/ray"}


def make_workload(name, rank, dev, pg, kernel):
    if name == "cfg3":
        return SplatterWorkload(rank, dev, pg)
    return RendererWorkload(name, rank, dev, pg, kernel)


def event_times(wl, reps):
    """Mean forward / backward time of the op itself (no collective): HIP events on the launch stream, the host runs
    ahead of the GPU so the events bracket the kernels back to back (the first two iterations fill the queue)."""
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(reps + 2)]
    for ev in evs:
        wl.zero_grads()
        ev[0].record()
        out = wl.forward(replicated=False)
        ev[1].record()
        loss = wl.loss(out)
        ev[2].record()
        loss.backward()
        ev[3].record()
    torch.cuda.synchronize()
    fwd_ms = sum(ev[0].elapsed_time(ev[1]) for ev in evs[2:]) / reps
    bwd_ms = sum(ev[2].elapsed_time(ev[3]) for ev in evs[2:]) / reps
    return fwd_ms, bwd_ms


def pmc_traffic(workload, kernel_substr):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (profiles/r*_pmc_summary.json:
    (FETCH_SIZE + WRITE_SIZE) * 1024 from separate --pmc runs of `bench.py --workload <w>`).  Not measured in this run."""
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_pmc_summary.json")))
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        for k, v in d.items():
            if k.startswith(workload + ":") and kernel_substr in k and "hbm_bytes_per_launch" in v:
                return int(v["hbm_bytes_per_launch"]), os.path.relpath(f, REPO)
    return None, None


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle = 'port' of the reference's naive path; test infrastructure, never the product)
# ----------------------------------------------------------------------------------------------------------------


def _time_cpu(one, budget_s, max_reps=50):
    one()
    t0 = time.perf_counter()
    reps = 0
    while reps < 3 or (time.perf_counter() - t0 < budget_s and reps < max_reps):
        one()
        reps += 1
    return (time.perf_counter() - t0) / reps, reps


def cpu_baseline(wl):
    import copy

    from oracle import lightplane_oracle as O
    from tests.synth import random_decoder, random_grids, random_rays

    torch.set_num_threads(min(16, os.cpu_count() or 1))  # more threads only thrash on problems this small
    cores = torch.get_num_threads()
    res = {"unit": "Mrays/s", "cores": cores, "kind": "port"}

    # BASELINE configs[0] ("cfg 1"): 1k random rays, 32^3 x 16 voxel grid, 64 samples, 2/2/2 x 32 decoder
    gen = torch.Generator().manual_seed(0)
    grids1 = random_grids(gen, [[1, 32, 32, 32, 16]])
    dec1 = random_decoder(gen, 2, 2, 2, 16, 32, 3, std=0.15)
    rays1 = random_rays(gen, 1000, 1, 32)
    rays1.near = torch.full((1000,), 0.1)
    rays1.far = torch.full((1000,), 3.0)

    def one_cfg1():
        rr = copy.copy(rays1)
        rr.encoding = rays1.encoding.clone().requires_grad_(True)
        d = copy.copy(dec1)
        d.mlp_params = dec1.mlp_params.clone().requires_grad_(True)
        gs = [g.clone().requires_grad_(True) for g in grids1]
        out = O.lightplane_renderer_naive(rr, gs, d, num_samples=64, gain=1.0)
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()

    dt, reps = _time_cpu(one_cfg1, 6.0)
    res["cfg1"] = {"value": round(1000 / dt / 1e6, 6), "sample": f"BASELINE configs[0] exactly: 1000 random rays, 32^3x16 voxel, "
                   f"64 samples, 2/2/2x32 MLPs, fwd+bwd, {reps} reps"}

    n_sub = 1024
    if isinstance(wl, RendererWorkload):
        idx = torch.arange(0, wl.n_rays, wl.n_rays // n_sub)[:n_sub]
        r = wl.rays_c[idx]

        def one():
            rr = copy.copy(r)
            rr.encoding = r.encoding.clone().requires_grad_(True)
            d = copy.copy(wl.dec_c)
            d.mlp_params = wl.dec_c.mlp_params.clone().requires_grad_(True)
            gs = [g.clone().requires_grad_(True) for g in wl.grids_c]
            out = O.lightplane_renderer_naive(rr, gs, d, num_samples=wl.S, gain=1.0)
            (out[0].sum() + out[1].sum() + out[2].sum()).backward()
    else:
        idx = torch.arange(0, wl.n_rays, wl.n_rays // n_sub)[:n_sub]
        r = wl.rays_c[idx]

        def one():
            rr = copy.copy(r)
            rr.encoding = r.encoding.clone().requires_grad_(True)
            out = O.lightplane_splatter_naive(rr, wl.sizes, num_samples=wl.S)
            out[0].sum().backward()

    dt, reps = _time_cpu(one, 10.0)
    res["value"] = round(n_sub / dt / 1e6, 6)
    res["sample"] = (f"{wl.name}-subsample: {n_sub} rays (every {wl.n_rays // n_sub}-th) of the benchmarked workload, fwd+bwd, "
                     f"{reps} reps, oracle/lightplane_oracle.py on CPU")
    return res


# ----------------------------------------------------------------------------------------------------------------


def measure_extra(name, dev, kernel, reps):
    """Short single-GPU measurement of another configuration (events only) for the `extras` block."""
    wl = make_workload(name, 0, dev, None, kernel)
    for _ in range(2 if name in ("cfg4", "1080p_s128") else 10):
        wl.step()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats(dev)
    mem0 = torch.cuda.memory_allocated(dev)
    fwd_ms, bwd_ms = event_times(wl, reps)
    peak_mb = (torch.cuda.max_memory_allocated(dev) - mem0) / 2**20
    out = {"workload": wl.desc, "rays": wl.n_rays, "fwd_ms": round(fwd_ms, 4), "bwd_ms": round(bwd_ms, 4),
           "Mrays_per_s_fwd_bwd": round(wl.n_rays / (fwd_ms + bwd_ms) / 1e3, 4), "reps": reps,
           "peak_bwd_mem_mb": round(peak_mb, 2), "roofline": wl.roofline(fwd_ms, bwd_ms)}
    if isinstance(wl, RendererWorkload):
        out["mlp_fp32_frac_of_peak"] = round(wl.mlp_flops_fwdbwd() / ((fwd_ms + bwd_ms) * 1e-3) / FP32_PEAK, 5)
    del wl
    torch.cuda.empty_cache()
    return out


def kernel_times(wl, reps):
    """Median device time of the lp::renderer_fwd* / lp::renderer_bwd* kernels per step, from torch.profiler's device
    timestamps: for batches whose kernels are as short as the host side of a call, events around Python calls would
    measure the host."""
    from torch.profiler import ProfilerActivity, profile
    for _ in range(3):
        wl.step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(reps):
            wl.step()
        torch.cuda.synchronize()
    per_kernel = {}  # kernel name -> device times of its launches (us)
    for e in prof.events():
        if e.device_type != torch.autograd.DeviceType.CUDA or not ("lp::renderer_fwd" in e.name or "lp::renderer_bwd" in e.name):
            continue
        per_kernel.setdefault(e.name.split("(")[0].replace("void ", ""), []).append(e.device_time_total)
    # median per kernel (a launch that shared the GPU with an allocator sync or a clock ramp does not move it), times the
    # launches per step
    med = {k: sorted(v)[len(v) // 2] * (len(v) / reps) for k, v in per_kernel.items()}
    kernel_times.last = {k: {"launches": len(v), "median_ms": round(sorted(v)[len(v) // 2] / 1e3, 4),
                             "max_ms": round(max(v) / 1e3, 4)} for k, v in per_kernel.items()}
    fwd = sum(t for k, t in med.items() if "renderer_fwd" in k)
    bwd = sum(t for k, t in med.items() if "renderer_bwd" in k)
    return fwd / 1e3, bwd / 1e3


def measure_small_batch(dev, kernel, reps):
    """A NeRF-style training batch (4 096 rays x 128 samples, cfg 2's grid and decoder): kernel times with the
    segment-parallel forward / backward (default; forward = segment march + combine pass) and with one sweep per ray."""
    wl = RendererWorkload("small", 0, dev, None, kernel)
    n_seg = lp.backward_segments(wl.rays, None, wl.dec, num_samples=wl.S, grid_sizes=wl.sizes)
    saved = lp.config.segment_backward, lp.config.segment_forward
    try:
        lp.config.segment_backward = lp.config.segment_forward = True
        f1, b1 = kernel_times(wl, reps)
        kernels = kernel_times.last
        lp.config.segment_backward = lp.config.segment_forward = False
        f0, b0 = kernel_times(wl, reps)
    finally:
        lp.config.segment_backward, lp.config.segment_forward = saved
    return {"workload": wl.desc, "rays": wl.n_rays, "backward_segments": n_seg, "fwd_ms": round(f1, 4), "bwd_ms": round(b1, 4),
            "bwd_ms_one_sweep_per_ray": round(b0, 4), "fwd_ms_one_sweep_per_ray": round(f0, 4), "reps": reps, "kernels": kernels,
            "timing": "torch.profiler device time of the lp:: kernels"}


def measure_sharded(name, rank, world, dev, pg, kernel, steps):
    """N > 1: one of the 1080p-per-rank configurations through the same ray-sharded step (grid / decoder gradients
    all-reduced over RCCL inside the step), timed like the headline: barrier + synchronize on both sides, max over ranks.
    Every rank runs this."""
    wl = make_workload(name, rank, dev, pg, kernel)
    wl.step()
    torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    torch.distributed.barrier()
    torch.cuda.synchronize(dev)
    t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    ms = float(t.item()) / steps * 1e3
    out = {"workload": wl.desc, "rays_per_gpu": wl.n_rays, "n_gpus": world, "steps": steps, "ms_per_step": round(ms, 3),
           "Mrays_per_s_fwd_bwd": round(wl.n_rays * world / ms / 1e3, 4)}
    del wl
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: >= 0.5 s of work: 200 for cfg2 / cfg3, "
                                                            "10 for the 1080p workloads)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "cfg4", "1080p_s128", "cfg3", "small"])
    ap.add_argument("--kernel", type=int, default=_lib.LP_KERNEL_AUTO)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the short runs of the other configurations (N = 1 only)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to exercise the "
                                                       "multi-rank code path with several ranks on one GPU)")
    args = ap.parse_args()
    big = args.workload in ("cfg4", "1080p_s128")
    steps = args.steps if args.steps is not None else (10 if big else 200)
    warmup = args.warmup if args.warmup is not None else (2 if big else 10)

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
    wl = make_workload(args.workload, rank, dev, pg, args.kernel)

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(warmup):
        wl.step()
    sync()
    torch.cuda.reset_peak_memory_stats(dev)
    mem0 = torch.cuda.memory_allocated(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        wl.step()
    sync()
    dt = time.perf_counter() - t0
    peak_mb = (torch.cuda.max_memory_allocated(dev) - mem0) / 2**20
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / steps * 1e3
    value = wl.n_rays * world / (ms_per_step * 1e-3) / 1e6

    fwd_ms, bwd_ms = event_times(wl, max(5, min(steps, 20)))

    # N > 1, default workload: north_star's reporting batches (1920x1080 rays per GPU) through the same sharded step, so
    # that the driver's scaling runs carry them.  A watchdog bounds the leg: if it has not finished in time, rank 0
    # prints the headline line without it and every rank exits (a stuck collective cannot take the run with it).
    sharded = None
    watchdog_fired = []
    if world > 1 and args.workload == "cfg2" and not args.no_extras:
        import threading
        done = threading.Event()
        line_ready = {}

        def bail():
            if done.is_set():
                return
            watchdog_fired.append(True)
            if rank == 0 and "res" in line_ready:
                r = dict(line_ready["res"])
                r["extras_error"] = "1080p legs did not finish within the watchdog limit"
                print(json.dumps(r), flush=True)
            os._exit(0)

        timer = threading.Timer(float(os.environ.get("LP_BENCH_EXTRAS_TIMEOUT", "240")), bail)
        timer.daemon = True
    else:
        timer = None

    def headline():
        roof = wl.roofline(fwd_ms, bwd_ms)
        traffic, src = pmc_traffic(args.workload, "renderer_bwd" if isinstance(wl, RendererWorkload) else "splat_fwd_walk")
        roof["traffic"] = traffic
        roof["traffic_source"] = (f"{src}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, NOT measured in "
                                  f"this run; L2 -> fabric requests, i.e. one 64 B write request per atomic segment") if src else None
        if isinstance(wl, RendererWorkload) and args.workload in ("cfg2", "1080p_s128"):
            roof["note"] = "effective bandwidth: the 786 KB grid is L2 resident, compulsory HBM bytes are ~0.5 KB/ray"
        res = {
            "metric": ("Mrays/sec fwd+bwd, 64^3x16ch triplane @128 samples; peak bwd mem (MB)" if args.workload == "cfg2"
                       else f"Mrays/sec fwd+bwd ({args.workload}); peak bwd mem (MB)"),
            "value": round(value, 4), "unit": "Mrays/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": wl.desc, "rays_per_gpu": wl.n_rays,
                       "parallelism": f"ray-shard dp{world}, grid replicated, "
                                      + ("un-normalised splat + weights all-reduced" if args.workload == "cfg3" else "grad all-reduce")},
            "peak_bwd_mem_mb": round(peak_mb, 2),
            "fwd_ms": round(fwd_ms, 4), "bwd_ms": round(bwd_ms, 4),
            "roofline": roof,
        }
        if isinstance(wl, RendererWorkload):
            res["mlp_fp32_frac_of_peak"] = round(wl.mlp_flops_fwdbwd() / ((fwd_ms + bwd_ms) * 1e-3) / FP32_PEAK, 5)
        return res

    if timer is not None:
        if rank == 0:
            line_ready["res"] = headline()
        timer.start()
        sharded = {}
        for key, name, st in (("renderer_1080p_s128", "1080p_s128", 3), ("renderer_cfg4_shard", "cfg4", 2)):
            try:
                sharded[key] = measure_sharded(name, rank, world, dev, pg, args.kernel, st)
            except Exception as e:  # (a failing rank leaves the others in a collective: the watchdog ends the run)
                sharded[key] = {"error": repr(e)}
                break
        done.set()
        timer.cancel()

    if rank == 0:
        res = headline()
        if sharded is not None:
            res["extras"] = sharded
        if world == 1 and args.workload == "cfg2" and not args.no_extras:
            res["extras"] = {
                "splatter_cfg3": measure_extra("cfg3", dev, args.kernel, 10),
                "renderer_1080p_s128": measure_extra("1080p_s128", dev, args.kernel, 4),
                "renderer_cfg4_shard": measure_extra("cfg4", dev, args.kernel, 3),
                "renderer_small_batch": measure_small_batch(dev, args.kernel, 20),
            }
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            res["cpu_baseline"] = cpu_baseline(wl)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
