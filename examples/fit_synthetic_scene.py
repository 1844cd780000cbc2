#!/usr/bin/env python
"""End-to-end check of the MI355X Renderer: fit a triplane + decoder to an analytic scene.

The target is a soft coloured ball rendered with a plain PyTorch emission-absorption march; the student is a
``LightplaneRenderer`` module (direction-dependent colours through the harmonic ray embedding) with three
plane grids as parameters, optimised with Adam on random rays.  The role of the reference's
examples/fit_single_scene.py training loop (:282-334) as a convergence check, on synthetic data because the
GPU boxes have no datasets.

    python examples/fit_synthetic_scene.py [--steps 300] [--rays 8192] [--stop-transmittance 0]

Prints one JSON line with the first / last losses and the PSNR of a held-out ray batch.
"""
import argparse
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightplane_amd as lp  # noqa: E402


def scene(p):
    """density and colour of the analytic scene at points p [..., 3]"""
    r = p.norm(dim=-1)
    sigma = 25.0 * torch.sigmoid((0.55 - r) * 25.0)
    rgb = 0.5 + 0.5 * torch.sin(5.0 * p + torch.tensor([0.0, 2.0, 4.0], device=p.device))
    return sigma, rgb


def render_target(origins, directions, near, far, num_samples):
    t = torch.linspace(0.0, 1.0, num_samples, device=origins.device)
    depth = near[:, None] + (far - near)[:, None] * t[None]
    p = origins[:, None] + depth[..., None] * directions[:, None]
    sigma, rgb = scene(p)
    delta = torch.cat([(far - near)[:, None] / (num_samples - 1), depth[:, 1:] - depth[:, :-1]], dim=1)
    nlt = torch.cumsum(sigma * delta, dim=1)
    trans = torch.exp(-torch.cat([torch.zeros_like(nlt[:, :1]), nlt], dim=1))
    w = trans[:, :-1] - trans[:, 1:]
    return (w[..., None] * rgb).sum(1), 1.0 - trans[:, -1]


def random_rays(n, gen, dev):
    o = torch.randn(n, 3, generator=gen)
    o = 2.5 * o / o.norm(dim=-1, keepdim=True)
    tgt = torch.randn(n, 3, generator=gen)
    tgt = 0.8 * tgt / tgt.norm(dim=-1, keepdim=True) * torch.rand(n, 1, generator=gen) ** (1 / 3)
    d = tgt - o
    d = d / d.norm(dim=-1, keepdim=True)
    near, far = torch.full((n,), 1.2), torch.full((n,), 3.8)
    return lp.Rays(directions=d.to(dev), origins=o.to(dev), grid_idx=torch.zeros(n, dtype=torch.int32, device=dev),
                   near=near.to(dev), far=far.to(dev), encoding=None)


def fit(steps=300, n_rays=8192, num_samples=96, res=64, chn=16, seed=0, stop_transmittance=0.0, verbose=False):
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(seed)
    torch.manual_seed(seed)
    lp.config.stop_transmittance = float(stop_transmittance)
    renderer = lp.LightplaneRenderer(num_samples=num_samples, color_chn=3, grid_chn=chn, mlp_hidden_chn=32,
                                     opacity_init_bias=-2.0, gain=1.0, bg_color=0.0).to(dev)
    shapes = [(1, 1, res, res, chn), (1, res, 1, res, chn), (1, res, res, 1, chn)]
    grids = torch.nn.ParameterList([torch.nn.Parameter(0.1 * torch.randn(*s, generator=gen).to(dev)) for s in shapes])
    opt = torch.optim.Adam([{"params": grids.parameters(), "lr": 3e-2}, {"params": renderer.parameters(), "lr": 3e-3}])
    losses = []
    for it in range(steps):
        rays = random_rays(n_rays, gen, dev)
        with torch.no_grad():
            tgt_rgb, tgt_alpha = render_target(rays.origins, rays.directions, rays.near, rays.far, num_samples)
        _, alpha, rgb = renderer(rays, list(grids))
        loss = ((rgb - tgt_rgb) ** 2).mean() + 0.1 * ((alpha - tgt_alpha) ** 2).mean()
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if verbose and (it % 50 == 0 or it == steps - 1):
            print(f"step {it:4d}  loss {losses[-1]:.5f}", flush=True)
    rays = random_rays(n_rays, torch.Generator().manual_seed(seed + 1), dev)
    with torch.no_grad():
        tgt_rgb, _ = render_target(rays.origins, rays.directions, rays.near, rays.far, num_samples)
        _, _, rgb = renderer(rays, list(grids))
        mse = float(((rgb - tgt_rgb) ** 2).mean())
    return {"first_loss": sum(losses[:5]) / 5, "last_loss": sum(losses[-5:]) / 5, "heldout_psnr_db": -10.0 * math.log10(mse),
            "steps": steps, "rays_per_step": n_rays, "stop_transmittance": stop_transmittance}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--stop-transmittance", type=float, default=0.0)
    a = ap.parse_args()
    print(json.dumps(fit(a.steps, a.rays, stop_transmittance=a.stop_transmittance, verbose=True)))
