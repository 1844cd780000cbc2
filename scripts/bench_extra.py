"""Timings of the other BASELINE.json configurations on one GPU (not the bench.py headline):
  cfg3  Splatter fwd+bwd: 256x256 rays x 32ch -> voxel 128^3 x 32ch, 256 samples
  cfg4s one GPU's shard of cfg 4: Renderer fwd+bwd, 1920x1080 rays, triplane 128^2 x 32ch, 256 samples
  cfg5  joint Splatter -> Renderer on ONE GPU: 100 views of 512x512 rays x 32ch splatted (view by view) into a
        256^3 x 32ch voxel grid, then a 1920x1080 render of that grid, end-to-end backward (S = 256)
Prints one JSON line per configuration."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids

dev = torch.device("cuda:0")
lp.config.check_inputs = False
which = sys.argv[1:] or ["cfg3", "cfg4s", "cfg5"]


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(n):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / n


if "cfg3" in which:
    gen = torch.Generator().manual_seed(0)
    rays = pinhole_rays(256, 256, enc_dim=32, gen=gen).to(dev)
    rays.encoding = torch.rand(rays.n_rays, 32, generator=gen).to(dev).requires_grad_(True)
    sizes = [[1, 128, 128, 128, 32]]
    up = torch.randn(128 ** 3, 32, device=dev)
    S = 256
    state = {}

    def fwd():
        state["out"] = lp.lightplane_splatter(rays, sizes, num_samples=S, return_list=False)

    def fwdbwd():
        rays.encoding.grad = None
        out = lp.lightplane_splatter(rays, sizes, num_samples=S, return_list=False)
        (out * up).sum().backward()

    t_f = timeit(fwd)
    t_fb = timeit(fwdbwd)
    n = rays.n_rays
    alg = n * S * (8 * 32 * 4 + 8 * 4) + n * S * 8 * 32 * 4
    print(json.dumps({"config": "cfg3 splatter 256x256 rays x32ch -> 128^3x32 voxel, S=256", "fwd_ms": round(t_f, 3),
                      "fwd_bwd_ms": round(t_fb, 3), "Mrays_per_s_fwd_bwd": round(n / t_fb / 1e3, 3),
                      "algorithmic_GB": round(alg / 1e9, 2), "effective_GBps": round(alg / t_fb / 1e6, 1),
                      "note": "fwd_bwd includes the loss (out*up).sum() and torch's zero-fills of the 268 MB grid"}))

if "cfg4s" in which:
    gen = torch.Generator().manual_seed(0)
    H, W, S, C = 1080, 1920, 256, 32
    rays = pinhole_rays(H, W, enc_dim=32, gen=gen).to(dev)
    rays.encoding.requires_grad_(True)
    sizes = grid_sizes_for((1, 128, 128, 128, C), True)
    grids = random_grids(gen, sizes)
    dec_c = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.15)
    flat, _ = lp.flatten_grid([g.to(dev) for g in grids])
    flat.requires_grad_(True)
    params = dec_c.mlp_params.to(dev).requires_grad_(True)
    dec = lp.DecoderParams(params, dec_c.n_hidden_trunk, dec_c.n_hidden_opacity, dec_c.n_hidden_color, 3)
    up = [torch.randn(rays.n_rays, device=dev), torch.randn(rays.n_rays, device=dev), torch.randn(rays.n_rays, 3, device=dev)]

    def fwd():
        with torch.no_grad():
            lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)

    def fwdbwd():
        flat.grad = params.grad = rays.encoding.grad = None
        o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
        ((o[0] * up[0]).sum() + (o[1] * up[1]).sum() + (o[2] * up[2]).sum()).backward()

    t_f = timeit(fwd, n=3, warm=1)
    t_fb = timeit(fwdbwd, n=3, warm=1)
    n = rays.n_rays
    alg = n * (S * 12 * C * 4 * 3 + 500)
    print(json.dumps({"config": "cfg4 shard: renderer 1920x1080 rays, triplane 128^2x32ch, S=256", "fwd_ms": round(t_f, 2),
                      "fwd_bwd_ms": round(t_fb, 2), "Mrays_per_s_fwd_bwd": round(n / t_fb / 1e3, 3),
                      "algorithmic_GB": round(alg / 1e9, 1), "effective_GBps": round(alg / t_fb / 1e6, 1)}))

if "cfg5" in which:
    import math
    gen = torch.Generator().manual_seed(0)
    S, C, G = 256, 32, 256
    sizes = [[1, G, G, G, C]]
    n_views = int(os.environ.get("CFG5_VIEWS", "100"))
    views = []
    for v in range(n_views):
        r = pinhole_rays(512, 512, enc_dim=C, gen=gen, azimuth_deg=360.0 * v / n_views,
                         elevation_deg=40.0 * math.sin(2 * math.pi * v / n_views)).to(dev)
        r.encoding = torch.rand(r.n_rays, C, generator=gen).to(dev).requires_grad_(True)
        views.append(r)
    cam = pinhole_rays(1080, 1920, enc_dim=32, gen=gen).to(dev)
    dec_c = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.15)
    params = dec_c.mlp_params.to(dev).requires_grad_(True)
    dec = lp.DecoderParams(params, dec_c.n_hidden_trunk, dec_c.n_hidden_opacity, dec_c.n_hidden_color, 3)
    torch.cuda.reset_peak_memory_stats()

    def joint():
        params.grad = None
        # the joint op splats ALL views into one grid (one normalisation): concatenate the rays and splat once
        rays = lp.Rays(torch.cat([v.directions for v in views]), torch.cat([v.origins for v in views]),
                      torch.cat([v.grid_idx for v in views]), torch.cat([v.near for v in views]),
                      torch.cat([v.far for v in views]), torch.cat([v.encoding for v in views]))
        grid = lp.lightplane_splatter(rays, sizes, num_samples=S, return_list=False)
        o = lp.lightplane_renderer(cam, grid, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()

    t = timeit(joint, n=2, warm=1)
    n_splat = n_views * 512 * 512
    print(json.dumps({"config": f"cfg5 on one GPU: {n_views} views 512x512 -> 256^3x32 voxel, then 1920x1080 render, fwd+bwd",
                      "ms": round(t, 1), "splat_Mrays": round(n_splat / 1e6, 1), "render_Mrays": round(cam.n_rays / 1e6, 2),
                      "Mrays_per_s_total": round((n_splat + cam.n_rays) / t / 1e3, 2),
                      "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}))
