"""Worker of tests/test_gpu_modules.py::test_two_rank_ray_shards_equal_single_process (launched with
torch.distributed.run, 2 ranks, gloo backend, both ranks on cuda:0): the ray-sharded Renderer / Splatter / MLP-Splatter
through the real autograd functions against the single-process result."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import lightplane_amd as lp  # noqa: E402
from lightplane_amd import parallel  # noqa: E402
from tests.synth import RENDERER_CASES, SPLATTER_CASES  # noqa: E402


def close(name, a, b, tol=2e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    scale = max(b.abs().max().item(), 1e-30)
    err = (a - b).abs().max().item() / scale
    assert err <= tol, f"{name}: {err:.3e} > {tol}"


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    dev = torch.device("cuda:0")
    pg = dist.group.WORLD

    # ---------------- Renderer: grads of the replicated grid-list / parameters are summed over the ray shards
    for name in ("triplane_plus_voxel", "colorgrid_c32_mixed"):
        d = next(c for c in RENDERER_CASES if c.name == name).build()
        dec = d["decoder"]
        up = [u.to(dev) for u in d["upstream"]]

        def run(rays, ups, group):
            params = dec.mlp_params.to(dev).clone().requires_grad_(True)
            grids = [g.to(dev).clone().requires_grad_(True) for g in d["grids"]]
            cgrids = None if d["color_grids"] is None else [g.to(dev).clone().requires_grad_(True) for g in d["color_grids"]]
            leaves = grids + (cgrids or []) + [params]
            rep = parallel.replicate_with_grad_allreduce(leaves, group) if group is not None else leaves
            g_r, c_r, p_r = rep[: len(grids)], rep[len(grids): len(grids) + len(cgrids or [])], rep[-1]
            hdec = lp.DecoderParams(p_r, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
            r = rays.to(dev)
            r.encoding = r.encoding.clone().requires_grad_(True)
            scaffold = None if d["scaffold"] is None else d["scaffold"].to(dev)
            out = lp.lightplane_renderer(r, list(g_r), hdec, color_grid=list(c_r) if cgrids else None, scaffold=scaffold, **d["cfg"])
            ((out[0] * ups[0]).sum() + (out[1] * ups[1]).sum() + (out[2] * ups[2]).sum()).backward()
            return [g.grad for g in grids], [g.grad for g in (cgrids or [])], params.grad, r.encoding.grad

        gg1, gc1, gp1, ge1 = run(d["rays"], up, None)
        lo, hi = parallel.shard_bounds(d["rays"].n_rays, rank, world)
        gg2, gc2, gp2, ge2 = run(d["rays"][lo:hi], [u[lo:hi] for u in up], pg)
        for i, (a, b) in enumerate(zip(gg2, gg1)):
            close(f"{name}: grad_grid{i}", a, b)
        for i, (a, b) in enumerate(zip(gc2, gc1)):
            close(f"{name}: grad_color_grid{i}", a, b)
        close(f"{name}: grad_mlp_params", gp2, gp1)
        close(f"{name}: grad_encoding (local shard)", ge2, ge1[lo:hi])

    # ---------------- Splatter: un-normalised features + weights summed, then normalised
    ds = next(c for c in SPLATTER_CASES if c.name == "voxel_basic").build()
    ups = [u.to(dev) for u in ds["upstream"]]

    def splat(rays, group):
        r = rays.to(dev)
        r.encoding = r.encoding.clone().requires_grad_(True)
        out = lp.lightplane_splatter(r, ds["out_sizes"], process_group=group, **ds["cfg"])
        sum((o * u).sum() for o, u in zip(out, ups)).backward()
        return out, r.encoding.grad

    out1, ge1 = splat(ds["rays"], None)
    lo, hi = parallel.shard_bounds(ds["rays"].n_rays, rank, world)
    out2, ge2 = splat(ds["rays"][lo:hi], pg)
    for a, b in zip(out2, out1):
        close("splatter: out", a, b)
    close("splatter: grad_encoding (local shard)", ge2, ge1[lo:hi])

    # ---------------- MLP-Splatter: mlp / input-grid gradients all-reduced in its backward
    dm = next(c for c in SPLATTER_CASES if c.name == "mlp2_triplane_c16").build()
    ups = [u.to(dev) for u in dm["upstream"]]

    def mlp_splat(rays, group):
        r = rays.to(dev)
        r.encoding = r.encoding.clone().requires_grad_(True)
        params = dm["mlp"].mlp_params.to(dev).clone().requires_grad_(True)
        in_grids = [g.to(dev).clone().requires_grad_(True) for g in dm["in_grids"]]
        out = lp.lightplane_mlp_splatter(r, dm["out_sizes"], lp.SplatterParams(params, dm["mlp"].n_hidden), in_grids,
                                         process_group=group, **dm["cfg"])
        sum((o * u).sum() for o, u in zip(out, ups)).backward()
        return out, params.grad, [g.grad for g in in_grids], r.encoding.grad

    o1, p1, g1, e1 = mlp_splat(dm["rays"], None)
    lo, hi = parallel.shard_bounds(dm["rays"].n_rays, rank, world)
    o2, p2, g2, e2 = mlp_splat(dm["rays"][lo:hi], pg)
    for a, b in zip(o2, o1):
        close("mlp-splatter: out", a, b)
    close("mlp-splatter: grad_mlp_params", p2, p1)
    for a, b in zip(g2, g1):
        close("mlp-splatter: grad_input_grid", a, b)
    close("mlp-splatter: grad_encoding (local shard)", e2, e1[lo:hi])

    # ---------------- joint Splatter -> Renderer (BASELINE configs[4]): every rank splats its share of the rays into the
    # grid (sum over ranks, then normalise), then renders its share of the camera rays from the replicated grid; end to end
    # backward: the grid gradient is summed over the ranks before it enters the (local) splat backward
    from tests.synth import pinhole_rays, random_decoder
    gen = torch.Generator().manual_seed(77)
    sizes = [[1, 12, 10, 14, 16]]
    srays = pinhole_rays(20, 24, gen=gen, azimuth_deg=40.0, elevation_deg=20.0)
    srays.encoding = torch.rand(srays.n_rays, 16, generator=gen)
    cam = pinhole_rays(16, 24, enc_dim=32, gen=gen, azimuth_deg=-60.0, elevation_deg=35.0)
    jdec = random_decoder(gen, 2, 2, 2, 16, 32, 3, std=0.2)
    jup = [torch.randn(cam.n_rays, generator=gen).to(dev), torch.randn(cam.n_rays, generator=gen).to(dev),
           torch.randn(cam.n_rays, 3, generator=gen).to(dev)]

    def joint(sr, cr, ups, group):
        r = sr.to(dev)
        r.encoding = r.encoding.clone().requires_grad_(True)
        params = jdec.mlp_params.to(dev).clone().requires_grad_(True)
        grid = lp.lightplane_splatter(r, sizes, num_samples=20, return_list=False, process_group=group)
        g, p = (grid, params) if group is None else parallel.replicate_with_grad_allreduce([grid, params], group)
        hdec = lp.DecoderParams(p, jdec.n_hidden_trunk, jdec.n_hidden_opacity, jdec.n_hidden_color, 3)
        c = cr.to(dev)
        out = lp.lightplane_renderer(c, g, hdec, num_samples=24, gain=2.0, grid_sizes=sizes)
        ((out[0] * ups[0]).sum() + (out[1] * ups[1]).sum() + (out[2] * ups[2]).sum()).backward()
        return out, params.grad, r.encoding.grad

    o1, p1, e1 = joint(srays, cam, jup, None)
    slo, shi = parallel.shard_bounds(srays.n_rays, rank, world)
    clo, chi = parallel.shard_bounds(cam.n_rays, rank, world)
    o2, p2, e2 = joint(srays[slo:shi], cam[clo:chi], [u[clo:chi] for u in jup], pg)
    for a, b in zip(o2, o1):
        close("joint: render output (local shard)", a, b[clo:chi])
    close("joint: grad_mlp_params", p2, p1)
    close("joint: grad of the splatted features (local shard)", e2, e1[slo:shi], tol=5e-5)

    dist.barrier()
    if rank == 0:
        print("DIST_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
