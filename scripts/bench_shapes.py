"""Decoder shapes across the Renderer's kernel families on the cfg-2 workload (256x256 rays, triplane G^2 x C, S = 128):
forward / forward+backward time, the family LP_KERNEL_AUTO picks, and time per multiply-accumulate of the decoder --
the evidence for "a 4-layer MLP costs its FLOPs, not a fall-back" (layer-looped family, lp_renderer_loop.hip).
    python scripts/bench_shapes.py [renderer|splatter] ; LP_LOOP=1 python scripts/bench_shapes.py   (everything through the loop family)
"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids, random_splatter_mlp

dev = torch.device("cuda:0"); lp.config.check_inputs = False; lp.config.warn_generic_kernel = False
n = int(os.environ.get("NPIX", "256")); S = int(os.environ.get("S", "128"))
what = sys.argv[1] if len(sys.argv) > 1 else "renderer"


def t(f, k=10):
    for _ in range(3): f()  # (round 3 timed 3 reps after ONE warm-up call: the first shape of a process read 15-30 % long)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k


if what == "list_vs_flat":
    # round-3 review, weak 10: this script reported 3.00 ms fwd+bwd (0.63 fwd) for the headline shape given as a LIST of grids,
    # bench.py 2.55 ms (0.48 fwd) for the flat tensor.  Same inputs, both input forms, warmed up (10 + `k` reps each, interleaved).
    gen = torch.Generator().manual_seed(0)
    d = random_decoder(gen, 2, 2, 2, 16, 32, 3, std=0.15)
    rays = pinhole_rays(n, n, enc_dim=32, gen=gen).to(dev)
    rays.encoding.requires_grad_(True)
    sizes = grid_sizes_for((1, 64, 64, 64, 16), True)
    grids = [g.to(dev).requires_grad_(True) for g in random_grids(gen, sizes)]
    flat = lp.flatten_grid([g.detach() for g in grids])[0].clone().requires_grad_(True)
    params = d.mlp_params.to(dev).requires_grad_(True)
    dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)

    def mk(g, kw):
        def fwd():
            with torch.no_grad():
                lp.lightplane_renderer(rays, g, dec, num_samples=S, gain=1.0, **kw)

        def fb():
            params.grad = rays.encoding.grad = flat.grad = None
            for x in grids:
                x.grad = None
            o = lp.lightplane_renderer(rays, g, dec, num_samples=S, gain=1.0, **kw)
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()
        return fwd, fb

    forms = {"list": mk(grids, {}), "flat": mk(flat, {"grid_sizes": sizes})}
    for _ in range(10):
        for fwd, fb in forms.values():
            fwd(); fb()
    for rnd in range(2):
        for name, (fwd, fb) in forms.items():
            print(json.dumps({"input": name, "round": rnd, "fwd_ms": round(t(fwd, 50), 4), "fwd_bwd_ms": round(t(fb, 50), 4),
                              "fwd_ms_3reps_cold_protocol": round(t(fwd, 3), 4)}), flush=True)
elif what == "renderer":
    #        trunk, opacity, colour, hidden, C, grid, separate colour grid
    shapes = [(2, 2, 2, 32, 16, 64, False), (4, 4, 4, 32, 16, 64, False), (4, 2, 4, 32, 16, 64, False), (2, 4, 2, 32, 16, 64, False),
              (3, 3, 3, 16, 16, 64, False), (0, 4, 4, 32, 16, 64, True), (0, 2, 2, 32, 16, 64, True), (4, 4, 4, 32, 32, 128, False),
              (2, 2, 2, 64, 32, 128, False), (1, 1, 1, 16, 16, 64, False)]
    if os.environ.get("SHAPESET") == "shallow":  # the shapes the shallow two-waves-per-SIMD looped backward covers (round 4)
        shapes = [(2, 2, 2, 32, 16, 64, False), (2, 2, 2, 32, 32, 128, False), (1, 1, 1, 16, 16, 64, False), (2, 1, 1, 32, 16, 64, False),
                  (1, 1, 2, 32, 32, 128, False), (1, 2, 1, 16, 32, 128, False), (2, 1, 2, 32, 32, 128, False), (0, 2, 2, 32, 16, 64, True),
                  (0, 2, 2, 32, 32, 128, True), (0, 1, 1, 16, 16, 64, True), (0, 2, 1, 32, 32, 128, True)]
    if os.environ.get("SHAPESET") == "h64":  # the 2/2/2 x 64 decoder on both BASELINE grid shapes (two-block looped kernels)
        shapes = [(2, 2, 2, 64, 16, 64, False), (2, 2, 2, 64, 32, 128, False), (0, 2, 2, 64, 16, 64, True), (0, 2, 2, 64, 32, 128, True)]
    if os.environ.get("SHAPESET") == "deep64":  # hidden 64 with more than two layers per MLP: the shape-generic kernels (NPIX=128 keeps it short)
        shapes = [(2, 2, 2, 64, 32, 128, False), (3, 2, 2, 64, 32, 128, False), (3, 3, 3, 64, 32, 128, False), (4, 4, 4, 64, 32, 128, False)]
    only = os.environ.get("SHAPES")  # e.g. SHAPES="4/2/4,4/4/4": only these layer triples
    for (nt, no, nc, H, C, G, sep) in shapes:
        if only and f"{nt}/{no}/{nc}" not in only.split(","):
            continue
        gen = torch.Generator().manual_seed(0)
        d = random_decoder(gen, nt, no, nc, C, H, 3, use_separate_color_grid=sep, std=0.1)
        enc_dim = int(d.n_hidden_color[0])
        rays = pinhole_rays(n, n, enc_dim=enc_dim, gen=gen).to(dev)
        rays.encoding.requires_grad_(True)
        sizes = grid_sizes_for((1, G, G, G, C), True)
        grids = [g.to(dev).requires_grad_(True) for g in random_grids(gen, sizes)]
        cgrids = [g.to(dev).requires_grad_(True) for g in random_grids(gen, sizes)] if sep else None
        params = d.mlp_params.to(dev).requires_grad_(True)
        dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
        fam = lp.kernel_family(rays, grids, dec, color_grid=cgrids)
        dims = [[int(v) for v in x] for x in (d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color)]
        mac = sum(a * b for x in dims for a, b in zip(x[:-1], x[1:]))

        def fwd():
            with torch.no_grad():
                lp.lightplane_renderer(rays, grids, dec, num_samples=S, gain=1.0, color_grid=cgrids)

        def fb():
            params.grad = rays.encoding.grad = None
            for g in grids + (cgrids or []):
                g.grad = None
            o = lp.lightplane_renderer(rays, grids, dec, num_samples=S, gain=1.0, color_grid=cgrids)
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()

        tf, tb = t(fwd), t(fb)
        print(json.dumps({"layers": f"{nt}/{no}/{nc}", "hidden": H, "C": C, "grid": G, "colour_grid": sep, "family": fam,
                          "fwd_ms": round(tf, 3), "fwd_bwd_ms": round(tb, 3), "Mrays_per_s_fwd_bwd": round(n * n / tb / 1e3, 3),
                          "MAC_per_sample": mac, "ps_per_MAC_fwd_bwd": round(tb * 1e9 / (n * n * S * mac * 4), 4)}), flush=True)
else:
    #        layers, feat, hidden, out
    shapes = [(2, 32, 32, 32), (3, 32, 64, 32), (4, 64, 64, 32), (4, 32, 64, 32), (3, 64, 64, 32), (3, 32, 32, 32), (4, 16, 16, 16)]
    if os.environ.get("SHAPESET") == "shallow":  # two-layer MLPs (looped family, two-waves-per-SIMD backward)
        shapes = [(2, 32, 32, 32), (2, 16, 16, 16), (2, 32, 16, 16), (2, 16, 32, 32)]
    Sx = int(os.environ.get("S", "256"))
    for (nl, E, H, CO) in shapes:
        gen = torch.Generator().manual_seed(0)
        rays = pinhole_rays(n, n, gen=gen)
        rays.encoding = torch.rand(rays.n_rays, E, generator=gen)
        rays = rays.to(dev)
        rays.encoding.requires_grad_(True)
        in_grid = torch.randn(1, 64, 64, 64, E, generator=gen).to(dev).requires_grad_(True)
        sp = random_splatter_mlp(gen, nl, E, H, CO, std=0.1)
        params = sp.mlp_params.to(dev).requires_grad_(True)
        mlp = lp.SplatterParams(params, sp.n_hidden)
        sizes = [[1, 128, 128, 128, CO]]
        up = torch.randn(128 ** 3, CO, device=dev)
        dims = [int(v) for v in sp.n_hidden]
        mac = sum(a * b for a, b in zip(dims[:-1], dims[1:]))

        def fwd():
            with torch.no_grad():
                lp.lightplane_mlp_splatter(rays, sizes, mlp, [in_grid], num_samples=Sx, return_list=False)

        def fb():
            rays.encoding.grad = params.grad = in_grid.grad = None
            out = lp.lightplane_mlp_splatter(rays, sizes, mlp, [in_grid], num_samples=Sx, return_list=False)
            (out * up).sum().backward()

        tf, tb = t(fwd, 2), t(fb, 2)
        print(json.dumps({"mlp": dims, "fwd_ms": round(tf, 3), "fwd_bwd_ms": round(tb, 3), "Mrays_per_s_fwd_bwd": round(n * n / tb / 1e3, 3),
                          "MAC_per_sample": mac, "ps_per_MAC_fwd_bwd": round(tb * 1e9 / (n * n * Sx * mac * 4), 4)}), flush=True)
