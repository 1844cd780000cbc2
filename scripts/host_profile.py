"""cProfile of the host side of Renderer calls (tiny workload): where the ~0.3 ms of Python per forward+backward go."""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
C, S = 16, 16
rays = pinhole_rays(32, 32, enc_dim=32, gen=gen).to(dev)
rays.encoding.requires_grad_(True)
sizes = grid_sizes_for((1, 64, 64, 64, C), True)
planes = [g.to(dev).requires_grad_(True) for g in random_grids(gen, sizes)]
flat = lp.flatten_grid([p.detach() for p in planes])[0].requires_grad_(True)
d = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.1)
params = d.mlp_params.to(dev).requires_grad_(True)
dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
module = lp.LightplaneRenderer(num_samples=S, color_chn=3, grid_chn=C, mlp_hidden_chn=32).to(dev)
rays_m = lp.Rays(directions=rays.directions, origins=rays.origins, grid_idx=rays.grid_idx, near=rays.near, far=rays.far, encoding=None)
lp.config.check_inputs = False
which = sys.argv[1] if len(sys.argv) > 1 else "flat"
def f():
    if which == "flat":
        o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    else:
        o = module(rays_m, planes)
    (o[0].sum() + o[2].sum()).backward()
for _ in range(20): f()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(300): f()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(32); print(s.getvalue()[:6000])
