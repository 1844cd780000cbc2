"""Two-grid decoder (separate colour grid-list, no trunk) on the cfg-2 workload: triplane 64^2 x 16 ch for opacity and
for colour, heads 16-32-1 / 16-32-3, S=128, 256x256 rays; MFMA family vs shape-generic kernels, fwd+bwd."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0"); lp.config.check_inputs = False; lp.config.warn_generic_kernel = False
C = 16; S = 128; n = 256; H = 32
gen = torch.Generator().manual_seed(0)
rays = pinhole_rays(n, n, enc_dim=C, gen=gen).to(dev)
rays.encoding.requires_grad_(True)
sizes = grid_sizes_for((1, 64, 64, 64, C), True)
flat, _ = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])
cflat, _ = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])
flat.requires_grad_(True); cflat.requires_grad_(True)
d = random_decoder(gen, 0, 2, 2, C, H, 3, use_separate_color_grid=True, std=0.1)
params = d.mlp_params.to(dev).requires_grad_(True)
dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
def t(f, k=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
res = {"config": "two-grid decoder, triplane 64^2x16 (x2), heads 16-32-1 / 16-32-3, S=128, 256x256 rays"}
for kn, kern in (("mfma", _lib.LP_KERNEL_AUTO), ("generic", _lib.LP_KERNEL_GENERIC)):
    def fb():
        flat.grad = cflat.grad = params.grad = rays.encoding.grad = None
        o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes, color_grid=cflat,
                                   color_grid_sizes=sizes, kernel=kern)
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()
    res[f"{kn}_fwd_bwd_ms"] = round(t(fb, 3 if kn == "mfma" else 1), 2)
print(json.dumps(res))
