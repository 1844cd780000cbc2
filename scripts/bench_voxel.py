"""Renderer on a voxel grid: 256x256 rays, voxel RES^3 x C ch (default 128^3 x 16), S=128, decoder 2/2/2 x 32; fwd+bwd."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import pinhole_rays, random_decoder
dev = torch.device("cuda:0"); lp.config.check_inputs = False
C = int(os.environ.get("C", "16")); RES = int(os.environ.get("RES", "128")); S = 128; n = 256
gen = torch.Generator().manual_seed(0)
rays = pinhole_rays(n, n, enc_dim=32, gen=gen).to(dev)
rays.encoding.requires_grad_(True)
sizes = [[1, RES, RES, RES, C]]
flat = torch.randn(RES ** 3, C, generator=gen).to(dev).requires_grad_(True)
d = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.1)
params = d.mlp_params.to(dev).requires_grad_(True)
dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
def fwd():
    with torch.no_grad():
        lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
def fb():
    flat.grad = params.grad = rays.encoding.grad = None
    o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    (o[0].sum() + o[1].sum() + o[2].sum()).backward()
def t(f, k=5):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
tf, tb = t(fwd), t(fb)
print(json.dumps({"config": f"voxel {RES}^3x{C}, 2/2/2x32, S={S}, {n}x{n} rays", "fwd_ms": round(tf, 3), "fwd_bwd_ms": round(tb, 3),
                  "Mrays_per_s_fwd_bwd": round(n * n / tb / 1e3, 3)}))
