"""The reference's own example configuration (examples/config/synthetic_overfit.json): triplane 128^2 x 32 ch,
hidden 64, 128 samples -- timing of forward / forward+backward for 256x256 rays."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0"); lp.config.check_inputs = False
H = int(os.environ.get("HID", "64")); C = 32; S = 128; n = int(os.environ.get("NPIX", "256"))
gen = torch.Generator().manual_seed(0)
rays = pinhole_rays(n, n, enc_dim=H, gen=gen).to(dev)
rays.encoding.requires_grad_(True)
sizes = grid_sizes_for((1, 128, 128, 128, C), True)
flat, _ = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])
flat.requires_grad_(True)
d = random_decoder(gen, 2, 2, 2, C, H, 3, std=0.1)
params = d.mlp_params.to(dev).requires_grad_(True)
dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
def fwd():
    with torch.no_grad():
        lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
def fb():
    flat.grad = params.grad = rays.encoding.grad = None
    o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    (o[0].sum() + o[1].sum() + o[2].sum()).backward()
def t(f, k=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
tf, tb = t(fwd), t(fb)
print(json.dumps({"config": f"triplane 128^2x{C}, hidden {H}, S={S}, {n}x{n} rays", "fwd_ms": round(tf, 2),
                  "fwd_bwd_ms": round(tb, 2), "Mrays_per_s_fwd_bwd": round(n * n / tb / 1e3, 3)}))
