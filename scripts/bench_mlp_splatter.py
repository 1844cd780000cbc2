"""MLP-Splatter timing: 256x256 rays x 32ch, input voxel grid 64^3 x 32ch, MLP 32->32->32, out 128^3 x 32ch, S=256."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import pinhole_rays
dev = torch.device("cuda:0"); lp.config.check_inputs = False
n = int(os.environ.get("NPIX", "256")); S = int(os.environ.get("S", "256"))
gen = torch.Generator().manual_seed(0)
rays = pinhole_rays(n, n, enc_dim=32, gen=gen).to(dev)
rays.encoding = torch.rand(rays.n_rays, 32, generator=gen).to(dev).requires_grad_(True)
in_grid = torch.randn(1, 64, 64, 64, 32, generator=gen).to(dev).requires_grad_(True)
sp = lp.init_splatter_params("cpu", 2, 32, 32, 32)
params = sp.mlp_params.to(dev).requires_grad_(True)
mlp = lp.SplatterParams(params, sp.n_hidden)
sizes = [[1, 128, 128, 128, 32]]
up = torch.randn(128 ** 3, 32, device=dev)
def fwd():
    with torch.no_grad():
        lp.lightplane_mlp_splatter(rays, sizes, mlp, [in_grid], num_samples=S, return_list=False)
def fb():
    rays.encoding.grad = params.grad = in_grid.grad = None
    out = lp.lightplane_mlp_splatter(rays, sizes, mlp, [in_grid], num_samples=S, return_list=False)
    (out * up).sum().backward()
def t(f, k=2):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
tf, tb = t(fwd), t(fb)
print(json.dumps({"config": f"MLP-splatter {n}x{n} rays, in 64^3x32, MLP 32-32-32, out 128^3x32, S={S}", "fwd_ms": round(tf, 2),
                  "fwd_bwd_ms": round(tb, 2), "Mrays_per_s_fwd_bwd": round(n * n / tb / 1e3, 3)}))
