"""MLP-Splatter (LightplaneMLPSplatter's kernels) on a cfg-3-sized launch: image H x W rays x E ch, input voxel grid Gi^3 x E ->
MLP (n_layers, hidden) -> output voxel grid Go^3 x C, S samples.  Kernel times by torch.profiler (device timestamps), forward and
backward; no reference configuration quotes this path -- a sanity number beside the Splatter's (profiles/r06_mlp_splatter.txt)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import pinhole_rays, random_splatter_mlp

dev = torch.device("cuda:0"); lp.config.check_inputs = False
CASES = [(256, 256, 64, 128, 32, 2, 32, 128), (256, 256, 64, 128, 32, 3, 64, 128), (128, 128, 32, 64, 16, 2, 32, 64),
         (256, 256, 64, 128, 32, 2, 32, 256)]  # (the last one: the launch of rounds 2-5's version of this script, profiles/r02_other_workloads.txt)
for H, W, Gi, Go, E, nl, hid, S in CASES:
    gen = torch.Generator().manual_seed(H + Go)
    rays = pinhole_rays(H, W, gen=gen, azimuth_deg=25.0, elevation_deg=20.0)
    rays.encoding = torch.rand(rays.n_rays, E, generator=gen)
    rays = rays.to(dev)
    rays.encoding.requires_grad_(True)
    mlp = random_splatter_mlp(gen, nl, E, hid, E, std=0.2)
    mlp.mlp_params = mlp.mlp_params.to(dev).requires_grad_(True)
    gin = (0.3 * torch.randn(1, Gi, Gi, Gi, E, generator=gen)).to(dev).requires_grad_(True)
    sizes = [[1, Go, Go, Go, E]]
    up = torch.randn(Go ** 3, E, device=dev)

    def step():
        rays.encoding.grad = mlp.mlp_params.grad = gin.grad = None
        out = lp.lightplane_mlp_splatter(rays, sizes, mlp, [gin], num_samples=S, return_list=False)
        out.backward(up)

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for _ in range(5):
            step()
        torch.cuda.synchronize()
    rows = {}
    for e in prof.key_averages():
        if e.key.startswith("lp::") or "lp::" in e.key:
            rows[e.key[:70]] = round(e.device_time_total / 5 / 1e3, 4)
    print(json.dumps({"case": f"{H}x{W} rays x {E} ch, in {Gi}^3 -> MLP {nl} x {hid} -> out {Go}^3, S {S}", "kernel_ms": rows}))
