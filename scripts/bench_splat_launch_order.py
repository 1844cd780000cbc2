"""Splatter forward walk: how the launch is cut into interleaved sample segments and in which order the (ray block, segment) workgroups
are issued (LP_SPLAT_FWD_SEGMENTS / LP_SPLAT_FWD_GROUP are read once per process: one run of this script per setting, see
profiles/r06_splat_launch_order.txt).  Forward only, HIP events; image H x W -> voxel grid G^3 x C, S samples."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import pinhole_rays

dev = torch.device("cuda:0"); lp.config.check_inputs = False
CASES = [(128, 128, 128, 32, 256), (256, 256, 128, 32, 256), (256, 256, 64, 32, 128), (256, 256, 256, 32, 256), (512, 512, 128, 32, 256),
         (512, 512, 256, 32, 256), (1024, 1024, 128, 32, 128), (256, 256, 128, 16, 256), (360, 640, 160, 32, 192)]
if len(sys.argv) > 1:
    CASES = [tuple(int(v) for v in c.split("x")) for c in sys.argv[1].split(",")]
out = {}
for H, W, G, C, S in CASES:
    gen = torch.Generator().manual_seed(H + G)
    rays = pinhole_rays(H, W, gen=gen, azimuth_deg=25.0, elevation_deg=20.0)
    rays.encoding = torch.rand(rays.n_rays, C, generator=gen)
    rays = rays.to(dev)
    sizes = [[1, G, G, G, C]]
    f = lambda: lp.lightplane_splatter(rays, sizes, num_samples=S, return_list=False, rays_per_row=W)
    for _ in range(3):
        f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        f()
    e1.record(); torch.cuda.synchronize()
    fwd = e0.elapsed_time(e1) / n
    # backward (gather walk): LP_SPLAT_SEGMENTS forces its segment count
    rays.encoding.requires_grad_(True)
    up = torch.randn(G ** 3, C, device=dev)
    def fb():
        rays.encoding.grad = None
        (f() * up).sum().backward()
    for _ in range(3):
        fb()
    torch.cuda.synchronize(); e0.record()
    for _ in range(n):
        fb()
    e1.record(); torch.cuda.synchronize()
    out[f"{H}x{W}->{G}^3x{C},S{S}"] = (round(fwd, 4), round(e0.elapsed_time(e1) / n - fwd, 4))
    rays.encoding.requires_grad_(False)
    del up
    torch.cuda.empty_cache()
print(json.dumps({"seg": os.environ.get("LP_SPLAT_FWD_SEGMENTS"), "grp": os.environ.get("LP_SPLAT_FWD_GROUP"), "bwd_seg": os.environ.get("LP_SPLAT_SEGMENTS"),
                  "fwd_bwd_ms": out}))  # (bwd = step - fwd: includes the loss kernels)
