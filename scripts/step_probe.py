"""Host-side cost of one Renderer call: forward+backward wall time for a tiny workload (1 024 rays, 16 samples), where the
kernels take ~20 us -- functional API on a flat grid, functional API on a list of three planes, module API."""
import os, sys, time
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
def f_flat():
    o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    (o[0].sum() + o[2].sum()).backward()
def f_list():
    o = lp.lightplane_renderer(rays, planes, dec, num_samples=S, gain=1.0)
    (o[0].sum() + o[2].sum()).backward()
def f_module():
    o = module(rays_m, planes)
    (o[0].sum() + o[2].sum()).backward()
def count_kernels(f):
    """GPU kernels (names) of one call, from torch.profiler."""
    from torch.profiler import ProfilerActivity, profile
    f(); torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        f(); torch.cuda.synchronize()
    names = [e.name for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA and "Memcpy" not in e.name and "Memset" not in e.name]
    return names


lp.config.check_inputs = False
for fused in (True, False):
    lp.config.fused_module_ops = fused
    names = count_kernels(f_module)
    ours = [n for n in names if "lp::" in n]
    print(f"module fwd+bwd, fused_module_ops={fused}: {len(names)} kernels ({len(ours)} lp::, cat copies: {sum('Cat' in n for n in names)})")
    if fused:
        print("   ", [n.split("(")[0][:60] for n in names])
lp.config.fused_module_ops = True
names = count_kernels(f_list)
print(f"functional, list of planes: {len(names)} kernels, cat copies: {sum('Cat' in n for n in names)}")
for chk in (True, False):
    lp.config.check_inputs = chk
    def f_module_ops():
        lp.config.fused_module_ops = False
        try:
            f_module()
        finally:
            lp.config.fused_module_ops = True
    for name, f in (("functional, flat grid", f_flat), ("functional, list of planes", f_list), ("module (fused)", f_module),
                    ("module (PyTorch op chain)", f_module_ops)):
        for _ in range(20): f()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(200): f()
        torch.cuda.synchronize()
        print(f"check_inputs={chk}  {name:28s} {(time.perf_counter() - t) / 200 * 1e6:8.1f} us per fwd+bwd")
