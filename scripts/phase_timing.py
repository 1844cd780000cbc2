"""Developer tool: per-phase shader cycles of the MFMA backward (library built with
LP_BUILD_FLAGS=-DLP_PHASE_TIMING).  Prints average cycles per (wave, sample) for every phase."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from bench import make_workload, S, COLOR

dev = torch.device("cuda:0")
lp.config.check_inputs = False
rays_c, grids_c, dec_c, sizes, up_c = make_workload(0, dev)
rays = rays_c.to(dev)
flat, _ = lp.flatten_grid([g.to(dev) for g in grids_c])
flat.requires_grad_(True)
params = dec_c.mlp_params.to(dev).requires_grad_(True)
rays.encoding.requires_grad_(True)
dec = lp.DecoderParams(params, dec_c.n_hidden_trunk, dec_c.n_hidden_opacity, dec_c.n_hidden_color, COLOR)
up = [u.to(dev) for u in up_c]
L = _lib.lib()
buf = (ctypes.c_ulonglong * 16)()
L.lp_debug_phase_cycles.argtypes = [ctypes.c_void_p]
n = 5
for it in range(n + 1):
    if it == 1:
        assert L.lp_debug_phase_cycles(buf) == 0, "library not built with -DLP_PHASE_TIMING"
    flat.grad = params.grad = rays.encoding.grad = None
    o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    ((o[0] * up[0]).sum() + (o[1] * up[1]).sum() + (o[2] * up[2]).sum()).backward()
rc = L.lp_debug_phase_cycles(buf)
print('rc', rc, L.lp_last_error())
names = ["fwd", "compositing", "heads_bwd", "c1", "o1", "t2", "t1", "fetch", "scatter", "outside"]
waves = (rays.n_rays + 31) // 32
tot = 0
for i, nm in enumerate(names):
    c = buf[i] / (n * waves * S)
    tot += c
    print(f"{nm:12s} {c:10.0f} cycles / wave-sample")
print(f"{'total':12s} {tot:10.0f}")
