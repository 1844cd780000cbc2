"""Time the cfg-2 Renderer forward / backward with parts of the backward switched off through
requires_grad (the C ABI skips outputs whose pointer is NULL).  GPU only."""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from bench import make_workload, S, COLOR

dev = torch.device("cuda:0")
lp.config.check_inputs = False
rays_c, grids_c, dec_c, sizes, up_c = make_workload(0, dev)
rays = rays_c.to(dev)
flat0, _ = lp.flatten_grid([g.to(dev) for g in grids_c])
up = [u.to(dev) for u in up_c]
res = {}
for name, (gg, gp, ge) in {"all": (1, 1, 1), "no_grid": (0, 1, 1), "no_params": (1, 0, 1), "enc_only": (0, 0, 1), "grid_only": (1, 0, 0)}.items():
    flat = flat0.clone().requires_grad_(bool(gg))
    params = dec_c.mlp_params.to(dev).clone().requires_grad_(bool(gp))
    rays.encoding = rays_c.encoding.to(dev).clone().requires_grad_(bool(ge))
    dec = lp.DecoderParams(params, dec_c.n_hidden_trunk, dec_c.n_hidden_opacity, dec_c.n_hidden_color, COLOR)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    f = b = 0.0
    n = 8
    for it in range(n + 2):
        flat.grad = params.grad = rays.encoding.grad = None
        ev[0].record()
        o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
        ev[1].record()
        loss = (o[0] * up[0]).sum() + (o[1] * up[1]).sum() + (o[2] * up[2]).sum()
        ev[2].record()
        loss.backward()
        ev[3].record()
        torch.cuda.synchronize()
        if it >= 2:
            f += ev[0].elapsed_time(ev[1]); b += ev[2].elapsed_time(ev[3])
    res[name] = {"fwd_ms": round(f / n, 3), "bwd_ms": round(b / n, 3)}
print(json.dumps(res))
