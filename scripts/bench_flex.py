"""Decoder shapes of the width-32 MFMA family other than 2/2/2 x 32 (the reference's notebook decoders): cfg-2 workload
(256x256 rays, triplane 64^2 x 16 ch, S=128), forward+backward, MFMA family vs shape-generic kernels."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from lightplane_amd import _lib
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0"); lp.config.check_inputs = False; lp.config.warn_generic_kernel = False
C = 16; S = 128; n = int(os.environ.get("NPIX", "256"))
def t(f, k=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k
for (nt, no, nc, H) in ((1, 1, 1, 16), (2, 1, 1, 32), (1, 2, 1, 16), (2, 2, 2, 32)):
    gen = torch.Generator().manual_seed(0)
    rays = pinhole_rays(n, n, enc_dim=H, gen=gen).to(dev)
    rays.encoding.requires_grad_(True)
    sizes = grid_sizes_for((1, 64, 64, 64, C), True)
    flat, _ = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])
    flat.requires_grad_(True)
    d = random_decoder(gen, nt, no, nc, C, H, 3, std=0.1)
    params = d.mlp_params.to(dev).requires_grad_(True)
    dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
    res = {"layers trunk/opacity/colour": f"{nt}/{no}/{nc}", "hidden": H}
    for kn, kern in (("mfma", _lib.LP_KERNEL_AUTO), ("generic", _lib.LP_KERNEL_GENERIC)):
        def fb():
            flat.grad = params.grad = rays.encoding.grad = None
            o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes, kernel=kern)
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()
        res[f"{kn}_fwd_bwd_ms"] = round(t(fb, 3 if kn == "mfma" else 1), 2)
    print(json.dumps(res))
