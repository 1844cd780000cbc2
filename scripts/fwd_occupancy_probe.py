"""Forward kernel time at 1, 2 and 4 waves per SIMD worth of rays (cfg-2 decoder and grids): how much do the
co-resident waves of a SIMD overlap?  32768 rays = one wave per SIMD, 65536 = two (the benchmark), 131072 = four."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0"); lp.config.check_inputs = False
C = 16; S = 128
gen = torch.Generator().manual_seed(0)
sizes = grid_sizes_for((1, 64, 64, 64, C), True)
flat, _ = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])
d = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.1)
dec = lp.DecoderParams(d.mlp_params.to(dev), d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
res = {}
for h in (128, 256, 512):
    rays = pinhole_rays(h, 256, enc_dim=32, gen=gen).to(dev)
    def f():
        with torch.no_grad():
            lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    res[f"{h * 256} rays"] = round(e0.elapsed_time(e1) / 20, 4)
print(json.dumps(res))
