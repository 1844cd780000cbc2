"""Forward-kernel timing at several ray counts (occupancy experiment)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0"); lp.config.check_inputs = False
def run(H, W, C, G, S):
    gen = torch.Generator().manual_seed(0)
    rays = pinhole_rays(H, W, enc_dim=32, gen=gen).to(dev)
    sizes = grid_sizes_for((1, G, G, G, C), True)
    flat, _ = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])
    d = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.15)
    dec = lp.DecoderParams(d.mlp_params.to(dev), d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
    f = lambda: lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
    with torch.no_grad():
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    return round(ms, 3), round(H * W / ms / 1e3, 2)
print(json.dumps({"variant": os.environ.get("LP_MFMA_FWD_VARIANT", "0"),
                  "256x256 C16 S128 (ms, Mrays/s)": run(256, 256, 16, 64, 128),
                  "512x512 C16 S128": run(512, 512, 16, 64, 128),
                  "1080p C32 S256": run(1080, 1920, 32, 128, 256)}))
