"""HIP-graph capture probe (torch.cuda.CUDAGraph): forward only, then forward+backward; prints where it fails."""
import faulthandler, os, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
dev = torch.device("cuda:0")
lp.config.check_inputs = False
gen = torch.Generator().manual_seed(0)
C, S = 16, int(os.environ.get("S", "64"))
rays = pinhole_rays(64, 64, enc_dim=32, gen=gen).to(dev)
sizes = grid_sizes_for((1, 64, 64, 64, C), True)
flat = lp.flatten_grid([g.to(dev) for g in random_grids(gen, sizes)])[0]
d = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.1)
params = d.mlp_params.to(dev)
dec = lp.DecoderParams(params, d.n_hidden_trunk, d.n_hidden_opacity, d.n_hidden_color, 3)
mode = sys.argv[1] if len(sys.argv) > 1 else "fwd"
def fwd():
    return lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=1.0, grid_sizes=sizes)
if mode == "fwd":
    with torch.no_grad():
        ref = fwd()
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fwd()
        torch.cuda.current_stream().wait_stream(s)
        print("capturing forward", flush=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fwd()
        print("captured", flush=True)
        out[2].zero_(); g.replay(); torch.cuda.synchronize()
        print("replay vs eager:", float((out[2] - ref[2]).abs().max()), flush=True)
else:
    flat.requires_grad_(True); params.requires_grad_(True); rays.encoding.requires_grad_(True)
    def step():
        o = fwd()
        (o[0].sum() + o[1].sum() + o[2].sum()).backward()
    step(); ref = flat.grad.clone()
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            flat.grad = params.grad = rays.encoding.grad = None
            step()
    torch.cuda.current_stream().wait_stream(s)
    flat.grad = params.grad = rays.encoding.grad = None
    print("capturing forward+backward", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    print("captured", flush=True)
    flat.grad.zero_(); g.replay(); torch.cuda.synchronize()
    print("replay vs eager:", float((flat.grad - ref).abs().max() / ref.abs().max()), flush=True)
    def t(f, k=200):
        f(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
    def eager():
        flat.grad = params.grad = rays.encoding.grad = None
        step()
    print("eager %.1f us, graph replay %.1f us per fwd+bwd" % (t(eager), t(g.replay)), flush=True)
