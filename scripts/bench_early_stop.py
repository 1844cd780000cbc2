"""Early ray termination (extension) on the cfg-2 workload with a dense medium: fwd+bwd time with and without
stop_transmittance.  gain = GAIN x the benchmark's (default 60: -log T reaches 11.5 after about a third of the march)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lightplane_amd as lp
from bench import RendererWorkload
from lightplane_amd import _lib

dev = torch.device("cuda:0")
lp.config.check_inputs = False
gain = float(os.environ.get("GAIN", "60"))
wl = RendererWorkload("cfg2", 0, dev, None, _lib.LP_KERNEL_AUTO)
rays, flat, params, dec, up, sizes, S = wl.rays, wl.flat, wl.params, wl.dec, wl.up, wl.sizes, wl.S

def step(stop):
    flat.grad = params.grad = rays.encoding.grad = None
    o = lp.lightplane_renderer(rays, flat, dec, num_samples=S, gain=gain, grid_sizes=sizes, stop_transmittance=stop)
    ((o[0] * up[0]).sum() + (o[2] * up[2]).sum()).backward()
    return o

def timeit(stop, k=10):
    step(stop); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): step(stop)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k

o0, o1 = step(0.0), step(1e-5)
res = {"gain": gain, "exact_ms": round(timeit(0.0), 3), "stop1e-5_ms": round(timeit(1e-5), 3),
       "max_abs_feature_diff": float((o0[2] - o1[2]).detach().abs().max()),
       "mean_neg_log_t_exact": float(o0[1].mean()), "mean_neg_log_t_stop": float(o1[1].mean())}
print(json.dumps(res))
