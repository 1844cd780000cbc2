"""Errors of one case of tests/test_gpu_sweep.py against the fp64 oracle: fp32 oracle, shape-generic kernels, auto kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightplane_amd import _lib
import lightplane_amd as lp
lp.config.warn_generic_kernel = False
from tests.test_gpu_sweep import _renderer_case, run_oracle_renderer64
from tests.test_gpu_parity import run_hip_renderer, run_oracle_renderer
dev = torch.device("cuda:0")
def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))
for i in [int(v) for v in sys.argv[1:]]:
    d = _renderer_case(i).build()
    o64 = run_oracle_renderer64(d)
    res = {"oracle32": run_oracle_renderer(d), "generic": run_hip_renderer(d, dev, _lib.LP_KERNEL_GENERIC),
           "auto": run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)}
    for k, r in res.items():
        e = {"out": max(rel(a, b) for a, b in zip(r[0], o64[0])), "gp": rel(r[1], o64[1]), "ge": rel(r[2], o64[2]),
             "gg": max(rel(a, b) for a, b in zip(r[3], o64[3]))}
        if r[4] is not None:
            e["gc"] = max(rel(a, b) for a, b in zip(r[4], o64[4]))
        print(i, k, {n: f"{v:.1e}" for n, v in e.items()})
