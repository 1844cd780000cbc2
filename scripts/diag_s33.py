"""Which pairs of {bf16x3 kernel, generic kernel, fp32 oracle, fp64 oracle} agree on grad_grid of the S=33 segmented case."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lightplane_amd as lp
from lightplane_amd import _lib
from tests.test_gpu_coherent import coherent_renderer_inputs, oracle_renderer64
from tests.test_gpu_parity import run_hip_renderer, run_oracle_renderer
dev = torch.device("cuda:0")
for S, seed in ((33, 11), (33, 3), (40, 11), (32, 11)):
    d = coherent_renderer_inputs("triplane24_c16", "64x64_axis", num_samples=S, seed=seed)
    res = {}
    res["bf3"] = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    res["generic"] = run_hip_renderer(d, dev, _lib.LP_KERNEL_GENERIC)
    res["o32"] = run_oracle_renderer(d)
    res["o64"] = oracle_renderer64(d)
    def gg(r, i): return r[3][i].detach().double().cpu().numpy()
    names = list(res)
    for i in range(3):
        print(f"S={S} seed={seed} grad_grid{i}:")
        for a in range(len(names)):
            for b in range(a + 1, len(names)):
                x, y = gg(res[names[a]], i), gg(res[names[b]], i)
                sc = np.abs(y).max()
                e = np.abs(x - y) / sc
                print(f"   {names[a]:8s} vs {names[b]:8s}: max {e.max():.2e}  n>1e-4 {int((e > 1e-4).sum()):4d}  relL2 {np.linalg.norm(x - y) / np.linalg.norm(y):.2e}")
