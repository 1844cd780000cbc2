"""Per-tensor errors of EVERY Renderer sweep case of tests/test_gpu_sweep.py (random sweep, segmented sweep, the reference's own
axes) against the fp32 AND the fp64 oracle, and the fp32 oracle's own error against fp64: the data behind the sweep's bars.
    python scripts/sweep_errors_all.py > gpurun_out/sweep_errors.json"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import warnings
warnings.filterwarnings("ignore")
import torch
from lightplane_amd import _lib
import lightplane_amd as lp
lp.config.warn_generic_kernel = False
from tests.test_gpu_sweep import _renderer_case, _segmented_case, _reference_axes_case, run_oracle_renderer64
from tests.test_gpu_parity import run_hip_renderer, run_oracle_renderer
dev = torch.device("cuda:0")


def rel(a, b):
    a = a.detach().double().cpu(); b = b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-6))


def tensors(r):
    out = {"ray_length": r[0][0], "neg_log_t": r[0][1], "feature": r[0][2], "grad_mlp_params": r[1], "grad_encoding": r[2]}
    for k, g in enumerate(r[3]):
        out[f"grad_grid{k}"] = g
    if r[4] is not None:
        for k, g in enumerate(r[4]):
            out[f"grad_color_grid{k}"] = g
    return out


rows = []
for fam, make, n in (("sweep", _renderer_case, 48), ("segsweep", _segmented_case, 16), ("refsweep", _reference_axes_case, 40)):
    for i in range(n):
        case = make(i)
        d = case.build()
        o64, o32, got = tensors(run_oracle_renderer64(d)), tensors(run_oracle_renderer(d)), tensors(run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO))
        for nm in got:
            e64, e32, eo = rel(got[nm], o64[nm]), rel(got[nm], o32[nm]), rel(o32[nm], o64[nm])
            rows.append(dict(case=case.name, inf=case.num_samples_inf, tensor=nm, e64=e64, e32=e32, oracle32_vs_64=eo))
bad = [r for r in rows if min(r["e64"], r["e32"]) > 1e-4]
print(json.dumps(dict(n=len(rows), n_above_1e4_vs_both=len(bad), above=bad,
                      worst_min=max(min(r["e64"], r["e32"]) for r in rows),
                      n_e64_above_1e4=sum(r["e64"] > 1e-4 for r in rows), n_e32_above_1e4=sum(r["e32"] > 1e-4 for r in rows)), indent=1))
