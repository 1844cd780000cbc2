"""Print per-tensor relative errors (max|err| / max|ref|) of one parity case vs the golden file, both kernels."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lightplane_amd import _lib
from tests.synth import RENDERER_CASES
from tests.test_gpu_parity import run_hip_renderer, _rel_err
name = sys.argv[1]
case = next(c for c in RENDERER_CASES if c.name == name)
d = case.build()
z = np.load(os.path.join("tests", "golden", f"renderer__{name}.npz"))
dev = torch.device("cuda:0")
for kern, kn in ((_lib.LP_KERNEL_GENERIC, "generic"), (_lib.LP_KERNEL_AUTO, "auto")):
    out, gp, ge, gg, gc = run_hip_renderer(d, dev, kern)
    errs = {"len": _rel_err(out[0], z["ray_length"]), "nlt": _rel_err(out[1], z["neg_log_t"]), "feat": _rel_err(out[2], z["feature"]),
            "gparams": _rel_err(gp, z["grad_mlp_params"]), "genc": _rel_err(ge, z["grad_encoding"])}
    for i, g in enumerate(gg):
        errs[f"ggrid{i}"] = _rel_err(g, z[f"grad_grid{i}"])
    print(kn, {k: f"{v:.2e}" for k, v in errs.items()})
    # per-ray error of grad_encoding
    e = (ge.cpu().numpy() - z["grad_encoding"])
    bad = np.abs(e).max(axis=1)
    print("  worst rays genc:", np.argsort(-bad)[:4], bad[np.argsort(-bad)[:4]], "scale", np.abs(z["grad_encoding"]).max())
    print("  nlt of worst rays:", z["neg_log_t"][np.argsort(-bad)[:4]])
