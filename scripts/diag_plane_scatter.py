"""Diagnostic for the coherent-ray grad_grid failure of the triplane MFMA backward (round 2).
Finds a minimal failing (wave, sample) and prints the per-ray cell pattern of the failing plane."""
import copy
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lightplane_amd as lp
from lightplane_amd import _lib
from oracle import lightplane_oracle as O
from tests.test_gpu_coherent import coherent_renderer_inputs
from tests.test_gpu_parity import run_hip_renderer, run_oracle_renderer, _rel_err

dev = torch.device("cuda:0")
which = sys.argv[1] if len(sys.argv) > 1 else "full"
d = coherent_renderer_inputs("triplane24_c16", "48x80_az30_el45")
print("env", {k: v for k, v in os.environ.items() if k.startswith("LP_")})


def errs(dd, kernel=_lib.LP_KERNEL_AUTO):
    out, gp, ge, gg, gc = run_hip_renderer(dd, dev, kernel)
    o_out, o_gp, o_ge, o_gg, o_gc = run_oracle_renderer(dd)
    res = {}
    for i, (a, b) in enumerate(zip(gg, o_gg)):
        a = a.cpu()
        diff = (a - b).abs()
        scale = b.abs().max().item() + 1e-30
        res[f"g{i}"] = (diff.max().item() / scale, int((diff > 1e-4 * scale).sum()), a.sum().item(), b.sum().item())
    res["gp"] = _rel_err(gp, o_gp.numpy())
    res["ge"] = _rel_err(ge, o_ge.numpy())
    return res, gg, o_gg


if which == "full":
    for k, nm in ((_lib.LP_KERNEL_AUTO, "auto"), (_lib.LP_KERNEL_GENERIC, "generic")):
        r, gg, o_gg = errs(d, k)
        print(nm, {k2: (v if not isinstance(v, tuple) else tuple(f"{x:.4g}" for x in v)) for k2, v in r.items()})
    for mask in (False,):
        d2 = coherent_renderer_inputs("triplane24_c16", "48x80_az30_el45", mask_oob=mask)
        r, _, _ = errs(d2)
        print("mask", mask, {k2: (v if not isinstance(v, tuple) else tuple(f"{x:.4g}" for x in v)) for k2, v in r.items()})
    sys.exit(0)

# ---- per-wave search: 32 consecutive rays at a time, all samples -------------------------------------------
rays = d["rays"]
n = rays.n_rays
bad = []
for w in range(n // 32):
    sl = slice(32 * w, 32 * w + 32)
    dd = dict(d, rays=rays[sl], upstream=tuple(u[sl] for u in d["upstream"]))
    r, gg, o_gg = errs(dd)
    worst = max(r["g0"][0], r["g1"][0], r["g2"][0])
    if worst > 1e-4:
        bad.append((w, r))
print(f"{len(bad)} of {n // 32} single-wave launches fail; first: {bad[:3]}")
if not bad:
    sys.exit(0)
w = bad[0][0]
sl = slice(32 * w, 32 * w + 32)
S = d["cfg"]["num_samples"]
# ---- per-sample search on that wave: near = far = depth of sample s, one sample --------------------------
rw = rays[sl]
depths = O.ray_depths(rw.near, rw.far, S, 0, 1e-5)
for s in range(S):
    r1 = copy.copy(rw)
    r1.near = depths[:, s].clone()
    r1.far = depths[:, s].clone()
    dd = dict(d, rays=r1, upstream=tuple(u[sl] for u in d["upstream"]), cfg=dict(d["cfg"], num_samples=1))
    r, gg, o_gg = errs(dd)
    worst = max(r["g0"][0], r["g1"][0], r["g2"][0])
    if worst > 1e-4:
        print(f"wave {w} sample {s}: ", {k2: (v if not isinstance(v, tuple) else tuple(f'{x:.4g}' for x in v)) for k2, v in r.items()})
        pts = depths[:, s, None] * rw.directions + rw.origins
        inb = (pts.abs() <= 1).all(-1)
        for gi, (ax_u, ax_v) in enumerate(((0, 1), (0, 2), (1, 2))):
            U = 24
            tu = ((pts[:, ax_u] + 1) * U - 1) / 2
            tv = ((pts[:, ax_v] + 1) * U - 1) / 2
            iu, iv = torch.floor(tu).long(), torch.floor(tv).long()
            print(f" plane {gi}: iu {iu.tolist()}\n          iv {iv.tolist()}\n          inb {inb.long().tolist()}")
        g = gg[0].cpu().reshape(24, 24, 16)
        o = o_gg[0].reshape(24, 24, 16)
        diff = (g - o).abs().sum(-1)
        idx = torch.nonzero(diff > 1e-4 * o.abs().max())
        print(" plane0 wrong cells (iv, iu):", idx.tolist()[:40])
        for (a, b) in idx.tolist()[:6]:
            print("   cell", a, b, "hip", g[a, b, :4].tolist(), "oracle", o[a, b, :4].tolist())
        break
