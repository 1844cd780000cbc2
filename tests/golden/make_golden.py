#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container (needs /root/reference; CPU only):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

For every seeded case in tests/synth.py this imports the reference's pure-PyTorch
implementations (`lightplane_renderer_naive`, `lightplane_splatter_naive`,
`lightplane_mlp_splatter_naive`, `int_to_randn_naive`) unmodified from
/root/reference, runs forward + backward on CPU and stores inputs, outputs and
gradients as ``<kind>__<case>.npz``.  The GPU box has no /root/reference, so these
files are what pins the oracle (tests/test_oracle_golden.py) and, through it, the
HIP kernels.  No reference source is copied: only numbers it produced.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

# the reference imports `cogapp` (a code generator for its Triton templates) at import
# time; it is not installed and not needed for the naive path -> stub it.
_stub = types.ModuleType("cogapp")
_stub.Cog = object
sys.modules["cogapp"] = _stub
sys.path.append("/root/reference")  # after REPO: the reference also has a `tests` package

import lightplane as ref  # noqa: E402
from lightplane.triton_src.shared.rand_util import int_to_randn_naive  # noqa: E402

from tests.synth import RENDERER_CASES, SPLATTER_CASES  # noqa: E402


def _np(t):
    return t.detach().cpu().numpy()


def make_renderer(case):
    d = case.build()
    rays = d["rays"]
    enc = rays.encoding.clone().requires_grad_(True)
    r = ref.Rays(directions=rays.directions, origins=rays.origins, grid_idx=rays.grid_idx,
                 near=rays.near, far=rays.far, encoding=enc)
    grids = [g.clone().requires_grad_(True) for g in d["grids"]]
    cgrids = None if d["color_grids"] is None else [g.clone().requires_grad_(True) for g in d["color_grids"]]
    dec = d["decoder"]
    params = dec.mlp_params.clone().requires_grad_(True)
    rdec = ref.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
    out = ref.lightplane_renderer_naive(r, grids, rdec, scaffold=d["scaffold"], color_grid=cgrids, **d["cfg"])
    g_len, g_nlt, g_feat = d["upstream"]
    loss = (out[0] * g_len).sum() + (out[1] * g_nlt).sum() + (out[2] * g_feat).sum()
    loss.backward()
    rec = dict(
        directions=_np(rays.directions), origins=_np(rays.origins), grid_idx=_np(rays.grid_idx),
        near=_np(rays.near), far=_np(rays.far), encoding=_np(rays.encoding), mlp_params=_np(dec.mlp_params),
        ray_length=_np(out[0]), neg_log_t=_np(out[1]), feature=_np(out[2]),
        grad_mlp_params=_np(params.grad), grad_encoding=_np(enc.grad),
    )
    for i, g in enumerate(grids):
        rec[f"grid{i}"] = _np(g)
        rec[f"grad_grid{i}"] = _np(g.grad)
    if cgrids is not None:
        for i, g in enumerate(cgrids):
            rec[f"cgrid{i}"] = _np(g)
            rec[f"grad_cgrid{i}"] = _np(g.grad)
    if d["scaffold"] is not None:
        rec["scaffold"] = _np(d["scaffold"])
    return rec


def make_splatter(case):
    d = case.build()
    rays = d["rays"]
    enc = rays.encoding.clone().requires_grad_(True)
    r = ref.Rays(directions=rays.directions, origins=rays.origins, grid_idx=rays.grid_idx,
                 near=rays.near, far=rays.far, encoding=enc)
    rec = dict(directions=_np(rays.directions), origins=_np(rays.origins), grid_idx=_np(rays.grid_idx),
               near=_np(rays.near), far=_np(rays.far), encoding=_np(rays.encoding))
    if d["mlp"] is None:
        out = ref.lightplane_splatter_naive(r, d["out_sizes"], **d["cfg"])
        params = in_grids = None
    else:
        params = d["mlp"].mlp_params.clone().requires_grad_(True)
        smlp = ref.SplatterParams(params, d["mlp"].n_hidden)
        in_grids = [g.clone().requires_grad_(True) for g in d["in_grids"]]
        out = ref.lightplane_mlp_splatter_naive(r, d["out_sizes"], smlp, in_grids, **d["cfg"])
        rec["mlp_params"] = _np(d["mlp"].mlp_params)
    loss = sum((o * u).sum() for o, u in zip(out, d["upstream"]))
    loss.backward()
    for i, o in enumerate(out):
        rec[f"out{i}"] = _np(o)
    rec["grad_encoding"] = _np(enc.grad)
    if params is not None:
        rec["grad_mlp_params"] = _np(params.grad)
        for i, g in enumerate(in_grids):
            rec[f"in_grid{i}"] = _np(g)
            rec[f"grad_in_grid{i}"] = _np(g.grad)
    return rec


def make_randn():
    rec = {}
    for seed in (0, 5, 123456):
        x1 = torch.arange(1, 4097) * 7919
        x2 = x1 + 104729
        rec[f"z_seed{seed}"] = _np(int_to_randn_naive(x1, x2, seed))
    rec["x1"] = _np(x1)
    rec["x2"] = _np(x2)
    return rec


def main():
    torch.manual_seed(0)
    only = set(sys.argv[1:])  # optional: case names to (re)generate; default = all
    if only:
        for case in RENDERER_CASES:
            if case.name in only:
                np.savez_compressed(os.path.join(HERE, f"renderer__{case.name}.npz"), **make_renderer(case))
                print("renderer", case.name)
        for case in SPLATTER_CASES:
            if case.name in only:
                np.savez_compressed(os.path.join(HERE, f"splatter__{case.name}.npz"), **make_splatter(case))
                print("splatter", case.name)
        return
    for case in RENDERER_CASES:
        rec = make_renderer(case)
        np.savez_compressed(os.path.join(HERE, f"renderer__{case.name}.npz"), **rec)
        print("renderer", case.name, {k: v.shape for k, v in rec.items() if k in ("feature", "grad_mlp_params")})
    for case in SPLATTER_CASES:
        rec = make_splatter(case)
        np.savez_compressed(os.path.join(HERE, f"splatter__{case.name}.npz"), **rec)
        print("splatter", case.name, rec["out0"].shape)
    np.savez_compressed(os.path.join(HERE, "randn.npz"), **make_randn())
    print("done")


if __name__ == "__main__":
    main()
