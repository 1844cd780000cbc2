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


# ---- Module level: the reference's LightplaneRenderer / LightplaneMLPSplatter modules on CPU with
# use_naive_impl=True (renderer_module.py:419-563, splatter_module.py:164-331): harmonic ray embedding + Linear ->
# render -> background / alpha, outputs and the gradients of every parameter.
from tests.synth import MODULE_RENDERER_CASES, module_renderer_inputs  # noqa: E402


def make_module_renderer(name):
    spec = MODULE_RENDERER_CASES[name]
    sizes, grids, rays, up, pgen = module_renderer_inputs(spec)
    mod = ref.LightplaneRenderer(use_naive_impl=True, **spec["ctor"])
    with torch.no_grad():
        mod.mlp_params.copy_(torch.randn(mod.mlp_params.shape, generator=pgen) * 0.2)
        mod.harmonic_ray_embedding_linear.weight.copy_(torch.randn(mod.harmonic_ray_embedding_linear.weight.shape, generator=pgen) * 0.3)
        mod.harmonic_ray_embedding_linear.bias.copy_(torch.randn(mod.harmonic_ray_embedding_linear.bias.shape, generator=pgen) * 0.1)
    gs = [g.clone().requires_grad_(True) for g in grids]
    r = ref.Rays(directions=rays.directions, origins=rays.origins, grid_idx=rays.grid_idx, near=rays.near, far=rays.far)
    out = mod(r, gs)
    (sum((o * u).sum() for o, u in zip(out, up))).backward()
    rec = {f"state__{k}": _np(v) for k, v in mod.state_dict().items()}
    rec.update(ray_length=_np(out[0]), alpha=_np(out[1]), feature=_np(out[2]),
               grad_mlp_params=_np(mod.mlp_params.grad),
               grad_linear_weight=_np(mod.harmonic_ray_embedding_linear.weight.grad),
               grad_linear_bias=_np(mod.harmonic_ray_embedding_linear.bias.grad))
    for i, g in enumerate(gs):
        rec[f"grad_grid{i}"] = _np(g.grad)
    return rec


def make_module_splatter():
    """LightplaneSplatter module (splatter_module.py:25-161).  (The reference's LightplaneMLPSplatter cannot run with
    use_naive_impl=True: its forward hands mlp_params to lightplane_splatter_naive, splatter_module.py:316-329, which
    has no such argument -- the MLP variant is pinned through the functional goldens instead.)"""
    from tests.synth import grid_sizes_for, random_rays
    gen = torch.Generator().manual_seed(41)
    mod = ref.LightplaneSplatter(num_samples=9, grid_chn=16, mask_out_of_bounds_samples=True, use_naive_impl=True)
    rays = random_rays(gen, 40, 2, 16)
    enc = torch.rand(40, 16, generator=gen).requires_grad_(True)
    r = ref.Rays(directions=rays.directions, origins=rays.origins, grid_idx=rays.grid_idx, near=rays.near, far=rays.far,
                 encoding=enc)
    out_sizes = grid_sizes_for((2, 6, 5, 7, 16), True)
    out = mod(r, out_sizes)
    up = [torch.randn(*s, generator=gen) for s in out_sizes]
    sum((o * u).sum() for o, u in zip(out, up)).backward()
    rec = dict(grad_encoding=_np(enc.grad))
    for i, o in enumerate(out):
        rec[f"out{i}"] = _np(o)
    return rec


def make_baseline_cfg1():
    """BASELINE.json configs[0] exactly (tests/synth.py baseline_cfg1): outputs and gradients only -- the inputs come from
    the seeded builder; a few input checksums make a generator drift visible."""
    from tests.synth import baseline_cfg1

    class _Case:
        def build(self):
            return baseline_cfg1()

    rec = make_renderer(_Case())
    keep = {k: v for k, v in rec.items() if k in ("ray_length", "neg_log_t", "feature", "grad_mlp_params", "grad_encoding",
                                                  "grad_grid0")}
    keep["checksum_inputs"] = np.array([rec["grid0"].astype(np.float64).sum(), rec["directions"].astype(np.float64).sum(),
                                        rec["encoding"].astype(np.float64).sum(), rec["mlp_params"].astype(np.float64).sum()])
    return keep


def make_randn():
    rec = {}
    for seed in (0, 5, 123456):
        x1 = torch.arange(1, 4097) * 7919
        x2 = x1 + 104729
        rec[f"z_seed{seed}"] = _np(int_to_randn_naive(x1, x2, seed))
    rec["x1"] = _np(x1)
    rec["x2"] = _np(x2)
    return rec


def make_modules():
    for name in MODULE_RENDERER_CASES:
        np.savez_compressed(os.path.join(HERE, f"module_renderer__{name}.npz"), **make_module_renderer(name))
        print("module renderer", name)
    np.savez_compressed(os.path.join(HERE, "module_splatter.npz"), **make_module_splatter())
    print("module splatter")


def main():
    torch.manual_seed(0)
    only = set(sys.argv[1:])  # optional: case names to (re)generate; default = all
    if only == {"modules"}:
        make_modules()
        return
    if only == {"baseline_cfg1"}:
        np.savez_compressed(os.path.join(HERE, "renderer__baseline_cfg1.npz"), **make_baseline_cfg1())
        print("renderer baseline_cfg1")
        return
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
    np.savez_compressed(os.path.join(HERE, "renderer__baseline_cfg1.npz"), **make_baseline_cfg1())
    make_modules()
    print("done")


if __name__ == "__main__":
    main()
