"""GPU parity tests: HIP kernels (through the C ABI) vs the CPU oracle and the golden fixtures.

Bar (BASELINE.json north_star): outputs and gradients within 1e-4 relative (max |err| /
max |ref| per tensor) of the naive reference; integer corner indices bit-exact.
"""
import os

import numpy as np
import pytest
import torch

import lightplane_amd as lp
from lightplane_amd import _lib
from oracle import lightplane_oracle as O
from tests.synth import RENDERER_CASES, SPLATTER_CASES, pinhole_rays, random_decoder, random_grids, grid_sizes_for

pytestmark = pytest.mark.gpu

REL_TOL = 1e-4
KERNELS = [_lib.LP_KERNEL_GENERIC, _lib.LP_KERNEL_AUTO]
KERNEL_IDS = ["generic", "auto"]


def _dev():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch.device("cuda:0")


def _rel_err(got, want):
    got = got.detach().double().cpu()
    want = torch.as_tensor(np.asarray(want)).double()
    assert got.shape == want.shape, f"shape {tuple(got.shape)} vs {tuple(want.shape)}"
    scale = max(want.abs().max().item(), 1e-6)
    return (got - want).abs().max().item() / scale


def _assert_close(name, got, want, tol=REL_TOL):
    e = _rel_err(got, want)
    assert e <= tol, f"{name}: max err / scale = {e:.3e} > {tol}"


FLIP_SAMPLES = 4
FLIP_EVENTS = []  # one record per use of the flip allowance; printed by conftest.pytest_terminal_summary


def rel_l2(got, want):
    got = got.detach().double().cpu()
    want = torch.as_tensor(np.asarray(want)).double()
    return (got - want).norm().item() / max(want.norm().item(), 1e-30)


UNEXPLAINED_MAX = 5e-3  # without a tie mask: an entry that misses the bar against BOTH oracles may not be further off than this
TIE_EPS = 1e-6          # a ReLU pre-activation below this fraction of its layer's largest one counts as a near tie (5e-6 until
                        # round 4; the CPU check test_near_tie_masks_contain_every_fp32_vs_fp64_relu_flip still holds at 5e-7)


class TieMasks:
    """WHERE may two correct fp32 evaluations of the gradient differ?  Only where a ReLU pre-activation is within round-off of
    zero: the forward value is continuous there, the gradient is not, and which branch an implementation takes depends on
    its summation order.  A flipped unit of sample (r, s) changes the gradient of the grid rows that sample's taps touch, of
    ray r's encoding and of rows / columns of the weight matrices -- nothing else.  The fp64 oracle's forward is run with
    oracle.relu_margin_recorder; samples whose smallest relative |pre-activation| is below TIE_EPS (~the round-off of an
    fp32 dot product of this size) are near ties, and their tap rows (oracle.renderer_corner_indices in fp32 -- the index
    arithmetic the kernels reproduce bit for bit) form the mask.  Computed lazily, once per test, only when some tensor misses
    the bar."""

    def __init__(self, d, idx=None, chunk=2048):
        self.d, self.idx, self.chunk, self._done = d, idx, chunk, False

    def _compute(self):
        if self._done:
            return
        import copy
        d = self.d
        rays = d["rays"] if self.idx is None else d["rays"][self.idx]
        cfg = d["cfg"]
        F64 = torch.float64
        dec = copy.copy(d["decoder"])
        dec.mlp_params = dec.mlp_params.detach().to(F64)
        grids = [g.detach().to(F64) for g in d["grids"]]
        cgrids = None if d.get("color_grids") is None else [g.detach().to(F64) for g in d["color_grids"]]
        scaffold = None if d.get("scaffold") is None else d["scaffold"].to(F64)
        margins = []
        old_threads = torch.get_num_threads()
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        try:
            with torch.no_grad():
                for lo in range(0, rays.n_rays, self.chunk):
                    r = rays[lo:lo + self.chunk]
                    for f in ("directions", "origins", "near", "far", "encoding"):
                        setattr(r, f, getattr(r, f).detach().to(F64))
                    with O.relu_margin_recorder() as rec:
                        O.lightplane_renderer_naive(r, grids, dec, scaffold=scaffold, color_grid=cgrids, **cfg)
                    margins.append(rec.margin)
        finally:
            torch.set_num_threads(old_threads)
        near = torch.cat(margins) < TIE_EPS                      # [R, S_tot]
        self.n_near = int(near.sum())
        self.ray = near.any(dim=1)
        r32 = rays

        def row_masks(tensors):
            sizes = [list(g.shape) for g in tensors]
            rows = O.renderer_corner_indices(r32, sizes, cfg["num_samples"], cfg.get("num_samples_inf", 0), cfg.get("contract_coords", False))
            out = []
            for g, rr in zip(tensors, rows):
                m = torch.zeros(g.numel() // g.shape[-1], dtype=torch.bool)
                hit = rr[near].reshape(-1)
                m[hit[hit >= 0]] = True
                out.append(m.reshape(g.shape[:-1] + (1,)))
            return out

        self.grid = row_masks(d["grids"])
        self.color_grid = None if d.get("color_grids") is None else row_masks(d["color_grids"])
        self._done = True

    class _Mask:
        """Lazy mask + the measured number of near-tie SAMPLES behind it (assert_grad_close bounds the allowance by it)."""

        def __init__(self, owner, get):
            self.owner, self.get = owner, get

        def __call__(self):
            self.owner._compute()
            return self.get()

        def n_near(self):
            self.owner._compute()
            return self.owner.n_near

    def grid_mask(self, i):
        return TieMasks._Mask(self, lambda: self.grid[i])

    def color_grid_mask(self, i):
        return TieMasks._Mask(self, lambda: self.color_grid[i])

    def encoding_mask(self):
        return TieMasks._Mask(self, lambda: self.ray[:, None])

    def params_mask(self):  # a flipped unit moves a row / a column of weight matrices and bias entries: any entry may move
        return TieMasks._Mask(self, lambda: torch.tensor(self.n_near > 0))


def assert_grad_close(name, got, want, entries_per_sample, tol=1e-4, want64=None, flip_samples=FLIP_SAMPLES, tie_mask=None):
    """Gradient tensor vs the fp32 oracle: the north_star bar, or the ReLU-flip allowance described in
    tests/test_gpu_coherent.py's docstring.  Every use of the allowance is recorded and printed.

    ``want64`` (the same oracle run in fp64; an array or a zero-argument callable that is only evaluated when the bar is
    missed): an entry that misses the bar against the fp32 oracle but meets it against the fp64 one is EXPLAINED -- the
    fp32 oracle took the other branch of a ReLU there, the kernel the exact one -- and does not count.  The entries that
    miss BOTH oracles (the kernel's own summation order took a branch neither oracle took) count against ``flip_samples``
    samples' worth of entries, and
    * with ``tie_mask`` (TieMasks: a callable returning a boolean tensor broadcastable to the gradient): every one of them
      has to lie where a near-tie ReLU can reach -- an entry outside the mask that misses both oracles FAILS, whatever
      its size; inside the mask the size of a miss is the flipped unit's whole contribution, so only loose magnitude bars
      apply (0.5 of the largest entry, relative L2 5e-2) next to the count;
    * without one (or when the mask covers more than half of the tensor, as it does at config scale), none may be off by more
      than UNEXPLAINED_MAX of the largest entry.
    The allowance is NOT available without the second oracle (round-3 review, weak 1)."""
    want = torch.as_tensor(np.asarray(want))
    g = got.detach().double().cpu()
    w = want.double()
    assert g.shape == w.shape, f"{name}: shape {tuple(g.shape)} vs {tuple(w.shape)}"
    scale = max(w.abs().max().item(), 1e-30)
    err = (g - w).abs() / scale
    l2 = rel_l2(got, want)
    worst = err.max().item() if err.numel() else 0.0
    if worst <= tol:
        assert l2 <= tol, f"{name}: relative L2 error {l2:.3e} > {tol} (max-norm {worst:.3e})"
        return
    assert want64 is not None, (f"{name}: max err / scale = {worst:.3e} > {tol} (relative L2 {l2:.3e}) and no second (fp64) oracle "
                                f"was given -- the ReLU-flip allowance needs one")
    if callable(want64):
        want64 = want64()
    off = err > tol
    n_off = int(off.sum())
    w64 = torch.as_tensor(np.asarray(want64)).double()
    err64 = (g - w64).abs() / scale
    unexplained = off & (err64 > tol)
    n_un = int(unexplained.sum())
    both = torch.minimum(err, err64)
    worst_un = float(both[unexplained].max()) if n_un else 0.0
    n_outside, worst_outside, dense, cover, n_near = None, 0.0, False, None, None
    if tie_mask is not None and n_un:
        tm = torch.as_tensor(tie_mask()).expand_as(unexplained)
        outside = unexplained & ~tm
        n_outside = int(outside.sum())
        worst_outside = float(both[outside].max()) if n_outside else 0.0
        cover = float(tm.float().mean())
        dense = cover > 0.5  # (config-scale batches: near ties reach most rows, the mask says little)
        n_near = tie_mask.n_near() if hasattr(tie_mask, "n_near") else None
    # the allowance is the number of near-tie samples the fp64 oracle MEASURED (never more than flip_samples), not a constant
    allowed = (flip_samples if n_near is None else min(flip_samples, n_near)) * entries_per_sample
    FLIP_EVENTS.append(dict(name=name, tol=tol, n_off=n_off, explained=n_off - n_un, unexplained=n_un, allowed=allowed, worst=worst,
                            worst_unexplained=worst_un, outside_tie_mask=n_outside, l2=l2, near_tie_samples=n_near, mask_cover=cover))
    print(f"flip-allowance {name}: {n_off} entries above {tol:g}, {n_off - n_un} explained by the second oracle, "
          f"worst {worst:.3e}, worst unexplained {worst_un:.3e}, outside the near-tie mask: {n_outside}, rel L2 {l2:.3e}, "
          f"near-tie samples measured: {n_near}, mask covers {cover if cover is None else round(cover, 4)} of the tensor, allowed {allowed}")
    if tie_mask is not None and not dense:
        # every entry that misses both oracles is PROVEN to sit on a near-tie ReLU (none outside the mask): its size is then the
        # full contribution of the flipped unit, which nothing bounds relative to the tensor's largest entry (grad_encoding of
        # one ray can be dominated by one sample) -- the count stays bounded by the MEASURED near-tie count and the relative
        # L2 bar of the other branch stays; only the per-entry magnitude bar is loose.  (The tuned family does not need this
        # branch at all: forced_oracle_check below replaces it by a proof.)
        # (relative L2 5e-3, round 4: 5e-2.  One flipped unit of ONE ray moves that ray's 32 encoding-gradient entries; on a
        # 3 840-ray image that alone is a relative L2 of 2.5e-3 -- voxel16_c64/48x80 on the shape-generic kernel, mask cover
        # 1.8 % of the tensor, every miss inside it.  The tuned family does not come here: test_flips_are_flips proves it.)
        ok = n_un <= allowed and not n_outside and worst <= 0.5 and l2 <= 5e-3
    else:
        ok = n_un <= allowed and worst <= 5e-2 and l2 <= 1e-3 and worst_un <= UNEXPLAINED_MAX and not n_outside
    assert ok, (f"{name}: max err / scale = {worst:.3e} > {tol} and not a ReLU-flip pattern: {n_un} entries above the bar against "
                f"both oracles, the worst by {worst_un:.3e}"
                + (f"; {n_outside} of them where no near-tie ReLU reaches (worst {worst_outside:.3e})" if tie_mask is not None
                   else f" (allowed {UNEXPLAINED_MAX:g} without a tie mask)")
                + f"; {n_off - n_un} more explained by the fp64 oracle; allowed count {allowed}; relative L2 {l2:.3e} (allowed 1e-3)")


def has_dump_twin(d, kernel=_lib.LP_KERNEL_AUTO, **extra):
    """The kernel that runs this case has a DUMP twin (every family since round 6: tuned, layer-looped, shape-generic; not the tuned
    family's eight-wave workgroups, not LP_ARITH_FP32, not early termination -- the oracle marches every sample)."""
    if extra.get("stop_transmittance"):
        return False
    from lightplane_amd.renderer import relu_dump_words
    return relu_dump_words(d["rays"], d["grids"], d["decoder"], color_grid=d.get("color_grids"),
                           num_samples_inf=d["cfg"].get("num_samples_inf", 0), kernel=kernel) > 0


def relu_site_widths(d):
    """Widths of the decoder's ReLU sites in the oracle's call order (oracle.eval_decoder; naive_renderer.py:328-501): single
    grid-list -- every trunk layer, the opacity head's hidden layers, the colour head's; two-grid decoder -- relu(sampled feature),
    opacity hidden layers, relu(sampled colour feature), colour hidden layers."""
    dec = d["decoder"]
    t, o, c = ([int(v) for v in x] for x in (dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color))
    if d.get("color_grids") is not None:
        C = int(d["grids"][0].shape[-1])
        return [C] + o[1:-1] + [C] + c[1:-1]
    return t[1:] + o[1:-1] + c[1:-1]


def unpack_relu_dump(dump, widths, words_per_site=1):
    """int32 [R, S, W] (lp_renderer_backward_relu_dump: ``words_per_site`` words per ReLU site, then the flag word) -> (list of
    bool [R, S, width_k] masks in the oracle's ReLU call order, bool [R, S] "the kernel visited this sample")."""
    d = dump.cpu().to(torch.int64) & 0xFFFFFFFF
    assert d.shape[-1] == words_per_site * len(widths) + 1, (tuple(d.shape), widths, words_per_site)
    bits = torch.arange(32, dtype=torch.int64)
    masks = []
    for k, w in enumerate(widths):
        words = [((d[..., k * words_per_site + b, None] >> bits) & 1).bool() for b in range(words_per_site)]
        masks.append(torch.cat(words, dim=-1)[..., :w])
    return masks, d[..., -1] != 0


def run_hip_renderer_with_dump(d, dev, kernel=_lib.LP_KERNEL_AUTO, **extra):
    """The production backward AND the same backward through the DUMP twin of its kernel (ReLU decisions recorded).
    Returns (production results, dump, words per site)."""
    from lightplane_amd.renderer import relu_dump_recorder
    prod = run_hip_renderer(d, dev, kernel, **extra)
    with relu_dump_recorder() as rec:
        twin = run_hip_renderer(d, dev, kernel, **extra)
    assert rec.dump is not None, "the backward did not go through the dump hook"
    # the twin is the same template with stores added: same arithmetic, so its gradients equal the production launch's up to
    # the order of the fp32 atomics
    pairs = [("grad_mlp_params", prod[1], twin[1]), ("grad_encoding", prod[2], twin[2])] + \
        [(f"grad_grid{i}", a, b) for i, (a, b) in enumerate(zip(prod[3], twin[3]))]
    if prod[4] is not None:
        pairs += [(f"grad_color_grid{i}", a, b) for i, (a, b) in enumerate(zip(prod[4], twin[4]))]
    for nm, a, b in pairs:
        sc = float(a.abs().max()) + 1e-30
        e = float((a - b).abs().max()) / sc
        assert e <= 2e-5, f"dump twin vs production launch: {nm} differs by {e:.3e}"
    return prod, rec.dump, rec.words_per_site


FORCED_TIE_K = 1  # a unit the kernel decided against the fp64 oracle's sign has to be a near tie IN the oracle: |pre-activation| <=
                  # FORCED_TIE_K * TIE_EPS = 1e-6 of its site's largest one.  Measured over the suite's 160 proofs (oracle on the reference's
                  # fp32 geometry): at most 22 forced units per launch, largest margin 3.5e-8 -- the round-off of an fp32-equivalent dot
                  # product of 16-64 terms whose inputs carry the round-off of up to seven earlier layers (profiles/r06_forced_oracle.json)


def oracle_forced(d, dump, idx=None, chunk=2048, dtype=torch.float64, words_per_site=1):
    """fp64 oracle forward + backward over the rays ``idx`` (default all), in chunks, with the kernel's own ReLU decisions
    (``dump`` [n_rays_total, S, W] from the DUMP twin) forced onto every unit of every sample the kernel visited.
    Returns (outs, grad_params, grad_encoding, grad_grids, grad_color_grids, stats) -- stats: n_forced = units whose forced branch
    differs from the oracle's own, max_forced_margin = the largest relative |pre-activation| among them, n_near_units = units of the
    visited samples within FORCED_TIE_K * TIE_EPS of zero (the pool a forced unit has to come from), n_units."""
    import copy
    rays = d["rays"] if idx is None else d["rays"][idx]
    up = d["upstream"] if idx is None else tuple(u[idx] for u in d["upstream"])
    dump = dump if idx is None else dump[idx.to(dump.device)]
    widths = relu_site_widths(d)
    masks, visited = unpack_relu_dump(dump, widths, words_per_site)
    dec = d["decoder"]
    params = dec.mlp_params.to(dtype).clone().requires_grad_(True)
    grids = [g.to(dtype).clone().requires_grad_(True) for g in d["grids"]]
    cgrids = None if d.get("color_grids") is None else [g.to(dtype).clone().requires_grad_(True) for g in d["color_grids"]]
    scaffold = None if d.get("scaffold") is None else d["scaffold"].to(dtype)
    outs, g_enc = [[], [], []], []
    stats = dict(n_forced=0, max_forced_margin=0.0, n_near_units=0, n_units=0)
    old_threads = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    try:
        for lo in range(0, rays.n_rays, chunk):
            r = rays[lo:lo + chunk]
            for f in ("directions", "origins", "near", "far", "encoding"):
                setattr(r, f, getattr(r, f).to(dtype))
            r.encoding = r.encoding.clone().requires_grad_(True)
            dd = copy.copy(dec)
            dd.mlp_params = params
            keep = [visited[lo:lo + chunk, :, None].expand(-1, -1, w) for w in widths]
            # (geometry_dtype: the march's geometry -- cells, interpolation weights, interval lengths -- in the reference's fp32, as
            # the kernels compute it; the fp64 part is the decoder.  With fp64 geometry a coordinate's 2^-24 x grid-extent round-off
            # moves a pre-activation by up to 1e-5 of its site's largest on a 128-cell axis: the forced margins of the cfg-4 block
            # measured 7e-6 .. 9.5e-6 that way, 3 .. 8 x what the decoder's own arithmetic explains.)
            with O.geometry_dtype(torch.float32), O.relu_mask_forcer([m[lo:lo + chunk] for m in masks], keep, near_eps=FORCED_TIE_K * TIE_EPS) as forcer:
                out = O.lightplane_renderer_naive(r, grids, dd, scaffold=scaffold, color_grid=cgrids, **d["cfg"])
            assert forcer.k == len(widths), f"the oracle evaluated {forcer.k} ReLU sites, the dump holds {len(widths)}"
            stats["n_forced"] += forcer.n_forced
            stats["n_near_units"] += forcer.n_near_units
            stats["n_units"] += forcer.n_units
            stats["max_forced_margin"] = max(stats["max_forced_margin"], forcer.max_forced_margin)
            u = [x[lo:lo + chunk].to(dtype) for x in up]
            ((out[0] * u[0]).sum() + (out[1] * u[1]).sum() + (out[2] * u[2]).sum()).backward()
            for k in range(3):
                outs[k].append(out[k].detach())
            g_enc.append(r.encoding.grad)
    finally:
        torch.set_num_threads(old_threads)
    return ([torch.cat(o) for o in outs], params.grad, torch.cat(g_enc), [g.grad for g in grids],
            None if cgrids is None else [g.grad for g in cgrids], stats)


FORCED_EVENTS = []  # one record per forced-oracle check; printed by conftest.pytest_terminal_summary


def forced_oracle_check(name, d, dev, idx=None, tol=1e-4, chunk=2048, kernel=_lib.LP_KERNEL_AUTO, **extra):
    """THE PROOF behind the ReLU-flip allowance (round-4 review, next 2): the production backward's own ReLU decisions, read
    back from the DUMP twin of its kernel, are forced onto the fp64 oracle; then EVERY entry of every gradient family and every
    output has to meet the north_star bar outright -- no allowance, no second oracle, no mask.  Whatever separated the kernel
    from the unforced oracles was a ReLU branch taken the other way at a near tie, or this fails.

    The forcing itself is bounded (round-5 review, weak 1): a kernel with WRONG pre-activations would decide many units against the
    oracle and forcing them all would hide it.  So every forced unit has to be a near tie in the fp64 oracle -- its |pre-activation|
    at most FORCED_TIE_K * TIE_EPS of its site's largest one -- and there cannot be more forced units than the oracle has units
    that close to zero."""
    prod, dump, wps = run_hip_renderer_with_dump(d, dev, kernel, **extra)
    out, gp, ge, gg, gc = prod
    f_out, f_gp, f_ge, f_gg, f_gc, st = oracle_forced(d, dump, idx, chunk=chunk, words_per_site=wps)
    sel = (lambda t: t) if idx is None else (lambda t: t[idx.to(t.device)])
    worst = {}
    for nm, a, b in [("ray_length", sel(out[0]), f_out[0]), ("neg_log_t", sel(out[1]), f_out[1]), ("feature", sel(out[2]), f_out[2]),
                     ("grad_mlp_params", gp, f_gp), ("grad_encoding", sel(ge), f_ge)] + \
            [(f"grad_grid{i}", a, b) for i, (a, b) in enumerate(zip(gg, f_gg))] + \
            ([] if gc is None else [(f"grad_color_grid{i}", a, b) for i, (a, b) in enumerate(zip(gc, f_gc))]):
        worst[nm] = _rel_err(a, b.numpy())
    visited = int((dump[..., -1] != 0).sum()) if idx is None else int((dump[idx.to(dump.device)][..., -1] != 0).sum())
    n_forced = st["n_forced"]
    FORCED_EVENTS.append(dict(name=name, forced_units=n_forced, visited_samples=visited, max_forced_margin=float(f"{st['max_forced_margin']:.3e}"),
                              near_tie_units=st["n_near_units"], units=st["n_units"], worst={k: float(f"{v:.3e}") for k, v in worst.items()}))
    print(f"forced-oracle {name}: {n_forced} ReLU units forced against the fp64 oracle's own sign over {visited} visited samples "
          f"(largest forced |pre-activation| / site max {st['max_forced_margin']:.2e}; the oracle has {st['n_near_units']} of {st['n_units']} units "
          f"within {FORCED_TIE_K * TIE_EPS:g}); max err / scale: " + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
    assert st["max_forced_margin"] <= FORCED_TIE_K * TIE_EPS, (
        f"{name}: the kernel decided a ReLU unit against the fp64 oracle whose pre-activation is {st['max_forced_margin']:.3e} of its "
        f"site's largest -- not a near tie (bar {FORCED_TIE_K * TIE_EPS:g}): the kernel's pre-activations are off")
    assert n_forced <= st["n_near_units"], f"{name}: {n_forced} forced units but only {st['n_near_units']} near-tie units in the fp64 oracle"
    bad = {k: v for k, v in worst.items() if not v <= tol}
    assert not bad, f"{name}: with the kernel's own ReLU decisions forced onto the fp64 oracle these still miss {tol:g}: {bad}"
    return n_forced


def _rays_to(rays, dev, requires_grad=False):
    r = rays.to(dev)
    if requires_grad and r.encoding is not None:
        r.encoding = r.encoding.clone().requires_grad_(True)
    return r


def run_hip_renderer(d, dev, kernel, flat=False, **extra):
    rays = _rays_to(d["rays"], dev, True)
    dec = d["decoder"]
    params = dec.mlp_params.to(dev).clone().requires_grad_(True)
    hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
    grids = [g.to(dev).clone().requires_grad_(True) for g in d["grids"]]
    cgrids = None if d["color_grids"] is None else [g.to(dev).clone().requires_grad_(True) for g in d["color_grids"]]
    scaffold = None if d["scaffold"] is None else d["scaffold"].to(dev)
    out = lp.lightplane_renderer(rays, grids, hdec, scaffold=scaffold, color_grid=cgrids, kernel=kernel, **d["cfg"], **extra)
    g_len, g_nlt, g_feat = (t.to(dev) for t in d["upstream"])
    ((out[0] * g_len).sum() + (out[1] * g_nlt).sum() + (out[2] * g_feat).sum()).backward()
    return out, params.grad, rays.encoding.grad, [g.grad for g in grids], None if cgrids is None else [g.grad for g in cgrids]


def run_oracle_renderer(d):
    import copy
    rays = copy.copy(d["rays"])
    rays.encoding = rays.encoding.clone().requires_grad_(True)
    dec = copy.copy(d["decoder"])
    dec.mlp_params = dec.mlp_params.clone().requires_grad_(True)
    grids = [g.clone().requires_grad_(True) for g in d["grids"]]
    cgrids = None if d["color_grids"] is None else [g.clone().requires_grad_(True) for g in d["color_grids"]]
    out = O.lightplane_renderer_naive(rays, grids, dec, scaffold=d["scaffold"], color_grid=cgrids, **d["cfg"])
    g_len, g_nlt, g_feat = d["upstream"]
    ((out[0] * g_len).sum() + (out[1] * g_nlt).sum() + (out[2] * g_feat).sum()).backward()
    return out, dec.mlp_params.grad, rays.encoding.grad, [g.grad for g in grids], None if cgrids is None else [g.grad for g in cgrids]


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("case", RENDERER_CASES, ids=lambda c: c.name)
def test_renderer_matches_oracle_and_golden(case, kernel, golden_dir):
    dev = _dev()
    d = case.build()
    z = np.load(os.path.join(golden_dir, f"renderer__{case.name}.npz"))
    import warnings
    with warnings.catch_warnings():
        if lp.kernel_family(d["rays"], d["grids"], d["decoder"], color_grid=d["color_grids"]) == 0:
            # a shape no MFMA family covers (hidden 64 with more than 2 layers per MLP): the shape-generic kernels are what is
            # under test here, their "10-100x slower" warning is expected
            warnings.simplefilter("ignore", UserWarning)
        out, gp, ge, gg, gc = run_hip_renderer(d, dev, kernel)
    o_out, o_gp, o_ge, o_gg, o_gc = run_oracle_renderer(d)
    # vs golden (numbers the reference itself produced)
    _assert_close("ray_length/golden", out[0], z["ray_length"])
    _assert_close("neg_log_t/golden", out[1], z["neg_log_t"])
    _assert_close("feature/golden", out[2], z["feature"])
    _assert_close("grad_mlp_params/golden", gp, z["grad_mlp_params"])
    _assert_close("grad_encoding/golden", ge, z["grad_encoding"])
    for i, g in enumerate(gg):
        _assert_close(f"grad_grid{i}/golden", g, z[f"grad_grid{i}"])
    if gc is not None:
        for i, g in enumerate(gc):
            _assert_close(f"grad_cgrid{i}/golden", g, z[f"grad_cgrid{i}"])
    # vs oracle on the same seeded inputs
    for name, a, b in (("ray_length", out[0], o_out[0]), ("neg_log_t", out[1], o_out[1]), ("feature", out[2], o_out[2]),
                       ("grad_mlp_params", gp, o_gp), ("grad_encoding", ge, o_ge)):
        _assert_close(name + "/oracle", a, b.detach().numpy())
    for i, (a, b) in enumerate(zip(gg, o_gg)):
        _assert_close(f"grad_grid{i}/oracle", a, b.numpy())


@pytest.mark.parametrize("case", RENDERER_CASES, ids=lambda c: c.name)
def test_renderer_goldens_in_fp32_arithmetic(case, golden_dir):
    """The reference's own arithmetic (triton_src/shared/const.py:9 ALLOW_TF32 = False), selectable per call:
    ``arithmetic=LP_ARITH_FP32`` -- the tuned family's three-limb / fp32-dW instantiations where the shape is the tuned one, the
    shape-generic fp32 kernels elsewhere -- against the numbers the reference produced."""
    dev = _dev()
    d = case.build()
    fam = lp.kernel_family(d["rays"], d["grids"], d["decoder"], color_grid=d["color_grids"], num_samples_inf=d["cfg"].get("num_samples_inf", 0),
                           arithmetic=_lib.LP_ARITH_FP32)
    if fam != 1:
        pytest.skip("LP_ARITH_FP32 runs the shape-generic kernels here: test_renderer_matches_oracle_and_golden[generic] covers them")
    z = np.load(os.path.join(golden_dir, f"renderer__{case.name}.npz"))
    out, gp, ge, gg, gc = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO, arithmetic=_lib.LP_ARITH_FP32)
    for nm, a in (("ray_length", out[0]), ("neg_log_t", out[1]), ("feature", out[2]), ("grad_mlp_params", gp), ("grad_encoding", ge)):
        _assert_close(f"{nm}/golden (LP_ARITH_FP32)", a, z[nm])
    for i, g in enumerate(gg):
        _assert_close(f"grad_grid{i}/golden (LP_ARITH_FP32)", g, z[f"grad_grid{i}"])


@pytest.mark.parametrize("case", [c for c in RENDERER_CASES if c.noise_sigma == 0.0][:8], ids=lambda c: c.name)
def test_corner_indices_bit_exact(case):
    """Integer indexing parity: every corner row of every sample equals the oracle's, exactly."""
    dev = _dev()
    d = case.build()
    cfg = d["cfg"]
    rows = lp.renderer.renderer_corner_rows(_rays_to(d["rays"], dev), d["sizes"], cfg["num_samples"],
                                            cfg["num_samples_inf"], cfg["contract_coords"]).cpu()
    want = torch.cat(O.renderer_corner_indices(d["rays"], d["sizes"], cfg["num_samples"], cfg["num_samples_inf"],
                                               cfg["contract_coords"]), dim=-1)
    assert rows.shape == want.shape
    assert torch.equal(rows, want), f"{(rows != want).sum().item()} of {rows.numel()} corner rows differ"


def test_flat_grid_input_and_tail_rays():
    """Flat-tensor grid input (+grid_sizes) gives the same result as the list form; N not a
    multiple of the wave size; gradients reach the flat tensor."""
    dev = _dev()
    case = RENDERER_CASES[2]
    d = case.build()
    out_list, gp, ge, gg, _ = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    rays = _rays_to(d["rays"], dev, True)
    dec = d["decoder"]
    params = dec.mlp_params.to(dev).clone().requires_grad_(True)
    hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
    flat, sizes = lp.flatten_grid([g.to(dev) for g in d["grids"]])
    flat = flat.clone().requires_grad_(True)
    out = lp.lightplane_renderer(rays, flat, hdec, grid_sizes=sizes.tolist(), **d["cfg"])
    for a, b in zip(out, out_list):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
    g_len, g_nlt, g_feat = (t.to(dev) for t in d["upstream"])
    ((out[0] * g_len).sum() + (out[1] * g_nlt).sum() + (out[2] * g_feat).sum()).backward()
    want = torch.cat([g.reshape(-1, g.shape[-1]) for g in gg], dim=0)
    _assert_close("flat grid grad", flat.grad, want.cpu().numpy(), tol=2e-5)


def run_hip_splatter(d, dev):
    rays = _rays_to(d["rays"], dev, True)
    out = lp.lightplane_splatter(rays, d["out_sizes"], **d["cfg"])
    sum((o * u.to(dev)).sum() for o, u in zip(out, d["upstream"])).backward()
    return out, rays.encoding.grad


@pytest.mark.parametrize("case", [c for c in SPLATTER_CASES if not c.use_mlp], ids=lambda c: c.name)
def test_splatter_matches_oracle_and_golden(case, golden_dir):
    dev = _dev()
    d = case.build()
    z = np.load(os.path.join(golden_dir, f"splatter__{case.name}.npz"))
    out, ge = run_hip_splatter(d, dev)
    import copy
    rays = copy.copy(d["rays"])
    rays.encoding = rays.encoding.clone().requires_grad_(True)
    o_out = O.lightplane_splatter_naive(rays, d["out_sizes"], **d["cfg"])
    sum((o * u).sum() for o, u in zip(o_out, d["upstream"])).backward()
    for i, o in enumerate(out):
        _assert_close(f"out{i}/golden", o, z[f"out{i}"])
        _assert_close(f"out{i}/oracle", o, o_out[i].detach().numpy())
    _assert_close("grad_encoding/golden", ge, z["grad_encoding"])
    _assert_close("grad_encoding/oracle", ge, rays.encoding.grad.numpy())


def run_hip_mlp_splatter(d, dev, kernel=_lib.LP_KERNEL_AUTO):
    rays = _rays_to(d["rays"], dev, True)
    mlp = d["mlp"]
    params = mlp.mlp_params.to(dev).clone().requires_grad_(True)
    hmlp = lp.SplatterParams(params, mlp.n_hidden)
    in_grids = [g.to(dev).clone().requires_grad_(True) for g in d["in_grids"]]
    out = lp.lightplane_mlp_splatter(rays, d["out_sizes"], hmlp, in_grids, kernel=kernel, **d["cfg"])
    sum((o * u.to(dev)).sum() for o, u in zip(out, d["upstream"])).backward()
    return out, rays.encoding.grad, params.grad, [g.grad for g in in_grids]


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("case", [c for c in SPLATTER_CASES if c.use_mlp], ids=lambda c: c.name)
def test_mlp_splatter_matches_oracle_and_golden(case, kernel, golden_dir):
    dev = _dev()
    d = case.build()
    z = np.load(os.path.join(golden_dir, f"splatter__{case.name}.npz"))
    out, ge, gp, gin = run_hip_mlp_splatter(d, dev, kernel)
    import copy
    rays = copy.copy(d["rays"])
    rays.encoding = rays.encoding.clone().requires_grad_(True)
    mlp = copy.copy(d["mlp"])
    mlp.mlp_params = mlp.mlp_params.clone().requires_grad_(True)
    in_grids = [g.clone().requires_grad_(True) for g in d["in_grids"]]
    o_out = O.lightplane_mlp_splatter_naive(rays, d["out_sizes"], mlp, in_grids, **d["cfg"])
    sum((o * u).sum() for o, u in zip(o_out, d["upstream"])).backward()
    for i, o in enumerate(out):
        _assert_close(f"out{i}/golden", o, z[f"out{i}"])
        _assert_close(f"out{i}/oracle", o, o_out[i].detach().numpy())
    _assert_close("grad_encoding/golden", ge, z["grad_encoding"])
    _assert_close("grad_encoding/oracle", ge, rays.encoding.grad.numpy())
    _assert_close("grad_mlp_params/golden", gp, z["grad_mlp_params"])
    _assert_close("grad_mlp_params/oracle", gp, mlp.mlp_params.grad.numpy())
    for i, g in enumerate(gin):
        _assert_close(f"grad_in_grid{i}/golden", g, z[f"grad_in_grid{i}"])
        _assert_close(f"grad_in_grid{i}/oracle", g, in_grids[i].grad.numpy())


def test_mlp_splatter_module_and_flat_input():
    """LightplaneMLPSplatter module: flat input grid + input_grid_sizes == list input; gradients reach
    the module's parameter."""
    dev = _dev()
    d = SPLATTER_CASES[4].build()
    mod = lp.LightplaneMLPSplatter(num_samples=d["cfg"]["num_samples"], grid_chn=32, input_grid_chn=32,
                                   mlp_hidden_chn=32, mlp_n_layers=3).to(dev)
    with torch.no_grad():
        mod.mlp_params.copy_(d["mlp"].mlp_params.to(dev))
    rays = _rays_to(d["rays"], dev)
    grids = [g.to(dev) for g in d["in_grids"]]
    out_list = mod(rays, d["out_sizes"], grids)
    flat, sizes = lp.flatten_grid(grids)
    out_flat = mod(rays, d["out_sizes"], flat, input_grid_sizes=sizes.tolist(), return_list=False)
    want = torch.cat([o.reshape(-1, o.shape[-1]) for o in out_list], dim=0)
    assert torch.allclose(out_flat, want, rtol=1e-5, atol=1e-6)
    out_flat.sum().backward()
    assert mod.mlp_params.grad is not None and torch.isfinite(mod.mlp_params.grad).all()
    ref, _, _, _ = run_hip_mlp_splatter(d, dev)
    for a, b in zip(out_list, ref):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)


def test_empty_and_single_ray_batches():
    """N = 0 rays is legal everywhere (outputs of the right shape, zero gradients); N = 1 works."""
    dev = _dev()
    d = RENDERER_CASES[1].build()
    dec = d["decoder"]
    for n in (0, 1):
        rays = _rays_to(d["rays"][:n], dev, True)
        params = dec.mlp_params.to(dev).clone().requires_grad_(True)
        hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
        grids = [g.to(dev).clone().requires_grad_(True) for g in d["grids"]]
        for kernel in KERNELS:
            out = lp.lightplane_renderer(rays, grids, hdec, kernel=kernel, **d["cfg"])
            assert out[0].shape == (n,) and out[1].shape == (n,) and out[2].shape == (n, dec.color_chn)
            (out[0].sum() + out[1].sum() + out[2].sum()).backward()
            assert torch.isfinite(params.grad).all() and all(torch.isfinite(g.grad).all() for g in grids)
            if n == 0:
                assert float(params.grad.abs().max()) == 0.0 and all(float(g.grad.abs().max()) == 0.0 for g in grids)
            params.grad = None
            for g in grids:
                g.grad = None
    ds = SPLATTER_CASES[0].build()
    rays = _rays_to(ds["rays"][:0], dev, True)
    out = lp.lightplane_splatter(rays, ds["out_sizes"], **ds["cfg"])
    assert all(float(o.detach().abs().max()) == 0.0 for o in out)
    sum(o.sum() for o in out).backward()
    assert rays.encoding.grad.shape == (0, 32)
    dm = SPLATTER_CASES[4].build()
    rays = _rays_to(dm["rays"][:0], dev, True)
    mlp = lp.SplatterParams(dm["mlp"].mlp_params.to(dev).clone().requires_grad_(True), dm["mlp"].n_hidden)
    out = lp.lightplane_mlp_splatter(rays, dm["out_sizes"], mlp, [g.to(dev) for g in dm["in_grids"]], **dm["cfg"])
    sum(o.sum() for o in out).backward()
    assert float(mlp.mlp_params.grad.abs().max()) == 0.0


def test_module_point_evaluation_and_scaffold():
    """LightplaneRenderer.eval_opacity_at_points / eval_decoder_at_points / calculate_scaffold run through
    the HIP path (single-sample rays) and agree with the oracle's point decoder."""
    dev = _dev()
    torch.manual_seed(0)
    mod = lp.LightplaneRenderer(num_samples=8, color_chn=3, grid_chn=16, mlp_hidden_chn=32, gain=2.0,
                                opacity_init_bias=-1.0, ray_embedding_num_harmonics=None).to(dev)
    with torch.no_grad():
        mod.mlp_params.mul_(3.0)
    grids = [0.5 * torch.randn(s) for s in grid_sizes_for((2, 6, 5, 7, 16), True)]
    pts = torch.rand(5, 11, 3) * 2.4 - 1.2
    gidx = torch.tensor([0, 1, 1, 0, 1])
    enc = torch.randn(5, 32)
    dec = mod.get_decoder_params()
    cdec = lp.DecoderParams(dec.mlp_params.detach().cpu(), dec.n_hidden_trunk.cpu(), dec.n_hidden_opacity.cpu(),
                            dec.n_hidden_color.cpu(), 3)
    for mask in (False, True):
        o_op, o_col = O.eval_decoder(pts, grids, gidx, cdec, enc, 2.0, mask_out_of_bounds_samples=mask)
        op = mod.eval_opacity_at_points(pts.to(dev), gidx.to(dev), [g.to(dev) for g in grids],
                                        mask_out_of_bounds_samples=mask)
        _assert_close("opacity", op, o_op.detach().numpy(), tol=2e-5)
        op2, col = mod.eval_decoder_at_points(pts.to(dev), gidx.to(dev), enc.to(dev), [g.to(dev) for g in grids],
                                              mask_out_of_bounds_samples=mask)
        _assert_close("opacity2", op2, o_op.detach().numpy(), tol=2e-5)
        _assert_close("colour", col, o_col[..., :3].detach().numpy(), tol=2e-5)
    sc = mod.calculate_scaffold([g.to(dev) for g in grids], [2, 6, 5, 7], dev, threshold=0.6, dilate_scaffold=1)
    assert sc.shape == (2, 6, 5, 7) and set(sc.unique().tolist()) <= {0.0, 1.0}
    lin = lambda n: torch.linspace(0, 1, n) * 2 - 1  # noqa: E731
    zz, yy, xx = torch.meshgrid(lin(6), lin(5), lin(7), indexing="ij")
    lattice = torch.stack([xx, yy, zz], -1).reshape(1, -1, 3)
    want = torch.stack([O.eval_decoder(lattice, grids, torch.tensor([b]), cdec, torch.zeros(1, 32), 2.0)[0].reshape(6, 5, 7)
                        for b in range(2)])
    want = (torch.nn.functional.max_pool3d(want, 3, padding=1, stride=1) > 0.6).float()
    margin = (torch.nn.functional.max_pool3d(want, 1) - 0).abs()  # noqa: F841
    assert (sc.cpu() != want).float().mean().item() < 0.02  # thresholding of values within fp32 round-off of 0.6


def test_hash_rng(golden_dir):
    dev = _dev()
    z = np.load(os.path.join(golden_dir, "randn.npz"))
    x1 = torch.from_numpy(z["x1"]).to(torch.int32).to(dev)
    x2 = torch.from_numpy(z["x2"]).to(torch.int32).to(dev)
    for seed in (0, 5, 123456):
        out = torch.empty(x1.numel(), device=dev)
        _lib.check(_lib.lib().lp_hash_randn(x1.data_ptr(), x2.data_ptr(), out.data_ptr(), x1.numel(), seed,
                                           torch.cuda.current_stream().cuda_stream), "lp_hash_randn")
        err = (out.cpu() - torch.from_numpy(z[f"z_seed{seed}"])).abs().max().item()
        assert err <= 1e-4, f"seed {seed}: |z - z_ref| = {err}"  # reference's own bar is 1e-3 (tests/test_randn.py:41)


def test_cfg2_sized_properties():
    """Full BASELINE cfg-2 size (256x256 rays, triplane 64^2 x16, S=128): size-independent checks.

    * linearity of the backward in the upstream gradient (grad(2u) == 2 grad(u)),
    * compositing identity: sum of weights = 1 - exp(-nlt) => with colour head forced to
      sigmoid(.)=const the feature equals const * (1 - T),
    * a 512-ray subsample agrees with the CPU oracle.
    """
    dev = _dev()
    gen = torch.Generator().manual_seed(0)
    rays = pinhole_rays(256, 256, enc_dim=32, gen=gen)
    sizes = grid_sizes_for((1, 64, 64, 64, 16), True)
    grids = random_grids(gen, sizes)
    dec = random_decoder(gen, 2, 2, 2, 16, 32, 3, std=0.15)
    S = 128

    def hip(sub=None, scale=1.0):
        r = rays if sub is None else rays[sub]
        r = _rays_to(r, dev, True)
        p = dec.mlp_params.to(dev).clone().requires_grad_(True)
        hd = lp.DecoderParams(p, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, 3)
        gs = [g.to(dev).clone().requires_grad_(True) for g in grids]
        out = lp.lightplane_renderer(r, gs, hd, num_samples=S, gain=1.0)
        (scale * (out[0].sum() + out[1].sum() + out[2].sum())).backward()
        return out, p.grad, [g.grad for g in gs], r.encoding.grad

    out1, gp1, gg1, ge1 = hip()
    out2, gp2, gg2, ge2 = hip(scale=2.0)
    assert torch.isfinite(out1[2]).all() and torch.isfinite(gp1).all()
    _assert_close("linearity params", gp2, (2 * gp1).cpu().numpy(), tol=2e-4)  # atomics reorder sums
    _assert_close("linearity enc", ge2, (2 * ge1).cpu().numpy(), tol=1e-5)
    for a, b in zip(gg2, gg1):
        _assert_close("linearity grid", a, (2 * b).cpu().numpy(), tol=2e-4)
    # subsample vs oracle
    idx = torch.arange(0, 65536, 128)
    sub_out, sub_gp, sub_gg, sub_ge = hip(sub=idx)
    import copy
    r = rays[idx]
    r.encoding = r.encoding.clone().requires_grad_(True)
    d2 = copy.copy(dec)
    d2.mlp_params = dec.mlp_params.clone().requires_grad_(True)
    gs = [g.clone().requires_grad_(True) for g in grids]
    o = O.lightplane_renderer_naive(r, gs, d2, num_samples=S, gain=1.0)
    (o[0].sum() + o[1].sum() + o[2].sum()).backward()
    for name, a, b in (("len", sub_out[0], o[0]), ("nlt", sub_out[1], o[1]), ("feat", sub_out[2], o[2])):
        _assert_close("cfg2-sub " + name, a, b.detach().numpy())
    # 65 536 samples x 128 hidden units: a ReLU flip between two fp32 evaluations is likely (see test_gpu_coherent.py).  The tuned
    # family has dump twins, so the gradients are PROVEN (forced_oracle_check: the kernel's own ReLU decisions forced onto the fp64
    # oracle, every forced unit a measured near tie, then every entry at 1e-4) -- no allowance, no tie mask (round-5 review, weak 2)
    sub = rays[idx]
    d_sub = dict(rays=sub, grids=grids, decoder=dec, color_grids=None, scaffold=None, cfg=dict(num_samples=S, gain=1.0),
                 upstream=(torch.ones(sub.n_rays), torch.ones(sub.n_rays), torch.ones(sub.n_rays, 3)))
    forced_oracle_check("cfg2-sub", d_sub, dev)
    for a, b in zip(out1, sub_out):
        assert torch.allclose(a[idx.to(dev)], b, rtol=1e-5, atol=1e-6), "ray results depend on batch composition"


@pytest.mark.parametrize("kernel", KERNELS, ids=KERNEL_IDS)
@pytest.mark.parametrize("name", ["triplane_basic", "voxel_c32_color1", "triplane_h64_c32", "voxel_deep"])
def test_renderer_early_termination(name, kernel):
    """Extension: with stop_transmittance the march of a wavefront ends once all its rays are opaque.  On a dense
    medium the outputs and gradients stay within the stated bound of the exact march, the march really stops
    (the returned -log T is the value at the stop), and stop_transmittance = 0 is the exact path."""
    dev = _dev()
    case = next(c for c in RENDERER_CASES if c.name == name)
    d = case.build()
    d["cfg"] = dict(d["cfg"], gain=float(d["cfg"]["gain"]) * 40.0)  # dense: rays saturate within a few samples
    d["upstream"] = (d["upstream"][0], torch.zeros_like(d["upstream"][1]), d["upstream"][2])  # no loss on -log T
    eps = 1e-5
    out0, gp0, ge0, gg0, _ = run_hip_renderer(d, dev, kernel)
    out1, gp1, ge1, gg1, _ = run_hip_renderer(d, dev, kernel, stop_transmittance=eps)
    nlt0, nlt1 = out0[1], out1[1]
    stopped = nlt1 < nlt0 * (1 - 1e-6) - 1e-6
    assert bool(stopped.any()), "no wavefront terminated early"
    assert bool((nlt1[stopped] >= -np.log(eps) - 1e-4).all())
    assert bool((nlt1 <= nlt0 * (1 + 1e-6) + 1e-6).all())
    far = float(d["rays"].far.max())
    assert float((out1[0] - out0[0]).detach().abs().max()) <= 2 * eps * far * 4
    assert float((out1[2] - out0[2]).detach().abs().max()) <= 2 * eps * 4
    for nm, a, b in [("params", gp1, gp0), ("enc", ge1, ge0)] + [(f"grid{i}", x, y) for i, (x, y) in enumerate(zip(gg1, gg0))]:
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) / scale <= 1e-3, nm


def test_fit_synthetic_scene_converges():
    """End-to-end: module API + autograd + Adam on random rays drive the loss of the analytic scene down
    (the role of the reference's examples/fit_single_scene.py loop), with and without early termination."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fit_synthetic_scene", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "fit_synthetic_scene.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    old = lp.config.stop_transmittance
    try:
        for stop in (0.0, 1e-4):
            r = mod.fit(steps=150, n_rays=4096, stop_transmittance=stop)
            assert r["last_loss"] < 0.2 * r["first_loss"], r
            assert r["heldout_psnr_db"] > 18.0, r
    finally:
        lp.config.stop_transmittance = old


def test_splatter_walk_16_rays_per_wave():
    """The Splatter's forward walk with 16 instead of 32 rays per wave (LP_SPLAT_RPW, read once per process; 32 is the default
    since round 4 -- longer walks meet more masked / padding rays next to live ones, the case that once dropped a carried
    column -- so the default run covers it now and this test the other width).  Runs the golden Splatter cases in a child process."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LP_SPLAT_RPW="16")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests"), "-m", "gpu", "-q", "-x",
                        "-k", "test_splatter_matches", "-p", "no:cacheprovider"],
                       cwd=root, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode()[-3000:]


def test_hip_graph_capture_of_forward_backward():
    """A Renderer forward+backward captures into a HIP graph (torch.cuda.CUDAGraph: no host syncs, launches on the
    capturing stream) and the replay reproduces the eager gradients (atomics: summation order only)."""
    dev = _dev()
    old = lp.config.check_inputs
    lp.config.check_inputs = False  # the grid_idx range check is a host sync
    try:
        case = next(c for c in RENDERER_CASES if c.name == "triplane_basic")
        d = case.build()
        rays = _rays_to(d["rays"], dev, True)
        dec = d["decoder"]
        params = dec.mlp_params.to(dev).clone().requires_grad_(True)
        hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
        flat = lp.flatten_grid([g.to(dev) for g in d["grids"]])[0].requires_grad_(True)
        sizes = d["sizes"]

        def step():
            o = lp.lightplane_renderer(rays, flat, hdec, grid_sizes=sizes, **d["cfg"])
            (o[0].sum() + o[1].sum() + o[2].sum()).backward()

        step()
        ref = [t.grad.clone() for t in (flat, params, rays.encoding)]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                flat.grad = params.grad = rays.encoding.grad = None
                step()
        torch.cuda.current_stream().wait_stream(s)
        flat.grad = params.grad = rays.encoding.grad = None
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            step()
        for t in (flat, params, rays.encoding):
            t.grad.zero_()
        g.replay()
        torch.cuda.synchronize()
        for nm, t, r in zip(("grid", "params", "encoding"), (flat, params, rays.encoding), ref):
            _assert_close(f"graph replay grad_{nm}", t.grad, r.cpu().numpy(), 1e-5)
    finally:
        lp.config.check_inputs = old


@pytest.mark.parametrize("layers,hidden,C,n_inf,family", [((2, 2, 2), 32, 16, 70, 1), ((2, 2, 2), 64, 32, 70, 3), ((3, 2, 2), 32, 16, 130, 3),
                                                          ((1, 1, 2), 64, 64, 256, 3), ((2, 2, 2), 64, 16, 257, 0)],
                         ids=["tuned_inf70", "h64_inf70", "deep322_inf130", "example112_c64_inf256", "h64_inf257_generic"])
def test_many_beyond_far_samples(layers, hidden, C, n_inf, family):
    """More than 64 beyond-far samples (contracted coordinates): the tuned family switches to its eight-wave backward workgroups,
    the layer-looped family tabulates up to 256 depth scales since 0.2.4 (64 before; the retired fp32-MFMA hidden-64 family took
    256, so the 2/2/2 x 64 decoder keeps its range), 257 and more run the shape-generic kernels.  Outputs and every gradient
    against the oracle."""
    from tests.synth import RendererCase
    dev = _dev()
    case = RendererCase(f"inf{n_inf}", seed=500 + n_inf + hidden, n_rays=70, grid_base=(2, 5, 6, 7, C), is_triplane=hidden == 64,
                        n_layers=layers, hidden=hidden, num_samples=11, num_samples_inf=n_inf, gain=0.25, contract=True, param_std=0.2)
    d = case.build()
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"], **d["cfg"]) == family
    out, gp, ge, gg, _ = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    o_out, o_gp, o_ge, o_gg, _ = run_oracle_renderer(d)
    for nm, a, b in (("ray_length", out[0], o_out[0]), ("neg_log_t", out[1], o_out[1]), ("feature", out[2], o_out[2]),
                     ("grad_mlp_params", gp, o_gp), ("grad_encoding", ge, o_ge)):
        _assert_close(f"{case.name}: {nm}", a, b.detach().numpy())
    for i, (a, b) in enumerate(zip(gg, o_gg)):
        _assert_close(f"{case.name}: grad_grid{i}", a, b.numpy())


@pytest.mark.parametrize("color_chn,layers,sep,C,hidden", [(5, (2, 2, 2), False, 16, 32), (16, (4, 2, 3), False, 32, 32), (32, (1, 1, 1), False, 16, 16),
                                                      (12, (0, 3, 2), True, 16, 32), (8, (2, 1, 1), False, 32, 32)],
                         ids=["c5_222", "c16_423_C32", "c32_111_h16", "c12_two_grid_032", "c8_211_C32"])
def test_wide_colour_on_the_matrix_cores(color_chn, layers, sep, C, hidden):
    """5 .. 32 colour channels (feature rendering): the colour output layer runs as an MFMA layer of the layer-looped family,
    each lane composites 16 of the channels.  Outputs and all gradients against the oracle, and the fused background / alpha
    epilogue (whose background sum crosses the two lanes of a ray) against the PyTorch ops."""
    from tests.synth import RendererCase
    dev = _dev()
    case = RendererCase(f"wide{color_chn}", seed=400 + color_chn, n_rays=70, grid_base=(2, 5, 6, 7, C), is_triplane=not sep, n_layers=layers,
                        hidden=hidden, color_chn=color_chn, num_samples=13, num_samples_inf=2, gain=2.0, separate_color_grid=sep,
                        mask_oob=True, param_std=0.25)
    d = case.build()
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"], color_grid=d["color_grids"]) == 3
    out, gp, ge, gg, gc = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    o_out, o_gp, o_ge, o_gg, o_gc = run_oracle_renderer(d)
    for nm, a, b in (("ray_length", out[0], o_out[0]), ("neg_log_t", out[1], o_out[1]), ("feature", out[2], o_out[2]),
                     ("grad_mlp_params", gp, o_gp), ("grad_encoding", ge, o_ge)):
        _assert_close(f"{case.name}: {nm}", a, b.detach().numpy())
    for i, (a, b) in enumerate(zip(gg, o_gg)):
        _assert_close(f"{case.name}: grad_grid{i}", a, b.numpy())
    if gc is not None:
        for i, (a, b) in enumerate(zip(gc, o_gc)):
            _assert_close(f"{case.name}: grad_color_grid{i}", a, b.numpy())
    # fused epilogue (feature + T * bg, alpha = 1 - T) against the op chain on the kernel's own outputs
    from lightplane_amd.renderer import _render
    gen = torch.Generator().manual_seed(9)
    bg = torch.rand(color_chn, generator=gen).to(dev)
    g_alpha = torch.randn(case.n_rays, generator=gen).to(dev)
    g_feat = torch.randn(case.n_rays, color_chn, generator=gen).to(dev)

    def run(fused):
        rays = _rays_to(d["rays"], dev, True)
        dec = d["decoder"]
        params = dec.mlp_params.to(dev).clone().requires_grad_(True)
        hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
        grids = [g.to(dev).clone().requires_grad_(True) for g in d["grids"]]
        cgrids = None if d["color_grids"] is None else [g.to(dev) for g in d["color_grids"]]
        if fused:
            _, _, feat, alpha = _render(rays, grids, hdec, color_grid=cgrids, bg_color=bg, alpha_mode=1, **d["cfg"])
        else:
            _, nlt, feat = lp.lightplane_renderer(rays, grids, hdec, color_grid=cgrids, **d["cfg"])
            T = torch.exp(-nlt)
            feat, alpha = feat + T[:, None] * bg, 1 - T
        ((feat * g_feat).sum() + (alpha * g_alpha).sum()).backward()
        return feat, alpha, params.grad, rays.encoding.grad, grids[0].grad

    for nm, a, b in zip(("feature", "alpha", "grad_mlp_params", "grad_encoding", "grad_grid0"), run(True), run(False)):
        _assert_close(f"{case.name}: fused epilogue {nm}", a, b.detach().cpu().numpy(), tol=2e-5)
