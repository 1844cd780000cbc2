"""Seeded random sweep of the HIP kernels against the CPU oracle (GPU box), in the spirit of the reference's own
sweeps (tests/test_renderer_with_autograd.py:35-56, tests/test_splatter_with_autograd.py:38-53): random grid types
and sizes, channels, decoder shapes, ray counts that leave partial waves, masks, contraction, beyond-far samples,
noise, scaffold, separate colour grid.  The named golden cases pin values the reference produced; this sweep walks
the combinations of code paths (MFMA families, flex / two-grid instantiations, column and window walks of the
Splatter, early-termination bookkeeping with stop = 0) that no single named case covers."""
import copy
import random

import numpy as np
import pytest
import torch

import lightplane_amd as lp
from lightplane_amd import _lib
from oracle import lightplane_oracle as O
from tests.synth import RendererCase, SplatterCase
from tests.test_gpu_parity import (_dev, _rel_err, assert_grad_close, forced_oracle_check, has_dump_twin, run_hip_mlp_splatter, run_hip_renderer,
                                   run_hip_splatter, run_oracle_renderer)

pytestmark = pytest.mark.gpu

F64 = torch.float64
TOL = 1e-4  # north_star's bar (2e-4 until round 4)
ILL_CONDITIONED = 3.0


def _check(name, got, want64, ref32=None, inf=False, grad_entries=None):
    """north_star's 1e-4 against the fp32 oracle (the reference's own arithmetic: the cell a sample falls into, the scaffold cell
    it looks up and the out-of-bounds mask are DEFINED by fp32 coordinate arithmetic) or against the fp64 oracle (where the fp32
    oracle's own summation order carries an error of that size).  Measured over all 104 sweep cases / 813 tensors
    (scripts/sweep_errors_all.py, profiles/r05_sweep_errors.json): 806 tensors meet 1e-4 against one of the two outright; the 7
    others are gradients of three cases with beyond-far samples, where interval lengths ~1e5 make the REFERENCE's fp32 gradients
    cancel catastrophically -- the fp32 oracle itself is 5e-4 .. 5.7e-3 away from fp64 there, the kernel 1.2e-3 .. 9.5e-3
    (at most 2.2x as far: sweep25 grad_mlp_params 2.6e-3 against 1.2e-3).  So:
    * ``inf`` cases only, and only for a tensor on which the fp32 oracle itself misses 1e-4 against fp64 ("1e-4 of the naive
      reference" is then not defined to better than that error): the kernel has to be within ILL_CONDITIONED (3x) of the fp32
      oracle's own error against fp64 (round 4: the same factor, but without the precondition and at a 2e-4 bar);
    * gradient tensors (``grad_entries`` = entries one sample touches): the counted ReLU-flip allowance of assert_grad_close
      (not needed by any of the 813 tensors at the committed seeds; kept because atomics reorder sums from run to run)."""
    e64 = _rel_err(got, want64.detach().numpy())
    if e64 <= TOL:
        return
    e32 = _rel_err(got, ref32.detach().numpy()) if ref32 is not None else float("inf")
    if e32 <= TOL:
        return
    if inf and ref32 is not None:
        own = _rel_err(ref32, want64.detach().numpy())
        if own > TOL:
            assert e64 <= ILL_CONDITIONED * own, (f"{name}: max err / scale = {e64:.3e} against fp64 > {ILL_CONDITIONED:g} x the fp32 oracle's own "
                                                  f"error {own:.3e} (beyond-far samples: fp32 ill-conditioned)")
            return
    if grad_entries is not None and ref32 is not None:
        assert_grad_close(name, got, ref32.detach().numpy(), grad_entries, tol=TOL, want64=want64.detach().numpy())
        return
    assert False, f"{name}: max err / scale = {e64:.3e} (fp64 oracle), {e32:.3e} (fp32 oracle) > {TOL:g}"


def _rays64(rays):
    r = copy.copy(rays)
    for f in ("directions", "origins", "near", "far", "encoding"):
        setattr(r, f, getattr(r, f).to(F64))
    r.encoding = r.encoding.clone().requires_grad_(True)
    return r


def _splatter_oracle(case, d, dtype):
    """The Splatter / MLP-Splatter oracle on the case's inputs in ``dtype``: (outputs, grad_encoding, grad_mlp_params | None,
    grad_input_grids | None)."""
    rays = copy.copy(d["rays"])
    for f in ("directions", "origins", "near", "far", "encoding"):
        setattr(rays, f, getattr(rays, f).to(dtype))
    rays.encoding = rays.encoding.clone().requires_grad_(True)
    up = [u.to(dtype) for u in d["upstream"]]
    if case.use_mlp:
        mlp = copy.copy(d["mlp"])
        mlp.mlp_params = mlp.mlp_params.to(dtype).clone().requires_grad_(True)
        in_grids = [g.to(dtype).clone().requires_grad_(True) for g in d["in_grids"]]
        o_out = O.lightplane_mlp_splatter_naive(rays, d["out_sizes"], mlp, in_grids, **d["cfg"])
        sum((o * u).sum() for o, u in zip(o_out, up)).backward()
        return [o.detach() for o in o_out], rays.encoding.grad, mlp.mlp_params.grad, [g.grad for g in in_grids]
    o_out = O.lightplane_splatter_naive(rays, d["out_sizes"], **d["cfg"])
    sum((o * u).sum() for o, u in zip(o_out, up)).backward()
    return [o.detach() for o in o_out], rays.encoding.grad, None, None


SPLAT_TOL = 1e-4  # north_star's bar; the Splatter sweeps held 2e-4 until round 3 (review, weak 2)


def _check_splatter_all(case, name, d, dev):
    """HIP Splatter / MLP-Splatter against the fp32 oracle at north_star's 1e-4 (the cell a sample falls into is DEFINED by the
    fp32 index arithmetic).  A tensor that misses it may instead meet 1e-4 against the fp64 oracle (the fp32 oracle's own
    scatter_add order carries error of that size on some seeds); gradients downstream of the MLP's ReLUs get the counted
    ReLU-flip allowance of assert_grad_close with the fp64 oracle as second opinion.  Outputs never get an allowance."""
    if case.use_mlp:
        out, ge, gp, gin = run_hip_mlp_splatter(d, dev)
    else:
        out, ge = run_hip_splatter(d, dev)
        gp = gin = None
    o32 = _splatter_oracle(case, d, torch.float32)
    q = []

    def o64():
        if not q:
            q.append(_splatter_oracle(case, d, F64))
        return q[0]

    def one(nm, got, want32, pick64, grad_entries=None):
        if _rel_err(got, want32.numpy()) <= SPLAT_TOL:
            return
        want64 = pick64(o64())
        if _rel_err(got, want64.numpy()) <= SPLAT_TOL:
            return
        if grad_entries is not None:
            assert_grad_close(f"{name}: {nm}", got, want32.numpy(), grad_entries, tol=SPLAT_TOL, want64=want64.numpy())
            return
        raise AssertionError(f"{name}: {nm}: max err / scale = {_rel_err(got, want32.numpy()):.3e} (fp32 oracle), "
                             f"{_rel_err(got, want64.numpy()):.3e} (fp64 oracle) > {SPLAT_TOL}")

    for k, o in enumerate(out):
        one(f"out{k}", o, o32[0][k], lambda r, k=k: r[0][k])
    width = int(max(d["mlp"].n_hidden)) if case.use_mlp else None
    one("grad_encoding", ge, o32[1], lambda r: r[1], grad_entries=(ge.shape[1] if case.use_mlp else None))
    if case.use_mlp:
        one("grad_mlp_params", gp, o32[2], lambda r: r[2], grad_entries=4 * width)
        for k, (a, b) in enumerate(zip(gin, o32[3])):
            one(f"grad_input_grid{k}", a, b, lambda r, k=k: r[3][k], grad_entries=8 * a.shape[-1])


def run_oracle_renderer64(d):
    rays = _rays64(d["rays"])
    dec = copy.copy(d["decoder"])
    dec.mlp_params = dec.mlp_params.to(F64).clone().requires_grad_(True)
    grids = [g.to(F64).clone().requires_grad_(True) for g in d["grids"]]
    cgrids = None if d["color_grids"] is None else [g.to(F64).clone().requires_grad_(True) for g in d["color_grids"]]
    scaffold = None if d["scaffold"] is None else d["scaffold"].to(F64)
    out = O.lightplane_renderer_naive(rays, grids, dec, scaffold=scaffold, color_grid=cgrids, **d["cfg"])
    g_len, g_nlt, g_feat = (u.to(F64) for u in d["upstream"])
    ((out[0] * g_len).sum() + (out[1] * g_nlt).sum() + (out[2] * g_feat).sum()).backward()
    return out, dec.mlp_params.grad, rays.encoding.grad, [g.grad for g in grids], None if cgrids is None else [g.grad for g in cgrids]


def _renderer_case(i):
    rnd = random.Random(1000 + i)
    C = rnd.choice([16, 32])
    tri = rnd.random() < 0.5
    sep = rnd.random() < 0.3
    fam = rnd.choice(["default", "flex", "wide", "deep"]) if not sep else rnd.choice(["flex", "flex", "deep"])
    if fam == "default":
        layers, hidden = (2, 2, 2), 32
    elif fam == "flex":
        layers, hidden = (rnd.choice([1, 2]), rnd.choice([1, 2]), rnd.choice([1, 2])), rnd.choice([16, 32])
    elif fam == "wide":
        layers, hidden = (2, 2, 2), 64
    else:
        layers, hidden = (rnd.choice([1, 3, 4]), rnd.choice([2, 3]), rnd.choice([2, 4])), rnd.choice([16, 32])
    if sep:
        layers = (0, layers[1], layers[2])
    B = rnd.choice([1, 2, 3])
    base = (B, rnd.randint(3, 9), rnd.randint(3, 9), rnd.randint(3, 9), C)
    contract = rnd.random() < 0.3
    kw = dict(seed=5000 + i, n_rays=rnd.choice([1, 7, 31, 32, 33, 65, 127, 130]), grid_base=base, is_triplane=tri,
              extra_voxel=tri and rnd.random() < 0.3 and not sep, n_layers=layers, hidden=hidden,
              color_chn=rnd.choice([1, 3, 3, 4]), num_samples=rnd.choice([1, 2, 9, 33, 40]),
              num_samples_inf=rnd.choice([0, 0, 3]), gain=rnd.choice([1.0, 3.0]),
              mask_oob=(not contract) and rnd.random() < 0.4, contract=contract,
              scaffold_size=(rnd.randint(2, 6), rnd.randint(2, 6), rnd.randint(2, 6)) if rnd.random() < 0.3 else None,
              separate_color_grid=sep, noise_sigma=rnd.choice([0.0, 0.0, 0.7]), noise_seed=rnd.randint(0, 2 ** 20),
              param_std=0.25)
    if sep and rnd.random() < 0.5:
        kw["color_grid_base"] = (B, rnd.randint(3, 8), rnd.randint(3, 8), rnd.randint(3, 8), C)
        kw["color_is_triplane"] = rnd.random() < 0.5
    return RendererCase(f"sweep{i}", **kw)


@pytest.mark.parametrize("i", range(48))
def test_renderer_sweep(i):
    case = _renderer_case(i)
    d = case.build()
    dev = _dev()
    out, gp, ge, gg, gc = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    o_out, o_gp, o_ge, o_gg, o_gc = run_oracle_renderer64(d)
    r_out, r_gp, r_ge, r_gg, r_gc = run_oracle_renderer(d)  # the reference's fp32 arithmetic
    _check_renderer_all(case, d, (out, gp, ge, gg, gc), (o_out, o_gp, o_ge, o_gg, o_gc), (r_out, r_gp, r_ge, r_gg, r_gc))


def _check_renderer_all(case, d, got, o64, r32):
    """Outputs: 1e-4 against the fp32 or the fp64 oracle (_check).  Gradients of a case WITHOUT beyond-far samples on an MFMA family:
    the proof (forced_oracle_check: the kernel's ReLU decisions forced onto the fp64 oracle, every forced unit a near tie, every
    entry at 1e-4 -- measured over all 104 sweep cases, scripts/sweep_forced_errors.py, profiles/r06_sweep_forced.json: 101 proven,
    worst entry 7.6e-5, at most 3 units forced per case).  Cases WITH beyond-far samples keep _check's bars: three of them
    (sweep2, sweep25, refsweep14) have no flipped unit at all and still sit 2.3e-4 .. 9.5e-3 from fp64 -- interval lengths ~1e5 make
    the reference's own fp32 gradients cancel catastrophically there (the fp32 oracle is 5e-4 .. 5.7e-3 from fp64 itself)."""
    out, gp, ge, gg, gc = got
    o_out, o_gp, o_ge, o_gg, o_gc = o64
    r_out, r_gp, r_ge, r_gg, r_gc = r32
    inf = case.num_samples_inf > 0
    C = d["grids"][0].shape[-1]
    for nm, a, b, c in (("ray_length", out[0], o_out[0], r_out[0]), ("neg_log_t", out[1], o_out[1], r_out[1]),
                        ("feature", out[2], o_out[2], r_out[2])):
        _check(f"{case.name}: {nm}", a, b, c, inf=inf)
    if not inf and has_dump_twin(d):
        forced_oracle_check(case.name, d, _dev(), chunk=d["rays"].n_rays)
        return
    _check(f"{case.name}: grad_mlp_params", gp, o_gp, r_gp, inf=inf, grad_entries=4 * max(case.hidden, C))
    _check(f"{case.name}: grad_encoding", ge, o_ge, r_ge, inf=inf, grad_entries=ge.shape[1])
    for k, (a, b, c) in enumerate(zip(gg, o_gg, r_gg)):
        _check(f"{case.name}: grad_grid{k}", a, b, c, inf=inf, grad_entries=8 * C)
    if gc is not None:
        for k, (a, b, c) in enumerate(zip(gc, o_gc, r_gc)):
            _check(f"{case.name}: grad_color_grid{k}", a, b, c, inf=inf, grad_entries=8 * C)


def _segmented_case(i):
    """Default decoder shape (every fourth case: a 3-4 layer decoder of the layer-looped family), 17 .. 130 samples, no
    beyond-far samples: the segment-parallel forward + backward (ragged last segments, partial waves, several grid batch
    entries, the non-PLAIN instantiations)."""
    rnd = random.Random(3000 + i)
    C = rnd.choice([16, 32])
    tri = rnd.random() < 0.5
    B = rnd.choice([1, 2, 3])
    contract = rnd.random() < 0.3
    kw = dict(seed=9000 + i, n_rays=rnd.choice([1, 33, 130, 300, 1000]), grid_base=(B, rnd.randint(3, 9), rnd.randint(3, 9), rnd.randint(3, 9), C),
              is_triplane=tri, extra_voxel=tri and rnd.random() < 0.3,
              n_layers=(2, 2, 2) if i % 4 else rnd.choice([(4, 2, 3), (3, 4, 2), (1, 3, 3)]), hidden=32,  # every fourth: layer-looped family
              color_chn=rnd.choice([1, 3, 3, 4]), num_samples=rnd.choice([17, 31, 32, 33, 48, 65, 100, 130]), num_samples_inf=0,
              gain=rnd.choice([1.0, 3.0]), mask_oob=(not contract) and rnd.random() < 0.4, contract=contract,
              scaffold_size=(rnd.randint(2, 6), rnd.randint(2, 6), rnd.randint(2, 6)) if rnd.random() < 0.3 else None,
              noise_sigma=rnd.choice([0.0, 0.0, 0.7]), noise_seed=rnd.randint(0, 2 ** 20), param_std=0.25)
    return RendererCase(f"segsweep{i}", **kw)


@pytest.mark.parametrize("i", range(16))
def test_renderer_segmented_sweep(i):
    case = _segmented_case(i)
    d = case.build()
    dev = _dev()
    import os
    if not os.environ.get("LP_LOOP"):  # (LP_LOOP=1 sends the default shape through the layer-looped family: one sweep per ray)
        assert lp.backward_segments(d["rays"], d["grids"], d["decoder"], **d["cfg"]) == -(-case.num_samples // _lib.LP_SEG_LEN)
    out, gp, ge, gg, _ = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    o_out, o_gp, o_ge, o_gg, _ = run_oracle_renderer64(d)
    r_out, r_gp, r_ge, r_gg, _ = run_oracle_renderer(d)
    _check_renderer_all(case, d, (out, gp, ge, gg, None), (o_out, o_gp, o_ge, o_gg, None), (r_out, r_gp, r_ge, r_gg, None))


def _splatter_case(i):
    rnd = random.Random(2000 + i)
    C = rnd.choice([16, 32, 32, 8])
    B = rnd.choice([1, 2])
    tri = rnd.random() < 0.3
    use_mlp = rnd.random() < 0.4 and C in (16, 32)
    contract = rnd.random() < 0.3
    kw = dict(seed=7000 + i, n_rays=rnd.choice([1, 15, 16, 17, 33, 64, 70, 130]),
              out_base=(B, rnd.randint(3, 20), rnd.randint(3, 20), rnd.randint(3, 20), C), is_triplane=tri,
              num_samples=rnd.choice([1, 9, 24]), num_samples_inf=rnd.choice([0, 0, 3]),
              mask_oob=(not contract) and rnd.random() < 0.4, contract=contract)
    if use_mlp:
        kw.update(use_mlp=True, n_layers=rnd.choice([2, 2, 3]), feat_dim=rnd.choice([16, 32]),
                  in_base=(B, rnd.randint(3, 8), rnd.randint(3, 8), rnd.randint(3, 8), 32), in_triplane=rnd.random() < 0.4)
    return SplatterCase(f"sweep{i}", **kw)


@pytest.mark.parametrize("i", range(32))
def test_splatter_sweep(i):
    case = _splatter_case(i)
    _check_splatter_all(case, str(case.name), case.build(), _dev())


# ---------------------------------------------------------------------------------------------------------------------
# the reference's OWN sweep axes (tests/test_renderer_with_autograd.py:35-56), sampled
# ---------------------------------------------------------------------------------------------------------------------
def _reference_axes_case(i):
    """One combination of the reference's test_sweep dictionary: grid [3,16,12,8,16] (voxel or triplane), optional separate
    colour grid [4,3,9] (then no trunk), scaffold [6,4,5] or none, 16 samples + 11 or 0 beyond-far samples, 128 or 3 rays, gain 1 or
    3, mask / contraction on or off, noise 1.0 or 0, 2 or 4 layers per MLP, hidden 32, 3 colour channels.  Parameters N(0, 0.01)
    like the reference's fixture (tests/utils.py:349) for every second combination, N(0, 0.2) for the others."""
    rnd = random.Random(4242 + i)
    sep = rnd.random() < 0.5
    tri = rnd.random() < 0.5
    layers = (0 if sep else rnd.choice([2, 4]), rnd.choice([2, 4]), rnd.choice([2, 4]))
    kw = dict(seed=12000 + i, n_rays=rnd.choice([128, 3]), grid_base=(3, 16, 12, 8, 16), is_triplane=tri, n_layers=layers, hidden=32,
              color_chn=3, num_samples=16, num_samples_inf=rnd.choice([11, 0]), gain=rnd.choice([1.0, 3.0]),
              mask_oob=rnd.random() < 0.5, contract=rnd.random() < 0.5,
              scaffold_size=(6, 4, 5) if rnd.random() < 0.5 else None, separate_color_grid=sep,
              noise_sigma=rnd.choice([1.0, 0.0]), noise_seed=rnd.randint(0, 2 ** 20), param_std=0.01 if i % 2 == 0 else 0.2)
    if sep:
        kw["color_grid_base"] = (3, 4, 3, 9, 16)
        kw["color_is_triplane"] = tri
    return RendererCase(f"refsweep{i}", **kw)


@pytest.mark.filterwarnings("ignore:The renderer has been configured to contract")  # the reference's own sweep combines the two switches
@pytest.mark.parametrize("i", range(40))
def test_reference_sweep_axes(i):
    """Every sampled combination runs on an MFMA family (never the shape-generic kernels) and matches the oracle."""
    case = _reference_axes_case(i)
    d = case.build()
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"], color_grid=d["color_grids"]) in (1, 3), case
    dev = _dev()
    got = run_hip_renderer(d, dev, _lib.LP_KERNEL_AUTO)
    _check_renderer_all(case, d, got, run_oracle_renderer64(d), run_oracle_renderer(d))


def _reference_splatter_axes_case(i):
    """One valid combination of the reference's Splatter sweep (tests/test_splatter_with_autograd.py:38-53): output grid
    [2,16,12,8,32] (voxel / triplane), plain splat of 32 features or an MLP of 3 / 4 layers x 64 hidden reading an input grid
    [2,10,14,16,feat] with feat in {64, 32}, 16 + 11 samples, 1 or 128 rays, mask / contraction on or off."""
    rnd = random.Random(777 + i)
    tri = rnd.random() < 0.5
    use_mlp = rnd.random() < 0.7
    kw = dict(seed=13000 + i, n_rays=rnd.choice([1, 128]), out_base=(2, 16, 12, 8, 32), is_triplane=tri, num_samples=16, num_samples_inf=11,
              mask_oob=rnd.random() < 0.5, contract=rnd.random() < 0.5)
    if use_mlp:
        feat = rnd.choice([64, 32])
        kw.update(use_mlp=True, n_layers=rnd.choice([3, 4]), hidden=64, feat_dim=feat, in_base=(2, 10, 14, 16, feat), in_triplane=tri)
    return SplatterCase(f"refsplat{i}", **kw)


@pytest.mark.filterwarnings("ignore:The splatter has been configured")
@pytest.mark.parametrize("i", range(24))
def test_reference_splatter_sweep_axes(i):
    """Every sampled combination runs on the walk / MFMA families and matches the fp32 oracle (the cell a sample falls into is
    defined by fp32 index arithmetic)."""
    case = _reference_splatter_axes_case(i)
    _check_splatter_all(case, case.name, case.build(), _dev())
