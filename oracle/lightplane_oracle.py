"""CPU ORACLE for the Renderer / Splatter hot path -- TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement, in plain PyTorch tensor ops (CPU,
autograd-differentiable, fp32 or fp64), of the algorithm of the reference's
pure-PyTorch implementations:

* Renderer : reference ``lightplane/naive_renderer.py`` :197-325 (march +
  compositing), :328-501 (decoder), :625-731 (grid-list sampling), :758-813
  (MLP, noise indices, contraction, inverse-sphere depths)
* Splatter : reference ``lightplane/naive_splatter.py`` :185-289 (march),
  :315-385 (grid-list splat), :388-413 / :595-614 (scatter-add), :416-592 /
  :617-668 (corner weights)
* MLP layout: reference ``lightplane/mlp_utils.py`` :489-560, :691-721
* Hash RNG : reference ``lightplane/triton_src/shared/rand_util.py`` :110-145

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product path (``lightplane_amd``) never does.

PARITY PINNING.  The reference stores no golden vectors (all of its tests are
kernel-vs-naive differentials), so this oracle is pinned against OUTPUTS OF THE
REFERENCE ITSELF: ``tests/golden/make_golden.py`` imports the reference's
``lightplane_renderer_naive`` / ``lightplane_splatter_naive`` /
``lightplane_mlp_splatter_naive`` / ``int_to_randn_naive`` from
``/root/reference`` (CPU), runs them on seeded inputs and commits inputs'
seeds + outputs + gradients under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks this file against those fixtures.

Differences from the reference implementation *by construction* (all below
fp32 round-off and documented in DESIGN.md):

* interpolation is written as explicit corner gathers with integer indices
  instead of ``F.grid_sample`` so the integer cell indices are observable
  (``renderer_corner_indices``) and can be compared bit-exactly with the HIP
  kernels.  The un-normalisation is torch's ``((x + 1) * size - 1) / 2`` for
  the Renderer and the naive splatter's ``(x + 1) / 2 * size - 0.5`` for the
  Splatter, every operation individually rounded (no FMA contraction).
* ``linspace(0, 1, S)`` is evaluated with torch's *scalar* formula
  (``i < S//2 ? i*step : 1 - (S-1-i)*step``, identical to torch's GPU linspace);
  torch's vectorised CPU linspace differs from it by <= 1 ulp on some entries.
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import torch

INT32_PRIME = 105097564  # reference rand_util.py:13
_MAX_INT_32_F = 2147483647.0
_MAX_UINT_32_F = 4294967295.0
_MAX_UINT_32_F_EPS = 3.0


# ======================================================================================
# small pieces
# ======================================================================================


def linspace01(num: int, dtype, device) -> torch.Tensor:
    """``torch.linspace(0, 1, num)`` with the scalar (GPU-style) formula."""
    if num == 1:
        return torch.zeros(1, dtype=dtype, device=device)
    step = torch.ones((), dtype=dtype, device=device) / (num - 1)
    idx = torch.arange(num, device=device)
    lo = step * idx
    hi = 1.0 - step * (num - 1 - idx)
    return torch.where(idx < num // 2, lo, hi)


def ray_depths(near, far, num_samples: int, num_samples_inf: int, disparity_at_inf: float):
    """Sample depths ``[N, S + S_inf]`` (naive_renderer.py:218-219, 239-247, 810-813)."""
    lsp = linspace01(num_samples, near.dtype, near.device)
    depths = near[:, None] + lsp[None, :] * (far - near)[:, None]
    if num_samples_inf > 0:
        cols = []
        for k in range(num_samples_inf):
            frac = (k + 1) / num_samples_inf
            n_disp = (disparity_at_inf - 1) * frac + 1
            cols.append(far * (1 / n_disp))
        depths = torch.cat([depths, torch.stack(cols, dim=-1)], dim=-1)
    return depths


def ray_deltas(near, far, depths, num_samples: int):
    """Interval lengths ``[N, S_tot]`` (naive_renderer.py:252-257)."""
    if num_samples > 1:
        first = (far - near) / (num_samples - 1)
    else:
        first = torch.ones_like(near)
    return torch.cat([first[:, None], depths[:, 1:] - depths[:, :-1]], dim=-1)


def contract_pi(p: torch.Tensor) -> torch.Tensor:
    """MeRF contraction followed by the x0.5 rescale (naive_renderer.py:796-807)."""
    a = p.abs()
    n = a.max(dim=-1, keepdim=True).values
    on_max = (a - n).abs() <= 1e-7
    big = torch.where(on_max, (2 - 1 / a) * (p / a), p / n)
    return torch.where(n <= 1.0, p, big) / 2


def in_bounds(p: torch.Tensor) -> torch.Tensor:
    return (p.abs() <= 1.0).all(dim=-1)


def split_mlp(flat: torch.Tensor, n_hidden) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """Flat -> (weights [in,out], biases); all weights first, then all biases
    (mlp_utils.py:691-721)."""
    dims = [int(v) for v in (n_hidden.tolist() if torch.is_tensor(n_hidden) else n_hidden)]
    pairs = list(zip(dims[:-1], dims[1:]))
    ws, bs, pos = [], [], 0
    for i, o in pairs:
        ws.append(flat[pos : pos + i * o].reshape(i, o))
        pos += i * o
    for _, o in pairs:
        bs.append(flat[pos : pos + o])
        pos += o
    assert pos == flat.numel(), f"mlp params: consumed {pos} of {flat.numel()}"
    return ws, bs


def split_decoder(mlp_params, n_hidden_trunk, n_hidden_opacity, n_hidden_color):
    def numel(nh):
        d = [int(v) for v in (nh.tolist() if torch.is_tensor(nh) else nh)]
        return sum(i * o + o for i, o in zip(d[:-1], d[1:]))

    nt, no, nc = numel(n_hidden_trunk), numel(n_hidden_opacity), numel(n_hidden_color)
    assert nt + no + nc == mlp_params.numel()
    return (
        split_mlp(mlp_params[:nt], n_hidden_trunk),
        split_mlp(mlp_params[nt : nt + no], n_hidden_opacity),
        split_mlp(mlp_params[nt + no :], n_hidden_color),
    )


_GEOMETRY_DTYPE = None


class geometry_dtype:
    """Test diagnostics: while active, ``lightplane_renderer_naive`` computes the GEOMETRY of the march -- sample depths, interval
    lengths, sample points, contraction, un-normalised coordinates, cell indices and interpolation weights -- in ``dtype`` (fp32: the
    reference's own arithmetic, which DEFINES which cell a sample falls into and with what weights; the HIP kernels reproduce it,
    tests/test_gpu_parity.py::test_corner_indices_bit_exact) whatever the dtype of the grids / parameters / ray encoding.  An fp64
    oracle under it is "the reference's geometry, a wide decoder": what separates it from a kernel is the decoder's arithmetic alone,
    not the 2^-24 x grid-extent round-off of a coordinate (1e-5 of a cell on a 128-cell axis -- as large as a ReLU near tie).  No
    effect on any result outside the context."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _GEOMETRY_DTYPE
        self._prev, _GEOMETRY_DTYPE = _GEOMETRY_DTYPE, self.dtype
        return self

    def __exit__(self, *exc):
        global _GEOMETRY_DTYPE
        _GEOMETRY_DTYPE = self._prev


_RELU_RECORDER = None


class relu_margin_recorder:
    """Test diagnostics (no effect on any result): while active, every ReLU of the decoder reports how close its
    pre-activations come to zero.  ``margin`` = per leading index (``[R, S]`` for the Renderer) the minimum over all ReLU
    sites and non-zero units of |pre-activation| / max |pre-activation of that site|.  The gradient of a ReLU network is
    discontinuous exactly where a pre-activation is zero: a sample whose margin is below the round-off of an fp32 dot
    product is where two correct implementations may legitimately take different branches (tests/test_gpu_parity.py
    ``TieMasks``)."""

    def __init__(self):
        self.margin = None

    def __enter__(self):
        global _RELU_RECORDER
        self._prev, _RELU_RECORDER = _RELU_RECORDER, self
        return self

    def __exit__(self, *exc):
        global _RELU_RECORDER
        _RELU_RECORDER = self._prev

    def update(self, x):
        with torch.no_grad():
            ax = x.detach().abs()
            # (an EXACT zero -- a masked / out-of-range sample, a dead input -- is no tie: every implementation gets exactly 0)
            m = torch.where(ax == 0, torch.full_like(ax, float("inf")), ax).amin(dim=-1) / ax.max().clamp(min=1e-30)
            self.margin = m if self.margin is None else torch.minimum(self.margin, m)


_RELU_FORCER = None


class relu_mask_forcer:
    """Test diagnostics: while active, the k-th ReLU site the decoder evaluates (call order: every hidden ReLU of the trunk, the
    trunk output, the opacity head's hidden ReLUs, the colour head's -- naive_renderer.py:328-501 as restated in ``eval_decoder``)
    takes its branch from ``masks[k]`` (bool, same shape as the pre-activation; entries of ``keep[k]`` that are False fall back to
    the sign of the pre-activation): ``y = x * mask``.  With the decisions a GPU backward really took (the DUMP twins of the HIP
    kernels, include/lightplane_hip.h ``lp_renderer_backward_relu_dump``) forced onto it, the oracle differentiates the SAME
    piecewise-linear branch: a near-tie unit the kernel resolved the other way stops being a difference of O(1) in the
    gradient and becomes one of O(|pre-activation|) in the output (tests: ``test_flips_are_flips``)."""

    def __init__(self, masks, keep=None, near_eps=None):
        self.masks, self.keep, self.k, self.n_forced = masks, keep, 0, 0
        # what was forced, measured in THIS (the forced) evaluation: the largest |pre-activation| of a forced unit relative to its
        # site's largest (a unit forced against a clear sign is not a near tie -- the caller's assertion), and, with ``near_eps``,
        # how many units of the kept samples sit within that relative distance of zero (the pool forced units may come from)
        self.near_eps, self.max_forced_margin, self.n_near_units, self.n_units = near_eps, 0.0, 0, 0

    def __enter__(self):
        global _RELU_FORCER
        self._prev, _RELU_FORCER = _RELU_FORCER, self
        self.k = 0
        return self

    def __exit__(self, *exc):
        global _RELU_FORCER
        _RELU_FORCER = self._prev

    def apply(self, x):
        m = self.masks[self.k]
        assert m.shape == x.shape, (self.k, tuple(m.shape), tuple(x.shape))
        xd = x.detach()
        own = xd > 0
        if self.keep is not None:
            m = torch.where(self.keep[self.k], m, own)
        forced = m != own
        nf = int(forced.sum())
        self.n_forced += nf
        ax = xd.abs()
        scale = float(ax.max().clamp(min=1e-30))
        if nf:
            self.max_forced_margin = max(self.max_forced_margin, float(ax[forced].max()) / scale)
        if self.near_eps is not None:
            kept = torch.ones_like(own) if self.keep is None else self.keep[self.k].expand_as(own)
            self.n_near_units += int((kept & (ax != 0) & (ax < self.near_eps * scale)).sum())
            self.n_units += int(kept.sum())
        self.k += 1
        return x * m.to(x.dtype)


def _relu(x):
    if _RELU_RECORDER is not None:
        _RELU_RECORDER.update(x)
    if _RELU_FORCER is not None:
        return _RELU_FORCER.apply(x)
    return torch.relu(x)


def mlp_forward(x, weights, biases):
    """ReLU between layers, last layer linear (naive_renderer.py:758-776)."""
    for li, (w, b) in enumerate(zip(weights, biases)):
        x = x @ w + b
        if li < len(weights) - 1:
            x = _relu(x)
    return x


# ======================================================================================
# hash RNG (rand_util.py:110-145) -- integer part is exact int32 wrap-around arithmetic
# ======================================================================================


def _wrap32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


def _hash32(x: torch.Tensor) -> torch.Tensor:
    x = ((x >> 16) ^ x) * 0x45D9F3B
    x = ((x >> 16) ^ x) * 0x45D9F3B
    return (x >> 16) ^ x


def _pair_hash_scalar(x: int, h: int) -> int:
    h = h ^ x
    return (h << 24) + h * 0x193


def _pair_hash(x_scalar: int, h: torch.Tensor) -> torch.Tensor:
    h = h ^ _wrap32(x_scalar)
    return (h << 24) + h * 0x193


def hash_uniform_pair(i1: torch.Tensor, i2: torch.Tensor, seed: int):
    """The two int32 hashes feeding Box-Muller (exposed for bit-exact checks)."""
    h1 = _pair_hash(_pair_hash_scalar(INT32_PRIME, seed), _hash32(i1.to(torch.int32)))
    h2 = _pair_hash(_pair_hash_scalar(INT32_PRIME, seed + 1), _hash32(i2.to(torch.int32)))
    return h1, h2


def int_to_randn(i1: torch.Tensor, i2: torch.Tensor, seed: int) -> torch.Tensor:
    h1, h2 = hash_uniform_pair(i1, i2, seed)
    denom = _MAX_UINT_32_F + _MAX_UINT_32_F_EPS
    u1 = (h1 + _MAX_INT_32_F + _MAX_UINT_32_F_EPS) / denom
    u2 = (h2 + _MAX_INT_32_F + _MAX_UINT_32_F_EPS) / denom
    return (-2 * u1.log()).sqrt() * (6.28318530718 * u2).cos()


def sample_noise(num_rays: int, tot_samples: int, seed: int, device) -> torch.Tensor:
    """Per-(ray, sample) standard normal noise (naive_renderer.py:779-793)."""
    pad = max(num_rays, 16)
    i1 = (
        tot_samples * torch.arange(num_rays, device=device)[:, None]
        + torch.arange(tot_samples, device=device)[None]
        + 1
    ).long()
    i2 = i1 + pad * tot_samples
    return int_to_randn(i1.reshape(-1), i2.reshape(-1), seed).reshape(num_rays, tot_samples)


# ======================================================================================
# grid-list sampling (Renderer side)
# ======================================================================================


def _grid_axes(shape) -> Tuple[str, Tuple[int, ...]]:
    """('voxel', (0,1,2)) or ('plane', (axis_a, axis_b)) with axes as xyz indices
    ordered (W-axis, H-axis[, D-axis]) of the *sampled* tensor."""
    _, D, H, W, _ = shape
    nonsing = [int(s > 1) for s in (D, H, W)]
    if sum(nonsing) == 3:
        return "voxel", (0, 1, 2)
    if sum(nonsing) == 2:
        if D == 1:
            return "plane", (0, 1)  # xy : x->W, y->H
        if H == 1:
            return "plane", (0, 2)  # xz : x->W, z->D
        return "plane", (1, 2)  # yz : y->H, z->D
    raise ValueError(f"Unexpected n non-singular dim of input grid ({sum(nonsing)})")


def _unnormalize_renderer(c: torch.Tensor, size: int) -> torch.Tensor:
    return ((c + 1) * size - 1) / 2


def _unnormalize_splatter(c: torch.Tensor, size: int) -> torch.Tensor:
    return (c + 1.0) / 2.0 * size - 0.5


def _axis_sizes(shape, kind, axes):
    """Sizes along the sampled axes, as dict xyz-index -> size."""
    _, D, H, W, _ = shape
    return {0: W, 1: H, 2: D}


def _corner_setup(points, shape, unnormalize):
    """Per grid: list of (flat_spatial_index [R,S] long, weight [R,S], valid [R,S] bool).

    flat_spatial_index is ((z*H + y)*W + x) with out-of-range coordinates clamped
    (weight is zeroed through ``valid`` instead).
    """
    _, D, H, W, _ = shape
    kind, axes = _grid_axes(shape)
    size = {0: W, 1: H, 2: D}
    lo, frac_lo, frac_hi = {}, {}, {}
    for ax in axes:
        t = unnormalize(points[..., ax], size[ax])
        f = torch.floor(t)
        lo[ax] = f
        frac_hi[ax] = t - f  # weight of the upper corner
        frac_lo[ax] = (f + 1) - t  # weight of the lower corner
    corners = []
    n_ax = len(axes)
    for bits in range(1 << n_ax):
        w = None
        valid = None
        coord = {0: None, 1: None, 2: None}
        for k, ax in enumerate(axes):
            up = (bits >> k) & 1
            c = lo[ax] + up
            wk = frac_hi[ax] if up else frac_lo[ax]
            vk = (c >= 0) & (c <= size[ax] - 1)
            w = wk if w is None else w * wk
            valid = vk if valid is None else (valid & vk)
            coord[ax] = c.clamp(0, size[ax] - 1).long()
        zero = torch.zeros_like(next(v for v in coord.values() if v is not None))
        x = coord[0] if coord[0] is not None else zero
        y = coord[1] if coord[1] is not None else zero
        z = coord[2] if coord[2] is not None else zero
        corners.append(((z * H + y) * W + x, w, valid))
    return corners


def sample_grid_list(grids, points, grid_idx, mask_out_of_bounds: bool, unnormalize=_unnormalize_renderer):
    """Sum over the grid-list of tri/bi-linear samples -> ``[R, S, C]``
    (naive_renderer.py:625-731; ``align_corners=False``, zero padding)."""
    out = None
    gi = grid_idx.long()
    for g in grids:
        B, D, H, W, C = g.shape
        flat = g.reshape(B * D * H * W, C)
        acc = None
        for idx, w, valid in _corner_setup(points, g.shape, unnormalize):
            rows = gi[:, None] * (D * H * W) + idx
            v = flat[rows] * (w * valid.to(w.dtype))[..., None]
            acc = v if acc is None else acc + v
        if mask_out_of_bounds:
            acc = acc * in_bounds(points).to(acc.dtype)[..., None]
        out = acc if out is None else out + acc
    return out


def sample_scaffold_nearest(scaffold, points, grid_idx):
    """Nearest-neighbour occupancy lookup, zero outside, times the in-bounds mask
    (naive_renderer.py:484-492 -> F.grid_sample(mode='nearest'): round-half-even)."""
    B, D, H, W = scaffold.shape
    flat = scaffold.reshape(B * D * H * W)
    ix = torch.round(_unnormalize_renderer(points[..., 0], W))
    iy = torch.round(_unnormalize_renderer(points[..., 1], H))
    iz = torch.round(_unnormalize_renderer(points[..., 2], D))
    valid = (ix >= 0) & (ix <= W - 1) & (iy >= 0) & (iy <= H - 1) & (iz >= 0) & (iz <= D - 1)
    idx = (iz.clamp(0, D - 1).long() * H + iy.clamp(0, H - 1).long()) * W + ix.clamp(0, W - 1).long()
    rows = grid_idx.long()[:, None] * (D * H * W) + idx
    val = flat[rows] * valid.to(flat.dtype)
    return val * in_bounds(points).to(flat.dtype)


def renderer_corner_indices(rays, grid_sizes, num_samples, num_samples_inf=0, contract_coords=False,
                            disparity_at_inf=1e-5):
    """Integer bookkeeping of the Renderer march, for bit-exact index parity.

    Returns per grid a ``[R, S_tot, K]`` int64 tensor of flat *row* indices into
    that grid (batch offset included; -1 where the corner is out of range),
    K = 8 (voxel) or 4 (plane), corner order = bit k of the corner id selects
    the upper neighbour along the k-th sampled axis (x, y, z order).
    """
    depths = ray_depths(rays.near, rays.far, num_samples, num_samples_inf, disparity_at_inf)
    points = depths[..., None] * rays.directions[:, None] + rays.origins[:, None]
    if contract_coords:
        points = contract_pi(points)
    out = []
    gi = rays.grid_idx.long()
    for gs in grid_sizes:
        B, D, H, W, C = [int(v) for v in gs]
        cols = []
        for idx, _, valid in _corner_setup(points, (B, D, H, W, C), _unnormalize_renderer):
            rows = gi[:, None] * (D * H * W) + idx
            cols.append(torch.where(valid, rows, torch.full_like(rows, -1)))
        out.append(torch.stack(cols, dim=-1))
    return out


# ======================================================================================
# Renderer
# ======================================================================================


def _as_grid_list(grid, grid_sizes):
    if torch.is_tensor(grid):
        assert grid_sizes is not None
        sizes = [[int(v) for v in gs] for gs in (grid_sizes.tolist() if torch.is_tensor(grid_sizes) else grid_sizes)]
        rows = [s[0] * s[1] * s[2] * s[3] for s in sizes]
        return [p.reshape(*s) for p, s in zip(grid.split(rows, dim=0), sizes)]
    return list(grid)


def eval_decoder(points, grids, grid_idx, decoder_params, rays_encoding, gain,
                 mask_out_of_bounds_samples=False, noise=None, scaffold=None, color_grids=None,
                 contract_coords=False):
    """Opacity ``[R,S]`` and colour ``[R,S,Cc]`` at ``points`` (naive_renderer.py:328-501)."""
    (wt, bt), (wo, bo), (wc, bc) = split_decoder(
        decoder_params.mlp_params,
        decoder_params.n_hidden_trunk,
        decoder_params.n_hidden_opacity,
        decoder_params.n_hidden_color,
    )
    if contract_coords:
        points = contract_pi(points)
    feat = sample_grid_list(grids, points, grid_idx, mask_out_of_bounds_samples)
    if color_grids is None:
        trunk = _relu(mlp_forward(feat, wt, bt))
        opacity_raw = mlp_forward(trunk, wo, bo)
        color_raw = mlp_forward(trunk + rays_encoding[:, None], wc, bc)
    else:
        assert len(wt) == 0
        cfeat = sample_grid_list(color_grids, points, grid_idx, mask_out_of_bounds_samples)
        opacity_raw = mlp_forward(_relu(feat), wo, bo)
        color_raw = mlp_forward(_relu(cfeat) + rays_encoding[:, None], wc, bc)
    assert opacity_raw.shape[-1] == 1
    opacity_raw = opacity_raw[..., 0]
    if noise is not None:
        opacity_raw = opacity_raw + noise
    opacity = gain * torch.nn.functional.softplus(opacity_raw)
    color = torch.sigmoid(color_raw)
    if scaffold is not None:
        occ = sample_scaffold_nearest(scaffold, points, grid_idx)
        opacity = opacity * occ
        color = color * occ[..., None]
    return opacity, color


def lightplane_renderer_naive(
    rays,
    grid,
    decoder_params,
    num_samples: int,
    gain: float,
    mask_out_of_bounds_samples: bool = False,
    num_samples_inf: int = 0,
    contract_coords: bool = False,
    inject_noise_sigma: float = 0.0,
    inject_noise_seed: Optional[int] = None,
    disparity_at_inf: float = 1e-5,
    scaffold: Optional[torch.Tensor] = None,
    color_grid=None,
    grid_sizes=None,
    color_grid_sizes=None,
    **_ignored,
):
    """Oracle Renderer; same call signature and returns as the reference's
    ``lightplane_renderer_naive`` (naive_renderer.py:39-60):
    ``(ray_length_render [N], negative_log_transmittance [N], feature_render [N, color_chn])``."""
    grids = _as_grid_list(grid, grid_sizes)
    color_grids = None if color_grid is None else _as_grid_list(color_grid, color_grid_sizes)
    n_rays = rays.directions.shape[0]
    tot = num_samples + num_samples_inf
    near, far, directions, origins = rays.near, rays.far, rays.directions, rays.origins
    gd = _GEOMETRY_DTYPE
    if gd is not None and directions.dtype != gd:  # test diagnostics (geometry_dtype): the march's geometry in its own dtype
        near, far, directions, origins = near.to(gd), far.to(gd), directions.to(gd), origins.to(gd)
    depths = ray_depths(near, far, num_samples, num_samples_inf, disparity_at_inf)
    noise = None
    if inject_noise_sigma > 0.0:
        seed = 0 if inject_noise_seed is None else int(inject_noise_seed)
        noise = sample_noise(n_rays, tot, seed, rays.directions.device).to(rays.directions.dtype) * inject_noise_sigma
    points = depths[..., None] * directions[:, None] + origins[:, None]  # (stays in the geometry dtype: cells and weights come from it)
    delta = ray_deltas(near, far, depths, num_samples).to(rays.directions.dtype)
    depths = depths.to(rays.directions.dtype)
    opacity, color = eval_decoder(
        points, grids, rays.grid_idx, decoder_params, rays.encoding, gain,
        mask_out_of_bounds_samples=mask_out_of_bounds_samples, noise=noise, scaffold=scaffold,
        color_grids=color_grids, contract_coords=contract_coords,
    )
    nlt = torch.cumsum(torch.nn.functional.pad(opacity * delta, (1, 0)), dim=-1)
    transmittance = torch.exp(-nlt)
    w = transmittance[:, :-1] - transmittance[:, 1:]
    ray_length = (depths * w).sum(dim=-1)
    feature = (color * w[..., None]).sum(dim=-2)
    feature = feature[..., : decoder_params.color_chn]
    return ray_length, nlt[:, -1], feature


# ======================================================================================
# Splatter
# ======================================================================================


def _splat_one_grid(out_flat, shape, points, grid_idx, feature, ray_mask):
    """scatter-add ``feature [R,S,C]`` into ``out_flat [B*D*H*W, C]`` (functional)."""
    B, D, H, W, C = shape
    gi = grid_idx.long()
    for idx, w, valid in _corner_setup(points, shape, _unnormalize_splatter):
        rows = (gi[:, None] * (D * H * W) + idx).reshape(-1)
        val = feature * (w * valid.to(w.dtype) * ray_mask)[..., None]
        out_flat = out_flat.index_add(0, rows, val.reshape(-1, val.shape[-1]))
    return out_flat


def _splatter_impl(rays, output_grid_size, num_samples, num_samples_inf, mask_out_of_bounds_samples,
                   contract_coords, disparity_at_inf, return_list, mlp_params=None, input_grid=None,
                   input_grid_sizes=None):
    sizes = [[int(v) for v in gs] for gs in (output_grid_size.tolist() if torch.is_tensor(output_grid_size) else output_grid_size)]
    dtype, device = rays.directions.dtype, rays.directions.device
    depths = ray_depths(rays.near, rays.far, num_samples, num_samples_inf, disparity_at_inf)
    points = depths[..., None] * rays.directions[:, None] + rays.origins[:, None]
    if contract_coords:
        points = contract_pi(points)
    tot = depths.shape[1]
    feat = rays.encoding[:, None, :].expand(-1, tot, -1)
    if mlp_params is not None:
        in_grids = _as_grid_list(input_grid, input_grid_sizes)
        ws, bs = split_mlp(mlp_params.mlp_params, mlp_params.n_hidden)
        sampled = sample_grid_list(in_grids, points, rays.grid_idx, mask_out_of_bounds_samples)
        feat = mlp_forward(sampled + feat, ws, bs)
    if mask_out_of_bounds_samples:
        ray_mask = in_bounds(points).to(dtype)
    else:
        ray_mask = torch.ones(points.shape[:-1], dtype=dtype, device=device)
    ones = torch.ones(points.shape[:-1] + (1,), dtype=dtype, device=device)
    out = []
    for gs in sizes:
        B, D, H, W, C = gs
        fgrid = torch.zeros(B * D * H * W, C, dtype=dtype, device=device)
        wgrid = torch.zeros(B * D * H * W, 1, dtype=dtype, device=device)
        fgrid = _splat_one_grid(fgrid, gs, points, rays.grid_idx, feat, ray_mask)
        wgrid = _splat_one_grid(wgrid, gs, points, rays.grid_idx, ones, ray_mask)
        out.append((fgrid / wgrid.clamp(min=1e-5)).reshape(B, D, H, W, C))
    if return_list:
        return out
    return torch.cat([g.reshape(-1, g.shape[-1]) for g in out], dim=0)


def lightplane_splatter_naive(rays, output_grid_size, num_samples, num_samples_inf=0,
                              mask_out_of_bounds_samples=False, contract_coords=False,
                              disparity_at_inf=1e-5, return_list=True, **_ignored):
    """Oracle Splatter (reference naive_splatter.py:41-103)."""
    return _splatter_impl(rays, output_grid_size, num_samples, num_samples_inf,
                          mask_out_of_bounds_samples, contract_coords, disparity_at_inf, return_list)


def lightplane_mlp_splatter_naive(rays, output_grid_size, mlp_params, input_grid, num_samples,
                                  num_samples_inf=0, mask_out_of_bounds_samples=False,
                                  contract_coords=False, disparity_at_inf=1e-5, input_grid_sizes=None,
                                  return_list=True, **_ignored):
    """Oracle MLP-Splatter (reference naive_splatter.py:106-182)."""
    return _splatter_impl(rays, output_grid_size, num_samples, num_samples_inf,
                          mask_out_of_bounds_samples, contract_coords, disparity_at_inf, return_list,
                          mlp_params=mlp_params, input_grid=input_grid, input_grid_sizes=input_grid_sizes)
