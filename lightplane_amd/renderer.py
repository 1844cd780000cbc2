"""Functional Lightplane Renderer on MI355X: ``lightplane_renderer`` + its autograd boundary.

Drop-in for the reference's ``lightplane/lightplane_renderer.py``
(`lightplane_renderer` :33-293, `LightplaneFunction` :296-756): same arguments, same
returns ``(ray_length_render [N], negative_log_transmittances [N], feature_render
[N, color_chn])``, same non-differentiable inputs (ray geometry, grid_idx, scaffold).
The two Triton launches (:505-555 forward, :657-711 backward) are replaced by
``lp_renderer_forward`` / ``lp_renderer_backward`` of ``liblightplane_hip.so``.

Differences that are deliberate:
* no ray padding, no power-of-two / >=16 channel restriction -- the kernels mask tails;
* grid sizes travel by value (no ``.item()`` sync); the ``grid_idx`` range check is the only
  device sync and can be switched off with ``lightplane_amd.config.check_inputs = False``;
* besides the final ``-log T`` the forward saves it every 32 samples (``[N, ceil(S/32)]``,
  still O(N)) so that the backward's far->near transmittance reconstruction cannot drift;
* the post-backward ``isfinite`` asserts (:719-722) are opt-in
  (``lightplane_amd.config.check_finite_grads``);
* ``triton_block_size`` / ``triton_num_warps`` / ``regenerate_code`` are accepted and ignored.
"""
from __future__ import annotations

import ctypes
import math
import random
import warnings
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib, config
from .grids import (
    GridDesc,
    check_grid_and_color_grid,
    make_grid_descs,
    process_and_flatten_grid,
)
from .params import DecoderParams, int_list_of, mlp_numel
from .rays import Rays


@dataclass
class _RendererCfg:
    descs: List[GridDesc]
    channels: int
    n_rows: int
    color_descs: Optional[List[GridDesc]]
    color_n_rows: int
    dims_trunk: List[int]
    dims_opacity: List[int]
    dims_color: List[int]
    color_chn: int
    num_samples: int
    num_samples_inf: int
    gain: float
    mask_out_of_bounds_samples: bool
    contract_coords: bool
    disparity_at_inf: float
    noise_sigma: float
    noise_seed: int
    scaffold_shape: Optional[Tuple[int, int, int, int]]
    kernel: int
    stop_neg_log_t: float = 0.0
    n_grid_tensors: int = 1        # tensors the grid-list arrives in: 1 flat tensor, or one per grid (zero-copy list)
    n_color_tensors: int = 0
    grid_is_list: bool = False
    alpha_mode: int = 0            # fused module epilogue: 0 none, 1 alpha = 1 - T, 2 log T
    arithmetic: int = 0            # LP_ARITH_* of the backward (include/lightplane_hip.h)
    march_order: int = 0           # LP_MARCH_* of the backward
    row_length: int = 0            # LpRays.row_length: rays per image row (0 = unknown)


def _fill_args(cfg: _RendererCfg, grids, color_grids, mlp_params, directions, origins, grid_idx, near, far,
               encoding, scaffold) -> _lib.LpRendererArgs:
    """``grids`` / ``color_grids``: tuples of tensors -- ONE flat ``[rows, C]`` tensor, or one ``[B, D, H, W, C]`` tensor
    per grid (zero-copy grid-list: per-grid base pointers in the ABI)."""
    a = _lib.LpRendererArgs()
    a.rays = _lib.make_rays(directions, origins, grid_idx, near, far, encoding, cfg.row_length)
    a.grid = _lib.make_grid_list(list(grids) if cfg.grid_is_list else grids[0], cfg.descs, cfg.channels, cfg.n_rows)
    if cfg.color_descs is not None:
        a.color_grid = _lib.make_grid_list(list(color_grids) if cfg.grid_is_list else color_grids[0], cfg.color_descs,
                                           cfg.channels, cfg.color_n_rows)
    else:
        a.color_grid = _lib.make_grid_list(None, [], 0, 0)
    if scaffold is not None:
        a.scaffold = _lib.ptr(scaffold)
        B, D, H, W = cfg.scaffold_shape
        a.scaffold_shape = _lib.LpGrid(B, D, H, W, 0, None)
    a.march = _lib.make_march(cfg.num_samples, cfg.num_samples_inf, cfg.mask_out_of_bounds_samples,
                              cfg.contract_coords, cfg.disparity_at_inf)
    a.mlp_params = _lib.ptr(mlp_params)
    a.n_mlp_params = mlp_params.numel()
    n_t, n_o = mlp_numel(cfg.dims_trunk), mlp_numel(cfg.dims_opacity)
    a.trunk = _lib.make_mlp(cfg.dims_trunk, 0)
    a.opacity = _lib.make_mlp(cfg.dims_opacity, n_t)
    a.color = _lib.make_mlp(cfg.dims_color, n_t + n_o)
    a.color_chn = cfg.color_chn
    a.gain = float(cfg.gain)
    a.noise_sigma = float(cfg.noise_sigma)
    a.noise_seed = ctypes.c_int32(int(cfg.noise_seed) & 0xFFFFFFFF).value
    a.kernel = cfg.kernel
    a.stop_neg_log_t = float(cfg.stop_neg_log_t)
    a.alpha_mode = int(cfg.alpha_mode)
    a.arithmetic = int(cfg.arithmetic)
    a.march_order = int(cfg.march_order)
    return a


_warned_shapes = set()


def _warn_if_generic(a, cfg: _RendererCfg) -> None:
    if not config.warn_generic_kernel or cfg.kernel != _lib.LP_KERNEL_AUTO or cfg.arithmetic != _lib.LP_ARITH_DEFAULT:
        return  # (LP_ARITH_FP32 outside the tuned family asks for the generic fp32 kernels by definition)
    key = (cfg.channels, tuple(cfg.dims_trunk), tuple(cfg.dims_opacity), tuple(cfg.dims_color), cfg.color_chn,
           cfg.color_descs is not None)
    if key in _warned_shapes:
        return
    if _lib.lib().lp_renderer_kernel_family(ctypes.byref(a)) == 0:
        _warned_shapes.add(key)
        warnings.warn(
            "lightplane_amd: this decoder shape runs on the shape-generic Renderer kernels (10-100x slower). The "
            "MFMA families cover grid channels 16/32, grid-lists below 2^31 rows, trunk 1-4 (0 with a separate colour grid) / "
            "opacity 1-4 / colour 1-4 layers with ONE hidden width of 16 or 32 and <= 32 colour channels, or up to 2/2/2 "
            "layers with hidden width 64 and / or 64 grid channels and <= 4 colour channels (a separate colour grid only with 16 / 32 "
            "grid channels); at most 256 "
            "beyond-far samples. "
            f"Got channels={cfg.channels}, trunk={cfg.dims_trunk}, opacity={cfg.dims_opacity}, "
            f"color={cfg.dims_color}, color_chn={cfg.color_chn}, separate colour grid={cfg.color_descs is not None}.")


def _shape_args(grid, decoder_params: DecoderParams, grid_sizes=None, color_grid=None, color_grid_sizes=None):
    """``LpRendererArgs`` carrying shapes only (no pointers) -- what the library's selection queries look at."""
    if isinstance(grid, (list, tuple)):
        grid_sizes = [list(g.shape) for g in grid]
    descs, channels, n_rows = make_grid_descs(grid_sizes)
    color_descs, color_n_rows = None, 0
    if color_grid is not None:
        if isinstance(color_grid, (list, tuple)):
            color_grid_sizes = [list(g.shape) for g in color_grid]
        color_descs, _, color_n_rows = make_grid_descs(color_grid_sizes)
    dims_t, dims_o, dims_c = _decoder_dims(decoder_params)
    a = _lib.LpRendererArgs()
    a.grid = _lib.make_grid_list(None, descs, channels, n_rows)
    a.color_grid = _lib.make_grid_list(None, color_descs or [], channels if color_descs else 0, color_n_rows)
    n_t, n_o = mlp_numel(dims_t), mlp_numel(dims_o)
    a.trunk, a.opacity, a.color = _lib.make_mlp(dims_t, 0), _lib.make_mlp(dims_o, n_t), _lib.make_mlp(dims_c, n_t + n_o)
    a.color_chn = int(decoder_params.color_chn)
    return a


#: keyword arguments of ``lightplane_renderer`` that ``kernel_family`` / ``backward_segments`` accept and ignore (they do not
#: influence the shape-only answer); anything else raises, so that a typo such as ``num_sample_inf=`` cannot pass silently
_RENDER_KWARGS = frozenset((
    "num_samples", "gain", "mask_out_of_bounds_samples", "contract_coords", "disparity_at_inf", "inject_noise_sigma",
    "inject_noise_seed", "scaffold", "stop_transmittance", "regenerate_code", "triton_block_size", "triton_num_warps",
    "allow_unsupported", "checkpointing", "use_naive_impl", "march_order"))


def _check_render_kwargs(fn: str, kw) -> None:
    unknown = sorted(set(kw) - _RENDER_KWARGS)
    if unknown:
        raise TypeError(f"{fn}() got unexpected keyword argument(s) {unknown}: not a keyword of lightplane_renderer")


def kernel_family(rays: Rays, grid, decoder_params: DecoderParams, grid_sizes=None, color_grid=None,
                  color_grid_sizes=None, num_samples_inf: int = 0, kernel: int = _lib.LP_KERNEL_AUTO,
                  arithmetic: Optional[int] = None, **_unused) -> int:
    """Kernel family that runs these shapes: 0 generic, 1 the tuned MFMA kernels of the default decoder (2/2/2 x 32), 3
    layer-looped MFMA (1-4 layers per MLP, widths 16 / 32 / 64) (``lp_renderer_kernel_family``; needs no GPU; 2, the fp32-MFMA
    hidden-64 family, was retired in 0.2.4).  ``num_samples_inf``: more than 256 beyond-far samples run the generic kernels;
    ``kernel=LP_KERNEL_GENERIC`` forces family 0; ``arithmetic=LP_ARITH_FP32`` (default ``config.arithmetic``) leaves the tuned
    family where it has such instantiations and family 0 elsewhere.  The other keywords of the render call (``_RENDER_KWARGS``) are
    accepted and ignored, an unknown keyword raises."""
    _check_render_kwargs("kernel_family", _unused)
    if int(kernel) == _lib.LP_KERNEL_GENERIC:
        return 0
    a = _shape_args(grid, decoder_params, grid_sizes, color_grid, color_grid_sizes)
    a.march.num_samples_inf = int(num_samples_inf)
    a.arithmetic = int(config.arithmetic if arithmetic is None else arithmetic)
    return int(_lib.lib().lp_renderer_kernel_family(ctypes.byref(a)))


def relu_dump_words(rays: Rays, grid, decoder_params: DecoderParams, grid_sizes=None, color_grid=None, color_grid_sizes=None,
                    num_samples_inf: int = 0, kernel: int = _lib.LP_KERNEL_AUTO, arithmetic: Optional[int] = None, **_unused) -> int:
    """Words per (ray, sample) of the ReLU dump of these shapes (``lp_renderer_relu_dump_words``; test hook, needs no GPU), or 0
    where the kernel that would run has no DUMP twin (LP_ARITH_FP32, the tuned family's eight-wave workgroups, a library built
    without -DLP_TEST_HOOKS)."""
    _check_render_kwargs("relu_dump_words", _unused)
    if not int(_lib.build_info().get("test_hooks", 0)):
        return 0
    a = _shape_args(grid, decoder_params, grid_sizes, color_grid, color_grid_sizes)
    a.march.num_samples_inf = int(num_samples_inf)
    a.kernel = int(kernel)
    a.arithmetic = int(config.arithmetic if arithmetic is None else arithmetic)
    w = int(_lib.lib().lp_renderer_relu_dump_words(ctypes.byref(a)))
    return max(w, 0)


def backward_segments(rays: Rays, grid, decoder_params: DecoderParams, num_samples: int, num_samples_inf: int = 0,
                      grid_sizes=None, color_grid=None, color_grid_sizes=None, stop_transmittance: float = 0.0,
                      kernel: int = _lib.LP_KERNEL_AUTO, arithmetic: Optional[int] = None, march_order: str = "rays", **_unused) -> int:
    """Number of ray segments the backward of this call is split into (``lp_renderer_backward_segments``; needs no
    GPU): 1 = one sweep per ray, > 1 = small batch, the number of state records per ray (one per LP_SEG_LEN = 8 samples; a workgroup sweeps one or two such blocks).  ``march_order``
    "samples" (the transposed march, where it applies) never segments: it deals a small batch over the chip by rays per wave."""
    _unused.pop("march_order", None)
    _check_render_kwargs("backward_segments", _unused)
    a = _shape_args(grid, decoder_params, grid_sizes, color_grid, color_grid_sizes)
    a.rays.n_rays = int(rays.directions.shape[0])
    a.march.num_samples, a.march.num_samples_inf = int(num_samples), int(num_samples_inf)
    a.stop_neg_log_t = -math.log(stop_transmittance) if stop_transmittance and stop_transmittance > 0 else 0.0
    a.kernel = int(kernel)
    a.arithmetic = int(config.arithmetic if arithmetic is None else arithmetic)
    a.march_order = _lib.LP_MARCH_SAMPLES_PER_WAVE if march_order == "samples" else _lib.LP_MARCH_RAYS_PER_WAVE
    return int(_lib.lib().lp_renderer_backward_segments(ctypes.byref(a)))


_RELU_DUMP = None


class relu_dump_recorder:
    """Test hook (``lp_renderer_backward_relu_dump``, include/lightplane_hip.h): inside the context every Renderer backward runs
    the DUMP twin of its kernel and leaves the ReLU decisions it took in ``.dump`` -- int32 ``[n_rays, S_tot, W]``,
    ``W = lp_renderer_relu_dump_words``: per ReLU site of the decoder, in the reference's evaluation order, ``words_per_site`` words
    (bit f of word b = unit 32 b + f active), then one flag word: 1 = the sample contributed, 2 = visited but beyond the ray's last
    marched sample, 0 = never visited.  The tuned family: 4 sites x 1 word + flag = 5; the shape-generic kernels: ceil(widest site / 32)
    words per site.  Raises where there is no dump twin (LP_ARITH_FP32, the tuned family's eight-wave workgroups) and for a library
    built without -DLP_TEST_HOOKS."""

    def __init__(self):
        self.dump = None
        self.words_per_site = 1

    def reset(self, n_rays, s_tot, words, dev):
        self.dump = torch.zeros(n_rays, s_tot, words, dtype=torch.int32, device=dev)
        return self.dump

    def __enter__(self):
        global _RELU_DUMP
        self._prev, _RELU_DUMP = _RELU_DUMP, self
        return self

    def __exit__(self, *exc):
        global _RELU_DUMP
        _RELU_DUMP = self._prev


class LightplaneFunction(torch.autograd.Function):
    """Autograd boundary of the Renderer (name kept from the reference, :296).

    The grid-list arrives as ``cfg.n_grid_tensors`` tensors (+ ``cfg.n_color_tensors`` for the colour grid-list): ONE
    flat ``[rows, C]`` tensor, or one tensor PER GRID -- a list input is never concatenated (the reference's
    ``flatten_grid``, misc_utils.py:42-45, copies the whole list every call and splits its gradient every backward):
    the kernels take per-grid base pointers and write per-grid gradient buffers."""

    @staticmethod
    def forward(ctx, cfg: _RendererCfg, mlp_params, encoding, directions, origins, grid_idx, near, far, scaffold,
                bg_color, *grid_tensors):
        grids = tuple(g.contiguous() for g in grid_tensors[: cfg.n_grid_tensors])
        color_grids = tuple(g.contiguous() for g in grid_tensors[cfg.n_grid_tensors:])
        dev = grids[0].device
        stream = _lib.current_stream(dev)
        mlp_params, encoding = mlp_params.contiguous(), encoding.contiguous()
        n = directions.shape[0]
        ray_length = torch.empty(n, device=dev, dtype=torch.float32)
        nlt = torch.empty(n, device=dev, dtype=torch.float32)
        feature = torch.empty(n, cfg.color_chn, device=dev, dtype=torch.float32)
        alpha = torch.empty(n if cfg.alpha_mode else 0, device=dev, dtype=torch.float32)
        a = _fill_args(cfg, grids, color_grids, mlp_params, directions, origins, grid_idx, near, far, encoding,
                       scaffold)
        a.ray_length, a.neg_log_t, a.feature = _lib.ptr(ray_length), _lib.ptr(nlt), _lib.ptr(feature)
        if cfg.alpha_mode:
            a.alpha = _lib.ptr(alpha)
        a.bg_color = _lib.ptr(bg_color)
        # running -log T every LP_NLT_CKPT samples: O(N) state that keeps the backward's
        # transmittance reconstruction exact (the reference saves only the final value, :558-573)
        ckpt = torch.empty(n, _lib.n_nlt_ckpt(cfg.num_samples, cfg.num_samples_inf), device=dev, dtype=torch.float32)
        a.neg_log_t_ckpt = _lib.ptr(ckpt)
        # small batches: the backward sweeps every block of LP_SEG_LEN samples of a ray in its own workgroup, from
        # running sums the forward saves per block (lightplane_hip.h, LpRendererArgs.seg_prefix)
        # ... and the forward marches the segments in parallel as well (config.segment_forward)
        seg = None
        want_bwd_seg = config.segment_backward and any(ctx.needs_input_grad)
        if want_bwd_seg or config.segment_forward:
            n_seg = _lib.lib().lp_renderer_backward_segments(ctypes.byref(a))
            if n_seg > 1:
                seg = torch.empty(n, n_seg, 8, device=dev, dtype=torch.float32)
                a.seg_prefix = _lib.ptr(seg)
                a.seg_forward_off = 0 if config.segment_forward else 1
        if not want_bwd_seg:
            seg_for_backward = None
        else:
            seg_for_backward = seg
        _warn_if_generic(a, cfg)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lp_renderer_forward(ctypes.byref(a), stream), "lp_renderer_forward")
        # O(N) state only: the final -log T (the reference saves the same, :558-573)
        ctx.save_for_backward(nlt, ckpt, seg_for_backward, mlp_params, encoding, directions, origins, grid_idx, near, far, scaffold,
                              bg_color, *grids, *color_grids)
        ctx.cfg = cfg
        # the backward re-uses the filled argument block (it re-reads the pointers from the saved tensors)
        a.ray_length = a.feature = a.alpha = None
        ctx.args = a
        if not cfg.alpha_mode:
            ctx.mark_non_differentiable(alpha)
        return ray_length, nlt, feature, alpha

    @staticmethod
    def backward(ctx, g_len, g_nlt, g_feat, g_alpha):
        (nlt, ckpt, seg, mlp_params, encoding, directions, origins, grid_idx, near, far, scaffold,
         bg_color) = ctx.saved_tensors[:12]
        cfg: _RendererCfg = ctx.cfg
        grids = ctx.saved_tensors[12: 12 + cfg.n_grid_tensors]
        color_grids = ctx.saved_tensors[12 + cfg.n_grid_tensors:]
        dev = grids[0].device
        stream = _lib.current_stream(dev)
        need_params, need_enc = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        need_g = ctx.needs_input_grad[10: 10 + cfg.n_grid_tensors]
        need_c = ctx.needs_input_grad[10 + cfg.n_grid_tensors:]
        # The forward's argument block is re-used: shapes, decoder layout and march are already in it.  Only the pointers
        # are taken again from the saved tensors (they are the forward's tensors unless a saved-tensor hook -- CPU
        # offloading, checkpointing -- unpacked them into new storage).
        # (a private copy of the block: backward calls through one graph may overlap -- multithreaded autograd with
        # retain_graph -- and must not patch each other's pointers)
        a = type(ctx.args).from_buffer_copy(ctx.args)
        r = a.rays
        r.directions, r.origins, r.grid_idx = _lib.ptr(directions), _lib.ptr(origins), _lib.ptr(grid_idx)
        r.near_t, r.far_t, r.encoding = _lib.ptr(near), _lib.ptr(far), _lib.ptr(encoding)
        a.mlp_params, a.scaffold, a.bg_color = _lib.ptr(mlp_params), _lib.ptr(scaffold), _lib.ptr(bg_color)
        a.neg_log_t, a.neg_log_t_ckpt, a.seg_prefix = _lib.ptr(nlt), _lib.ptr(ckpt), _lib.ptr(seg)
        if cfg.grid_is_list:
            for i, g in enumerate(grids):
                a.grid.grids[i].data = _lib.ptr(g)
            for i, g in enumerate(color_grids):
                a.color_grid.grids[i].data = _lib.ptr(g)
        else:
            a.grid.data = _lib.ptr(grids[0])
            if color_grids:
                a.color_grid.data = _lib.ptr(color_grids[0])
        g_len = None if g_len is None else g_len.contiguous()
        g_nlt = None if g_nlt is None else g_nlt.contiguous()
        g_feat = None if g_feat is None else g_feat.contiguous()
        a.grad_ray_length, a.grad_neg_log_t, a.grad_feature = _lib.ptr(g_len), _lib.ptr(g_nlt), _lib.ptr(g_feat)
        g_alpha = g_alpha.contiguous() if (cfg.alpha_mode and g_alpha is not None) else None
        a.grad_alpha = _lib.ptr(g_alpha)
        # the kernels scatter into every grid of a list or into none: allocate all buffers if any grid needs one
        grad_grids = [torch.zeros_like(g) for g in grids] if any(need_g) else None
        grad_cgrids = [torch.zeros_like(g) for g in color_grids] if (color_grids and any(need_c)) else None
        grad_params = torch.zeros_like(mlp_params) if need_params else None
        grad_enc = torch.zeros_like(encoding) if need_enc else None
        a.grad_grid = a.grad_color_grid = None
        for i in range(_lib.LP_MAX_GRIDS):
            a.grad_grid_list[i] = a.grad_color_grid_list[i] = None
        if grad_grids is not None:
            if cfg.grid_is_list:
                _lib.fill_ptr_list(a.grad_grid_list, grad_grids)
            else:
                a.grad_grid = _lib.ptr(grad_grids[0])
        if grad_cgrids is not None:
            if cfg.grid_is_list:
                _lib.fill_ptr_list(a.grad_color_grid_list, grad_cgrids)
            else:
                a.grad_color_grid = _lib.ptr(grad_cgrids[0])
        a.grad_mlp_params, a.grad_encoding = _lib.ptr(grad_params), _lib.ptr(grad_enc)
        with torch.cuda.device(dev):
            if _RELU_DUMP is not None:  # test hook (relu_dump_recorder): the DUMP twin of the same kernel
                words = _lib.lib().lp_renderer_relu_dump_words(ctypes.byref(a))
                if words < 0:
                    _lib.check(words, "lp_renderer_relu_dump_words")
                d = _RELU_DUMP.reset(directions.shape[0], cfg.num_samples + cfg.num_samples_inf, words, dev)
                # ReLU sites in the reference's evaluation order (include/lightplane_hip.h): every kernel family writes the same number
                # of words for each of them, then the flag word
                n_sites = (2 if color_grids else len(cfg.dims_trunk) - 1) + max(len(cfg.dims_opacity) - 2, 0) + max(len(cfg.dims_color) - 2, 0)
                assert n_sites > 0 and (words - 1) % n_sites == 0, (words, n_sites)
                _RELU_DUMP.words_per_site = (words - 1) // n_sites
                _lib.check(_lib.lib().lp_renderer_backward_relu_dump(ctypes.byref(a), d.data_ptr(), d.numel(), stream),
                           "lp_renderer_backward_relu_dump")
            else:
                _lib.check(_lib.lib().lp_renderer_backward(ctypes.byref(a), stream), "lp_renderer_backward")
        if config.check_finite_grads:
            for name, g in [("mlp_params", grad_params), ("encoding", grad_enc)] + \
                    [("grid", g) for g in (grad_grids or [])] + [("color_grid", g) for g in (grad_cgrids or [])]:
                assert g is None or torch.isfinite(g).all(), f"non-finite gradient w.r.t. {name}"
        gg = [None] * cfg.n_grid_tensors if grad_grids is None else [g if nd else None for g, nd in zip(grad_grids, need_g)]
        gc = [None] * cfg.n_color_tensors if grad_cgrids is None else [g if nd else None for g, nd in zip(grad_cgrids, need_c)]
        return (None, grad_params, grad_enc) + (None,) * 7 + tuple(gg) + tuple(gc)


def _neighbours(a, b, da, db):
    """Rays (origin a, direction da) and (b, db) are pixel neighbours: directions within 5 %, origins within 0.05 scene units."""
    return ((da - db).norm(dim=-1) <= 0.05 * da.norm(dim=-1)) & ((a - b).norm(dim=-1) <= 0.05)


def check_inputs_and_plan(rays: Rays, grid_idx: torch.Tensor, B: int, march_order: Optional[str] = None,
                          rays_per_row: Optional[int] = None):
    """The ``grid_idx`` range check (``config.check_inputs``: the one device sync of a call, like the reference's min / max
    asserts, lightplane_renderer.py:464-467) and, riding on the same sync, what the kernels should know about the ORDER of the
    batch.  Returns ``(march order, rays per image row)``:
    * march order for "auto": a batch is image-coherent when most consecutive rays are NEIGHBOURS -- directions within 5 % of each
      other and origins within 0.05 scene units (pinhole rows: same origin, one pixel of angle; orthographic rows: same direction,
      one pixel of offset) -- and marches 32 rays per wavefront; unrelated rays (random rays, but also random PIXELS of one camera:
      same origin, unrelated directions) march 32 samples of one ray (``LP_MARCH_*``);
    * rays per row (``LpRays.row_length``; 0 = unknown) unless given: the neighbour chain of a scanline-ordered image breaks exactly
      at the row ends -- first break at W - 1, N / W - 1 breaks in all -- and ray W is a neighbour of ray 0 (the pixel below).
    With ``check_inputs`` off nothing is looked at: "auto" is "rays", the row length is what the caller says (or unknown)."""
    march_order = config.march_order if march_order is None else march_order
    assert march_order in ("auto", "rays", "samples"), f"march_order has to be 'auto', 'rays' or 'samples' (got {march_order!r})"
    march = _lib.LP_MARCH_SAMPLES_PER_WAVE if march_order == "samples" else _lib.LP_MARCH_RAYS_PER_WAVE
    n = int(grid_idx.numel())
    row_length = 0
    if rays_per_row is not None:
        row_length = int(rays_per_row)
        assert row_length >= 0 and (row_length == 0 or n % row_length == 0), (
            f"rays_per_row = {rays_per_row} does not divide the {n} rays of the batch")
    if config.check_inputs and n > 0:
        lo, hi = torch.aminmax(grid_idx)
        stats = [lo.float(), hi.float()]
        look = n > 1 and (march_order == "auto" or rays_per_row is None)
        if look:
            o, dd = rays.origins, rays.directions
            near = _neighbours(o[1:], o[:-1], dd[1:], dd[:-1])
            brk = ~near
            first = brk.float().argmax()                      # first row end (0 if the chain never breaks)
            below = torch.clamp(first + 1, max=n - 1)           # the ray one row below ray 0, if rows are first + 1 long
            vert = _neighbours(o[0], o[below], dd[0], dd[below])
            stats += [near.float().mean(), first.float(), brk.sum().float(), vert.float()]
        vals = torch.stack(stats).tolist()   # the one device sync of the call
        lo, hi = int(vals[0]), int(vals[1])
        assert lo >= 0, f"Negative grid index: {lo}"
        assert hi <= B - 1, f"A grid index is out of bounds ({hi} >= {B})"
        if look:
            coherent = vals[2] >= 0.5
            if march_order == "auto" and not coherent:
                march = _lib.LP_MARCH_SAMPLES_PER_WAVE
            if rays_per_row is None and coherent:
                w, n_brk = int(vals[3]) + 1, int(vals[4])
                if n_brk > 0 and w >= 8 and n % w == 0 and n // w >= 4 and n_brk == n // w - 1 and vals[5] > 0.5:
                    row_length = w
    return march, row_length


def check_inputs_and_choose_march(rays: Rays, grid_idx: torch.Tensor, B: int, march_order: Optional[str] = None) -> int:
    """``check_inputs_and_plan(...)[0]``: the march order alone."""
    return check_inputs_and_plan(rays, grid_idx, B, march_order, rays_per_row=0)[0]


def _decoder_dims(decoder_params: DecoderParams):
    return (int_list_of(decoder_params.n_hidden_trunk), int_list_of(decoder_params.n_hidden_opacity),
            int_list_of(decoder_params.n_hidden_color))


def lightplane_renderer(
    rays: Rays,
    grid,
    decoder_params: DecoderParams,
    # ------ config keys ------
    num_samples: int,
    gain: float,
    num_samples_inf: int = 0,
    mask_out_of_bounds_samples: bool = False,
    contract_coords: bool = False,
    disparity_at_inf: float = 1e-5,
    inject_noise_sigma: float = 0.0,
    inject_noise_seed: Optional[int] = None,
    scaffold: Optional[torch.Tensor] = None,
    color_grid=None,
    grid_sizes=None,
    color_grid_sizes=None,
    regenerate_code: bool = False,  # ignored (no code generation in the HIP build)
    triton_block_size: int = 16,  # ignored
    triton_num_warps: int = 4,  # ignored
    kernel: int = _lib.LP_KERNEL_AUTO,
    stop_transmittance: Optional[float] = None,
    arithmetic: Optional[int] = None,
    march_order: Optional[str] = None,
    rays_per_row: Optional[int] = None,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Render ``rays`` through the grid-list ``grid`` (emission-absorption ray march).

    For every ray, ``num_samples`` points equispaced in ``[near, far]`` (both ends included)
    plus ``num_samples_inf`` points beyond ``far`` (linear in disparity down to
    ``disparity_at_inf``) are sampled; the grid-list is interpolated at each point (features
    of all grids summed), decoded by the trunk / opacity / colour MLPs and composited:
    ``T_i = exp(-sum_{j<=i} gain*softplus(o_j)*delta_j)``, ``w_i = T_{i-1} - T_i``,
    ``ray_length = sum w_i depth_i``, ``feature = sum w_i sigmoid(c_i)``.

    Arguments and returns are those of the reference's ``lightplane_renderer``
    (lightplane/lightplane_renderer.py:33-212): ``grid`` is a *list* of ``[B, D, H, W, C]``
    tensors or a flat ``[sum BDHW, C]`` tensor with ``grid_sizes``; ``color_grid`` (optional)
    switches to the two-grid decoder without trunk MLP; ``scaffold`` ``[B, D, H, W]`` masks
    empty space.  Gradients flow to ``grid``, ``color_grid``, ``decoder_params.mlp_params``
    and ``rays.encoding``.

    Extension (not in the reference, off by default): ``stop_transmittance`` in (0, 1) -- or
    ``config.stop_transmittance`` when the argument is ``None`` -- enables early ray termination.  A wavefront
    (32 or 64 consecutive rays) stops marching once the transmittance of all its rays is below that value and
    the backward skips the same samples.  ``ray_length`` / ``feature`` then miss contributions bounded by
    ``stop_transmittance`` (times depth / colour) and the returned negative log transmittance is the value
    reached at the stop (``>= -log(stop_transmittance)``), i.e. alpha is exact to ``stop_transmittance``.

    ``arithmetic`` (default ``config.arithmetic`` = ``LP_ARITH_DEFAULT``): ``_lib.LP_ARITH_FP32`` runs the backward in the
    reference's arithmetic -- three bf16 limbs for every operand of the dX chains and fp32 weight-gradient products instead of
    two-limb operands; shapes outside the tuned family then run the shape-generic fp32 kernels.

    ``march_order`` (default ``config.march_order`` = "auto"): "rays" / "samples" / "auto" -- which pairs of (ray, sample) share a
    wavefront in the backward, i.e. what its gradient scatter can merge (``LP_MARCH_*``, include/lightplane_hip.h): neighbouring
    rays of an image, or consecutive samples of one ray (random ray batches).  Results agree up to fp32 summation order.

    ``rays_per_row`` (default: detected together with the march order when ``config.check_inputs`` is on, else unknown): the batch is
    made of image rows in scanline order, this many consecutive rays each (``LpRays.row_length``).  The Renderer kernels ignore the
    hint today (dealt 8 x 4 pixel patches they measured slower, profiles/r06_ray_order.txt); the Splatter's backward walk uses it.
    """
    out = _render(rays, grid, decoder_params, num_samples, gain, num_samples_inf, mask_out_of_bounds_samples,
                  contract_coords, disparity_at_inf, inject_noise_sigma, inject_noise_seed, scaffold, color_grid,
                  grid_sizes, color_grid_sizes, kernel, stop_transmittance, arithmetic=arithmetic, march_order=march_order,
                  rays_per_row=rays_per_row)
    return out[0], out[1], out[2]


def _render(rays: Rays, grid, decoder_params: DecoderParams, num_samples, gain, num_samples_inf=0,
            mask_out_of_bounds_samples=False, contract_coords=False, disparity_at_inf=1e-5, inject_noise_sigma=0.0,
            inject_noise_seed=None, scaffold=None, color_grid=None, grid_sizes=None, color_grid_sizes=None,
            kernel=_lib.LP_KERNEL_AUTO, stop_transmittance=None, bg_color=None, alpha_mode=0, arithmetic=None, march_order=None,
            rays_per_row=None):
    """``lightplane_renderer`` plus the module front-end's fused epilogue: returns ``(ray_length, neg_log_t, feature,
    alpha)``; with ``bg_color [color_chn]`` the feature is composited over it (``+ T * bg``), with ``alpha_mode`` 1 / 2
    ``alpha`` is ``1 - T`` / ``log T`` (empty tensor otherwise) -- reference renderer_module.py:552-561, in-kernel."""
    if stop_transmittance is None:
        stop_transmittance = config.stop_transmittance
    stop_transmittance = float(stop_transmittance or 0.0)
    assert 0.0 <= stop_transmittance < 1.0, "stop_transmittance has to be in [0, 1)"
    stop_neg_log_t = -math.log(stop_transmittance) if stop_transmittance > 0.0 else 0.0
    grid, color_grid, grid_sizes, color_grid_sizes = check_grid_and_color_grid(
        grid, color_grid, grid_sizes, color_grid_sizes)
    grid_is_list = isinstance(grid, (list, tuple))
    if grid_is_list:
        # zero-copy grid-list: the tensors go to the kernels as they are (per-grid base pointers); the reference
        # (and round 1) concatenated the list into one flat tensor on every call (misc_utils.py:42-45)
        grid_tensors = tuple(grid)
        grid_sizes = [list(g.shape) for g in grid_tensors]
        color_tensors = tuple(color_grid) if color_grid is not None else ()
        color_grid_sizes = [list(g.shape) for g in color_tensors] if color_grid is not None else None
        for g in grid_tensors + color_tensors:
            assert g.ndim == 5, "every grid of a grid-list has to be [B, D, H, W, C]"
    else:
        grid, color_grid, grid_sizes, color_grid_sizes = process_and_flatten_grid(
            grid, color_grid, grid_sizes, color_grid_sizes)
        grid_tensors = (grid,)
        color_tensors = (color_grid,) if color_grid is not None else ()
    descs, channels, n_rows = make_grid_descs(grid_sizes)
    if not grid_is_list:
        assert grid.ndim == 2 and grid.shape[1] == channels and grid.shape[0] == n_rows, (
            "flat grid tensor does not match grid_sizes")
    color_descs, color_n_rows = None, 0
    if color_grid is not None:
        color_descs, c_channels, color_n_rows = make_grid_descs(color_grid_sizes)
        assert c_channels == channels and color_descs[0].B == descs[0].B
        if not grid_is_list:
            assert color_grid.ndim == 2 and color_grid.shape == (color_n_rows, channels)

    if mask_out_of_bounds_samples and contract_coords:
        warnings.warn(
            "The renderer has been configured to contract the coordinates lying outside the [-1,1] cube"
            " (contract_coords=True) and to also mask out all such points (mask_out_of_bounds_samples=True).")

    dims_t, dims_o, dims_c = _decoder_dims(decoder_params)
    use_color_grid = color_grid is not None
    if use_color_grid:
        assert len(dims_t) == 0, "mlp_n_layers_trunk has to be 0 when use_separate_color_grid"
    mlp_params = decoder_params.mlp_params
    assert mlp_params.ndim == 1 and mlp_params.dtype == torch.float32
    expected = mlp_numel(dims_t) + mlp_numel(dims_o) + mlp_numel(dims_c)
    assert expected == mlp_params.numel(), (
        f"The number of elements in mlp param should be {expected}. Got {mlp_params.numel()} instead.")
    assert rays.encoding is not None, "rays.encoding is required by the functional renderer"
    assert rays.encoding.shape[1] == dims_c[0], "ray_encoding should have the same dimension as dim_in_color"
    # every raw pointer handed to the kernels: same GPU as the grid, fp32 where the kernels read floats
    dev = grid_tensors[0].device
    f32 = {"decoder_params.mlp_params": mlp_params, "rays.directions": rays.directions, "rays.origins": rays.origins,
           "rays.near": rays.near, "rays.far": rays.far, "rays.encoding": rays.encoding, "bg_color": bg_color}
    f32.update({f"grid[{i}]": g for i, g in enumerate(grid_tensors)})
    f32.update({f"color_grid[{i}]": g for i, g in enumerate(color_tensors)})
    _lib.check_tensors(dev, f32, {"rays.grid_idx": rays.grid_idx, "scaffold": scaffold})

    if inject_noise_sigma > 0.0:
        if inject_noise_seed is None:
            inject_noise_seed = int(random.randint(0, 1000000))
    else:
        inject_noise_seed = 0

    B = descs[0].B
    grid_idx = rays.grid_idx.to(torch.int32).contiguous()
    march, row_length = check_inputs_and_plan(rays, grid_idx, B, march_order, rays_per_row)

    scaffold_shape = None
    if scaffold is not None:
        assert scaffold.ndim == 4 and scaffold.shape[0] == B, "scaffold has to be [B, D, H, W]"
        scaffold_shape = tuple(int(v) for v in scaffold.shape)
        scaffold = scaffold.to(torch.float32).contiguous()
    if bg_color is not None:
        assert bg_color.ndim == 1 and bg_color.numel() == int(decoder_params.color_chn)
        bg_color = bg_color.contiguous()

    cfg = _RendererCfg(
        descs=descs, channels=channels, n_rows=n_rows, color_descs=color_descs, color_n_rows=color_n_rows,
        dims_trunk=dims_t, dims_opacity=dims_o, dims_color=dims_c, color_chn=int(decoder_params.color_chn),
        num_samples=int(num_samples), num_samples_inf=int(num_samples_inf), gain=float(gain),
        mask_out_of_bounds_samples=bool(mask_out_of_bounds_samples), contract_coords=bool(contract_coords),
        disparity_at_inf=float(disparity_at_inf), noise_sigma=float(inject_noise_sigma),
        noise_seed=int(inject_noise_seed), scaffold_shape=scaffold_shape, kernel=int(kernel),
        stop_neg_log_t=stop_neg_log_t, n_grid_tensors=len(grid_tensors), n_color_tensors=len(color_tensors),
        grid_is_list=grid_is_list, alpha_mode=int(alpha_mode),
        arithmetic=int(config.arithmetic if arithmetic is None else arithmetic), march_order=int(march), row_length=int(row_length),
    )
    return LightplaneFunction.apply(
        cfg, mlp_params, rays.encoding, rays.directions.contiguous(), rays.origins.contiguous(), grid_idx,
        rays.near.contiguous(), rays.far.contiguous(), scaffold, bg_color, *grid_tensors, *color_tensors)


def renderer_corner_rows(rays: Rays, grid_sizes, num_samples: int, num_samples_inf: int = 0,
                         contract_coords: bool = False, disparity_at_inf: float = 1e-5) -> torch.Tensor:
    """Integer corner rows of the march ``[N, S_tot, K_tot]`` (int64, -1 = out of range), rows
    relative to each grid's start.  Parity hook for the bit-exact index tests."""
    descs, channels, n_rows = make_grid_descs(grid_sizes)
    dev = rays.device
    stream = _lib.current_stream(dev)
    k_tot = sum(8 if d.kind == "voxel" else 4 for d in descs)
    s_tot = num_samples + num_samples_inf
    out = torch.empty(rays.n_rays, s_tot, k_tot, dtype=torch.int64, device=dev)
    a = _lib.LpRendererArgs()
    a.rays = _lib.make_rays(rays.directions.contiguous(), rays.origins.contiguous(),
                            rays.grid_idx.to(torch.int32).contiguous(), rays.near.contiguous(),
                            rays.far.contiguous(), None)
    a.grid = _lib.make_grid_list(None, descs, channels, n_rows)
    a.march = _lib.make_march(num_samples, num_samples_inf, False, contract_coords, disparity_at_inf)
    keep = (a, out)  # noqa: F841  (keep tensors alive across the async launch)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().lp_renderer_corner_rows(ctypes.byref(a), out.data_ptr(), stream),
                   "lp_renderer_corner_rows")
    return out
