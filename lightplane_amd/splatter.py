"""Functional Lightplane Splatter on MI355X: ``lightplane_splatter`` (+ MLP variant).

Drop-in for the reference's ``lightplane/lightplane_splatter.py`` (`lightplane_splatter`
:31-164, `lightplane_mlp_splatter` :167-338, `LightplaneSplatterFunction` :341-700).  The
reference marches every ray twice (feature launch :505, unit-weight launch :507-539) and
normalises with three whole-grid PyTorch passes (:541, :584, backward :608); here one HIP
launch splats features and weights together (``lp_splatter_forward``), one fused pass
normalises in place (``lp_splatter_normalize``) and the backward kernel divides by the
clamped weight while gathering (``lp_splatter_backward``).

Multi-GPU: pass ``process_group`` to sum the UN-normalised feature and weight grids over
the ray shards (RCCL all-reduce) *before* normalising -- see ``lightplane_amd.parallel``.  The
output is then a REPLICATED tensor and its incoming gradient has to be the gradient of the total
(all-rank) loss: if what follows is itself ray-sharded (cfg 5: every rank renders its own rays from
the splatted grid), wrap the output with ``parallel.replicate_with_grad_allreduce`` so that the
partial gradients are summed before they reach the Splatter backward.  The MLP-Splatter's
``grad_mlp_params`` / ``grad_input_grid`` (partial sums over the local rays) are all-reduced in
its backward.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib, config
from .grids import GridDesc, check_grid, make_grid_descs, process_and_flatten_grid, sizes_to_list, unflatten_grid
from .params import SplatterParams, int_list_of
from .rays import Rays


@dataclass
class _SplatterCfg:
    descs: List[GridDesc]
    channels: int
    n_rows: int
    num_samples: int
    num_samples_inf: int
    mask_out_of_bounds_samples: bool
    contract_coords: bool
    disparity_at_inf: float
    process_group: object = None
    # MLP-Splatter only
    in_descs: Optional[List[GridDesc]] = None
    in_channels: int = 0
    in_n_rows: int = 0
    mlp_dims: Optional[List[int]] = None
    kernel: int = 0
    in_is_list: bool = False   # the input grid-list arrives as one tensor per grid (zero-copy) instead of a flat tensor
    n_in_tensors: int = 1
    march_order: int = 0       # LP_MARCH_* of the plain Splatter's forward walk
    row_length: int = 0        # LpRays.row_length: rays per image row (0 = unknown)


def _fill_args(cfg: _SplatterCfg, directions, origins, grid_idx, near, far, feature, mlp_params=None,
               input_grid=None) -> _lib.LpSplatterArgs:
    a = _lib.LpSplatterArgs()
    a.rays = _lib.make_rays(directions, origins, grid_idx, near, far, feature, cfg.row_length)
    a.march = _lib.make_march(cfg.num_samples, cfg.num_samples_inf, cfg.mask_out_of_bounds_samples,
                              cfg.contract_coords, cfg.disparity_at_inf)
    a.out = _lib.make_grid_list(None, cfg.descs, cfg.channels, cfg.n_rows)
    if cfg.mlp_dims is not None:
        a.input_grid = _lib.make_grid_list(list(input_grid) if cfg.in_is_list else input_grid[0], cfg.in_descs,
                                           cfg.in_channels, cfg.in_n_rows)
        a.mlp_params = _lib.ptr(mlp_params)
        a.n_mlp_params = mlp_params.numel()
        a.mlp = _lib.make_mlp(cfg.mlp_dims, 0)
        a.kernel = int(cfg.kernel)
    a.march_order = int(cfg.march_order)
    return a


class LightplaneSplatterFunction(torch.autograd.Function):
    """Autograd boundary of the Splatter (name kept from the reference, :341)."""

    @staticmethod
    def forward(ctx, feature, cfg: _SplatterCfg, directions, origins, grid_idx, near, far):
        dev = feature.device
        stream = _lib.current_stream(dev)
        feature = feature.contiguous()
        out = torch.zeros(cfg.n_rows, cfg.channels, device=dev, dtype=torch.float32)
        weight = torch.zeros(cfg.n_rows, device=dev, dtype=torch.float32)
        a = _fill_args(cfg, directions, origins, grid_idx, near, far, feature)
        a.out.data = _lib.ptr(out)
        a.out_feature, a.out_weight = _lib.ptr(out), _lib.ptr(weight)
        L = _lib.lib()
        with torch.cuda.device(dev):
            _lib.check(L.lp_splatter_forward(ctypes.byref(a), stream), "lp_splatter_forward")
            if cfg.process_group is not None:
                from .parallel import allreduce_sum_
                allreduce_sum_([out, weight], cfg.process_group)
            _lib.check(L.lp_splatter_normalize(out.data_ptr(), weight.data_ptr(), cfg.n_rows, cfg.channels, stream),
                       "lp_splatter_normalize")
        ctx.save_for_backward(weight, feature, directions, origins, grid_idx, near, far)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, feature, directions, origins, grid_idx, near, far = ctx.saved_tensors
        cfg: _SplatterCfg = ctx.cfg
        if not ctx.needs_input_grad[0]:
            return (None,) * 7
        dev = feature.device
        stream = _lib.current_stream(dev)
        grad_out = grad_out.contiguous()
        grad_feature = torch.zeros_like(feature)
        a = _fill_args(cfg, directions, origins, grid_idx, near, far, feature)
        a.grad_out, a.weight, a.grad_encoding = _lib.ptr(grad_out), _lib.ptr(weight), _lib.ptr(grad_feature)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lp_splatter_backward(ctypes.byref(a), stream), "lp_splatter_backward")
        if config.check_finite_grads:
            assert torch.isfinite(grad_feature).all()
        return (grad_feature,) + (None,) * 6


class LightplaneMLPSplatterFunction(torch.autograd.Function):
    """Autograd boundary of the MLP-Splatter (the reference routes both variants through
    ``LightplaneSplatterFunction``, lightplane_splatter.py:341-700; the MLP path is :440-501, :608-700).  The input
    grid-list arrives as one flat tensor or as one tensor per grid (never concatenated, see ``LpGrid.data``)."""

    @staticmethod
    def forward(ctx, feature, mlp_params, cfg: _SplatterCfg, directions, origins, grid_idx, near, far, *input_grids):
        dev = feature.device
        stream = _lib.current_stream(dev)
        feature, mlp_params = feature.contiguous(), mlp_params.contiguous()
        input_grids = tuple(g.contiguous() for g in input_grids)
        out = torch.zeros(cfg.n_rows, cfg.channels, device=dev, dtype=torch.float32)
        weight = torch.zeros(cfg.n_rows, device=dev, dtype=torch.float32)
        a = _fill_args(cfg, directions, origins, grid_idx, near, far, feature, mlp_params, input_grids)
        a.out.data = _lib.ptr(out)
        a.out_feature, a.out_weight = _lib.ptr(out), _lib.ptr(weight)
        L = _lib.lib()
        with torch.cuda.device(dev):
            _lib.check(L.lp_splatter_forward(ctypes.byref(a), stream), "lp_splatter_forward")
            if cfg.process_group is not None:
                from .parallel import allreduce_sum_
                allreduce_sum_([out, weight], cfg.process_group)
            _lib.check(L.lp_splatter_normalize(out.data_ptr(), weight.data_ptr(), cfg.n_rows, cfg.channels, stream),
                       "lp_splatter_normalize")
        ctx.save_for_backward(weight, feature, mlp_params, directions, origins, grid_idx, near, far, *input_grids)
        ctx.cfg = cfg
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, feature, mlp_params, directions, origins, grid_idx, near, far = ctx.saved_tensors[:8]
        input_grids = ctx.saved_tensors[8:]
        cfg: _SplatterCfg = ctx.cfg
        need_feat, need_params = ctx.needs_input_grad[:2]
        need_grids = ctx.needs_input_grad[8:]
        if not (need_feat or need_params or any(need_grids)):
            return (None,) * (8 + len(input_grids))
        dev = feature.device
        stream = _lib.current_stream(dev)
        grad_out = grad_out.contiguous()
        grad_feature = torch.zeros_like(feature) if need_feat else None
        grad_params = torch.zeros_like(mlp_params) if need_params else None
        grad_in = [torch.zeros_like(g) for g in input_grids] if any(need_grids) else None
        a = _fill_args(cfg, directions, origins, grid_idx, near, far, feature, mlp_params, input_grids)
        a.grad_out, a.weight = _lib.ptr(grad_out), _lib.ptr(weight)
        a.grad_encoding, a.grad_mlp_params = _lib.ptr(grad_feature), _lib.ptr(grad_params)
        if grad_in is not None:
            if cfg.in_is_list:
                _lib.fill_ptr_list(a.grad_input_grid_list, grad_in)
            else:
                a.grad_input_grid = _lib.ptr(grad_in[0])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lp_splatter_backward(ctypes.byref(a), stream), "lp_splatter_backward")
        if cfg.process_group is not None:
            # ray-sharded MLP-Splatter: the MLP / input-grid gradients are partial sums over this rank's rays
            # (SURVEY.md 8(e)); grad_feature belongs to the local rays and stays local
            from .parallel import allreduce_sum_
            allreduce_sum_([grad_params] + (grad_in or []), cfg.process_group)
        if config.check_finite_grads:
            for g in [grad_feature, grad_params] + (grad_in or []):
                assert g is None or torch.isfinite(g).all()
        gi = [None] * len(input_grids) if grad_in is None else [g if nd else None for g, nd in zip(grad_in, need_grids)]
        return (grad_feature, grad_params) + (None,) * 6 + tuple(gi)


def _prep_rays(rays: Rays, B: int, march_order: Optional[str] = "rays", rays_per_row: Optional[int] = 0):
    """Device / dtype checks, the ``grid_idx`` range check and (riding on its sync) the march order of the forward walk and the
    row length of an image batch (``renderer.check_inputs_and_plan``).  Returns ((march order, row length), ray tensors)."""
    from .renderer import check_inputs_and_plan
    _lib.check_tensors(
        rays.encoding.device,
        {"rays.directions": rays.directions, "rays.origins": rays.origins, "rays.near": rays.near,
         "rays.far": rays.far, "rays.encoding": rays.encoding},
        {"rays.grid_idx": rays.grid_idx})
    grid_idx = rays.grid_idx.to(torch.int32).contiguous()
    march = check_inputs_and_plan(rays, grid_idx, B, march_order, rays_per_row)
    return march, (rays.directions.contiguous(), rays.origins.contiguous(), grid_idx, rays.near.contiguous(),
                   rays.far.contiguous())


def lightplane_splatter(
    rays: Rays,
    output_grid_size,
    # ------ config keys ------
    num_samples: int,
    num_samples_inf: int = 0,
    mask_out_of_bounds_samples: bool = False,
    contract_coords: bool = False,
    disparity_at_inf: float = 1e-5,
    return_list: bool = True,
    regenerate_code: bool = False,  # ignored
    triton_block_size: int = 16,  # ignored
    triton_num_warps: int = 4,  # ignored
    process_group=None,
    march_order: Optional[str] = None,
    rays_per_row: Optional[int] = None,
):
    """Splat ``rays.encoding`` into a zero-initialised grid-list of shape ``output_grid_size``.

    Every sample point of every ray adds ``encoding * w_k`` to the 8 (voxel) / 4 (plane)
    neighbouring cells with tri/bi-linear weights ``w_k`` and ``w_k`` to a weight grid; the
    result is ``features / clamp(weights, 1e-5)``.  Arguments / returns follow the reference's
    ``lightplane_splatter`` (lightplane/lightplane_splatter.py:31-164): a list of
    ``[B, D, H, W, C]`` tensors, or the flat ``[sum BDHW, C]`` tensor if ``return_list=False``.

    ``march_order`` ("auto" / "rays" / "samples", default ``config.march_order``): which (ray, sample) pairs share a wavefront of
    the forward walk -- 32 neighbouring rays (image-coherent batches) or 32 consecutive samples of one ray (unrelated rays, e.g. the
    reference's ``tests/splatter_speed_benchmark.py``); see ``lightplane_renderer``.  Same result up to fp32 summation order.
    ``rays_per_row``: rays per image row of a scanline-ordered batch (detected with the input check unless given; a hint for the
    backward walk, which then gathers for 2 x 4 pixel patches per wavefront; see ``lightplane_renderer``).
    """
    sizes = sizes_to_list(output_grid_size)
    descs, channels, n_rows = make_grid_descs(sizes)
    assert rays.encoding is not None, "rays.encoding (the splatted feature) is required"
    assert rays.encoding.shape[1] == channels, (
        f"splatting feature width {rays.encoding.shape[1]} != output grid channels {channels}")
    assert rays.encoding.dtype == torch.float32
    cfg = _SplatterCfg(descs, channels, n_rows, int(num_samples), int(num_samples_inf),
                       bool(mask_out_of_bounds_samples), bool(contract_coords), float(disparity_at_inf),
                       process_group)
    (cfg.march_order, cfg.row_length), ray_tensors = _prep_rays(rays, descs[0].B, march_order, rays_per_row)
    out = LightplaneSplatterFunction.apply(rays.encoding, cfg, *ray_tensors)
    if return_list:
        return list(unflatten_grid(out, sizes))
    return out


def mlp_splatter_kernel_family(output_grid_size, mlp_params: SplatterParams, input_grid_sizes, num_samples_inf: int = 0) -> int:
    """Kernel family that runs an MLP-Splatter of these shapes: 3 = layer-looped MFMA family (2-4 layers, widths 16 / 32 / 64),
    0 = shape-generic kernels (``lp_splatter_kernel_family``: shapes only, needs no GPU)."""
    descs, channels, n_rows = make_grid_descs(sizes_to_list(output_grid_size))
    in_descs, in_channels, in_n_rows = make_grid_descs(sizes_to_list(input_grid_sizes))
    a = _lib.LpSplatterArgs()
    a.march = _lib.make_march(1, int(num_samples_inf), False, False, 1e-5)
    a.out = _lib.make_grid_list(None, descs, channels, n_rows)
    a.input_grid = _lib.make_grid_list(None, in_descs, in_channels, in_n_rows)
    a.mlp = _lib.make_mlp(int_list_of(mlp_params.n_hidden), 0)
    return int(_lib.lib().lp_splatter_kernel_family(ctypes.byref(a)))


def lightplane_mlp_splatter(
    rays: Rays,
    output_grid_size,
    mlp_params: SplatterParams,
    input_grid,
    num_samples: int,
    num_samples_inf: int = 0,
    mask_out_of_bounds_samples: bool = False,
    contract_coords: bool = False,
    disparity_at_inf: float = 1e-5,
    input_grid_sizes=None,
    return_list: bool = True,
    regenerate_code: bool = False,  # ignored
    triton_block_size: int = 16,  # ignored
    triton_num_warps: int = 4,  # ignored
    process_group=None,
    kernel: int = _lib.LP_KERNEL_AUTO,
):
    """Splat ``MLP(sample(input_grid, x) + rays.encoding)`` into a zero-initialised grid-list.

    For every sample point of every ray the *input* grid-list is interpolated (features of all its
    grids summed, like the Renderer does), the ray encoding is added, the result goes through the
    MLP (ReLU between layers, none after the last) and is splatted into the output grid-list with
    tri/bi-linear weights; the output is normalised by the splatted weights
    (``features / clamp(weights, 1e-5)``).  Arguments / returns follow the reference's
    ``lightplane_mlp_splatter`` (lightplane/lightplane_splatter.py:167-338).  Gradients flow to
    ``rays.encoding``, ``mlp_params.mlp_params`` and ``input_grid``.
    """
    sizes = sizes_to_list(output_grid_size)
    descs, channels, n_rows = make_grid_descs(sizes)
    assert mlp_params is not None and len(mlp_params.n_hidden) > 1, (
        "mlp depth has to be bigger than 1 when using input_grid")
    assert input_grid is not None, "input_grid cannot be None when mlp_params is not None"
    check_grid(input_grid, input_grid_sizes)
    in_is_list = isinstance(input_grid, (list, tuple))
    if in_is_list:  # zero-copy: one tensor per grid goes to the kernels as it is
        in_tensors = tuple(input_grid)
        input_grid_sizes = [list(g.shape) for g in in_tensors]
    else:
        input_grid, _, input_grid_sizes, _ = process_and_flatten_grid(input_grid, None, input_grid_sizes, None)
        in_tensors = (input_grid,)
    in_descs, in_channels, in_n_rows = make_grid_descs(input_grid_sizes)
    if not in_is_list:
        assert input_grid.ndim == 2 and input_grid.shape == (in_n_rows, in_channels), (
            "flat input grid tensor does not match input_grid_sizes")
    dims = int_list_of(mlp_params.n_hidden)
    assert rays.encoding is not None, "rays.encoding is required"
    assert rays.encoding.dtype == torch.float32
    assert dims[0] == in_channels == rays.encoding.shape[1], (
        f"MLP input width {dims[0]} must equal the input grid channels {in_channels} and the ray encoding "
        f"width {rays.encoding.shape[1]}")
    assert dims[-1] == channels, f"MLP output width {dims[-1]} != output grid channels {channels}"
    assert in_descs[0].B == descs[0].B, "input and output grid-lists must share the batch size"
    flat_params = mlp_params.mlp_params
    assert flat_params.ndim == 1 and flat_params.dtype == torch.float32
    from .params import mlp_numel
    assert flat_params.numel() == mlp_numel(dims), (
        f"The number of elements in mlp param should be {mlp_numel(dims)}. Got {flat_params.numel()} instead.")
    f32 = {"mlp_params.mlp_params": flat_params}
    f32.update({f"input_grid[{i}]": g for i, g in enumerate(in_tensors)})
    _lib.check_tensors(rays.encoding.device, f32)
    cfg = _SplatterCfg(descs, channels, n_rows, int(num_samples), int(num_samples_inf),
                       bool(mask_out_of_bounds_samples), bool(contract_coords), float(disparity_at_inf),
                       process_group, in_descs, in_channels, in_n_rows, dims, int(kernel), in_is_list, len(in_tensors))
    out = LightplaneMLPSplatterFunction.apply(rays.encoding, flat_params, cfg, *_prep_rays(rays, descs[0].B)[1],
                                              *in_tensors)
    if return_list:
        return list(unflatten_grid(out, sizes))
    return out
