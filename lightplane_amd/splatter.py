"""Functional Lightplane Splatter on MI355X: ``lightplane_splatter`` (+ MLP variant).

Drop-in for the reference's ``lightplane/lightplane_splatter.py`` (`lightplane_splatter`
:31-164, `lightplane_mlp_splatter` :167-338, `LightplaneSplatterFunction` :341-700).  The
reference marches every ray twice (feature launch :505, unit-weight launch :507-539) and
normalises with three whole-grid PyTorch passes (:541, :584, backward :608); here one HIP
launch splats features and weights together (``lp_splatter_forward``), one fused pass
normalises in place (``lp_splatter_normalize``) and the backward kernel divides by the
clamped weight while gathering (``lp_splatter_backward``).

Multi-GPU: pass ``process_group`` to sum the UN-normalised feature and weight grids over
the ray shards (RCCL all-reduce) *before* normalising -- see ``lightplane_amd.parallel``.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional

import torch

from . import _lib, config
from .grids import GridDesc, check_grid, make_grid_descs, process_and_flatten_grid, sizes_to_list, unflatten_grid
from .params import SplatterParams
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


def _fill_args(cfg: _SplatterCfg, directions, origins, grid_idx, near, far, feature) -> _lib.LpSplatterArgs:
    a = _lib.LpSplatterArgs()
    a.rays = _lib.make_rays(directions, origins, grid_idx, near, far, feature)
    a.march = _lib.make_march(cfg.num_samples, cfg.num_samples_inf, cfg.mask_out_of_bounds_samples,
                              cfg.contract_coords, cfg.disparity_at_inf)
    a.out = _lib.make_grid_list(None, cfg.descs, cfg.channels, cfg.n_rows)
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
        grad_feature = torch.empty_like(feature)
        a = _fill_args(cfg, directions, origins, grid_idx, near, far, feature)
        a.grad_out, a.weight, a.grad_encoding = _lib.ptr(grad_out), _lib.ptr(weight), _lib.ptr(grad_feature)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lp_splatter_backward(ctypes.byref(a), stream), "lp_splatter_backward")
        if config.check_finite_grads:
            assert torch.isfinite(grad_feature).all()
        return (grad_feature,) + (None,) * 6


def _prep_rays(rays: Rays, B: int):
    grid_idx = rays.grid_idx.to(torch.int32).contiguous()
    if config.check_inputs and grid_idx.numel() > 0:
        lo, hi = torch.aminmax(grid_idx)
        lo, hi = int(lo), int(hi)
        assert lo >= 0, f"Negative grid index: {lo}"
        assert hi <= B - 1, f"A grid index is out of bounds ({hi} >= {B})"
    return (rays.directions.contiguous(), rays.origins.contiguous(), grid_idx, rays.near.contiguous(),
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
):
    """Splat ``rays.encoding`` into a zero-initialised grid-list of shape ``output_grid_size``.

    Every sample point of every ray adds ``encoding * w_k`` to the 8 (voxel) / 4 (plane)
    neighbouring cells with tri/bi-linear weights ``w_k`` and ``w_k`` to a weight grid; the
    result is ``features / clamp(weights, 1e-5)``.  Arguments / returns follow the reference's
    ``lightplane_splatter`` (lightplane/lightplane_splatter.py:31-164): a list of
    ``[B, D, H, W, C]`` tensors, or the flat ``[sum BDHW, C]`` tensor if ``return_list=False``.
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
    out = LightplaneSplatterFunction.apply(rays.encoding, cfg, *_prep_rays(rays, descs[0].B))
    if return_list:
        return list(unflatten_grid(out, sizes))
    return out


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
    regenerate_code: bool = False,
    triton_block_size: int = 16,
    triton_num_warps: int = 4,
):
    """MLP-Splatter: splat ``MLP(sample(input_grid, x) + rays.encoding)`` (reference
    lightplane_splatter.py:167-338).  Listed as "next" in SURVEY.md 8(f); not built yet."""
    raise NotImplementedError(
        "lightplane_mlp_splatter is not implemented in the HIP library yet (SURVEY.md 8(f) item 1)")
