"""Grid-list <-> flat tensor helpers and grid checks.

Public names follow the reference's ``lightplane/misc_utils.py``
(`flatten_grid` :25-46, `unflatten_grid` :49-70, `if_not_none_else` :73-75,
`pad_feature_to_block_size` :78-94, `is_in_bounds` :97-101, `check_grid`
:115-140, `check_grid_and_color_grid` :143-198, `process_and_flatten_grid`
:201-234).

HBM layout the HIP kernels read (same as the reference): a grid-list is one
flat ``[sum_g B*D_g*H_g*W_g, C]`` f32 channels-last tensor; grid ``g`` starts
at row ``sum_{j<g} B*D_j*H_j*W_j`` and cell ``(b, z, y, x)`` is row
``((b*D + z)*H + y)*W + x`` inside it.  Shapes travel to the kernel as host
integers (``GridDesc``), never as a device tensor, so no device sync is needed.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, List, Optional, Sequence, Tuple

import torch


def assert_shape(x: torch.Tensor, shape: Sequence[int]) -> None:
    assert tuple(x.shape) == tuple(shape), f"expected shape {tuple(shape)}, got {tuple(x.shape)}"


def if_not_none_else(x: Any, y: Any) -> Any:
    return y if x is None else x


def sizes_to_list(grid_sizes) -> List[List[int]]:
    """Normalise grid sizes (tensor / nested sequence) to ``List[List[int]]`` on the host."""
    if torch.is_tensor(grid_sizes):
        from .params import int_list_of  # cached per tensor object: a GPU-resident size tensor is read once
        grid_sizes = int_list_of(grid_sizes)
    out = [[int(v) for v in gs] for gs in grid_sizes]
    for gs in out:
        assert len(gs) == 5, f"each grid size has to be [B, D, H, W, C], got {gs}"
    return out


def flatten_grid(grid: Sequence[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """Stack a grid-list into ``([sum BDHW, C] tensor, [G, 5] int32 sizes)``."""
    dev = grid[0].device
    sizes = torch.tensor([list(g.shape) for g in grid], dtype=torch.int32, device=dev)
    return _flatten_only(grid), sizes


def _flatten_only(grid: Sequence[torch.Tensor]) -> torch.Tensor:
    """The flat ``[sum BDHW, C]`` tensor of a grid-list (no size tensor: no host->device copy)."""
    if len(grid) == 1:
        # zero-copy for single-grid lists (a 256^3 x 32 voxel grid is 2.1 GB: torch.cat would copy it,
        # and its backward would copy the gradient again)
        flat = grid[0].reshape(-1, grid[0].shape[-1]).contiguous()
    else:
        flat = torch.cat([g.reshape(-1, g.shape[-1]) for g in grid], dim=0).contiguous()
    return flat


def unflatten_grid(grid: torch.Tensor, grid_sizes) -> Tuple[torch.Tensor, ...]:
    """Views of the flat tensor as the list of ``[B, D, H, W, C]`` grids."""
    sizes = sizes_to_list(grid_sizes)
    rows = [gs[0] * gs[1] * gs[2] * gs[3] for gs in sizes]
    parts = grid.split(rows, dim=0)
    return tuple(p.reshape(*gs) for p, gs in zip(parts, sizes))


def pad_feature_to_block_size(feature: torch.Tensor, block_size: int) -> torch.Tensor:
    """Zero-pad dim 0 to a multiple of ``block_size`` (API compatibility only)."""
    n_pad = (-feature.shape[0]) % int(block_size)
    if n_pad == 0:
        return feature
    tail = feature.new_zeros((n_pad,) + tuple(feature.shape[1:]))
    return torch.cat([feature, tail], dim=0)


def is_in_bounds(points: torch.Tensor) -> torch.Tensor:
    """``[..., 1]`` bool: all coordinates within ``[-1, 1]``."""
    return (points.abs() <= 1.0).all(dim=-1, keepdim=True)


def _numel_of_sizes(grid_sizes) -> int:
    total = 0
    for gs in sizes_to_list(grid_sizes):
        n = 1
        for v in gs:
            n *= v
        total += n
    return total


def _check_list_against_sizes(grid: Sequence[torch.Tensor], grid_sizes) -> None:
    for g, gs in zip(grid, sizes_to_list(grid_sizes)):
        assert_shape(g, gs)


def check_grid(grid, grid_sizes=None):
    """A grid is a *list* of 5-D tensors or a 2-D flat tensor (then ``grid_sizes`` is
    mandatory).  Anything else raises ``NotImplementedError`` like the reference."""
    if isinstance(grid, list):
        if grid_sizes is not None:
            _check_list_against_sizes(grid, grid_sizes)
    elif isinstance(grid, torch.Tensor):
        assert grid_sizes is not None, "grid_sizes cannot be None when grid is a tensor"
        assert _numel_of_sizes(grid_sizes) == grid.numel(), (
            "grid_sizes has to be compatible to grid tensor shapes!"
        )
    else:
        raise NotImplementedError("grid should be either tensor or list")
    return grid, grid_sizes


def check_grid_and_color_grid(grid, color_grid, grid_sizes=None, color_grid_sizes=None):
    """Consistency checks of the (opacity) grid and the optional colour grid."""
    if color_grid is not None:
        assert type(grid) == type(color_grid), "grid and color_grid should have the same type"
    check_grid(grid, grid_sizes)
    if color_grid is None:
        return grid, color_grid, grid_sizes, color_grid_sizes
    if isinstance(grid, list):
        assert all(cg.shape[0] == g.shape[0] for cg, g in zip(color_grid, grid)), (
            "color_grid's batch size should be the same as grid's batch_size"
        )
        assert all(cg.shape[-1] == g.shape[-1] for cg, g in zip(color_grid, grid)), (
            "color_grid's feature dimension should be the same as grid's feature dimension"
        )
        if color_grid_sizes is not None:
            _check_list_against_sizes(color_grid, color_grid_sizes)
    else:
        assert color_grid_sizes is not None, (
            "color_grid_sizes cannot be None when color_grid is a tensor"
        )
        assert _numel_of_sizes(color_grid_sizes) == color_grid.numel(), (
            "grid_sizes has to be compatible to grid tensor shapes!"
        )
    return grid, color_grid, grid_sizes, color_grid_sizes


def process_and_flatten_grid(grid, color_grid, grid_sizes=None, color_grid_sizes=None):
    """Bring (grid, color_grid) to the flat form; sizes come back as host lists.

    (The reference returns size *tensors*, misc_utils.py:201-234; the HIP path
    passes sizes by value, so they stay on the host.)
    """
    if isinstance(grid, list):
        if color_grid is not None:
            color_grid_sizes = [list(g.shape) for g in color_grid]
            color_grid = _flatten_only(color_grid)
        else:
            color_grid_sizes = None
        grid_sizes = [list(g.shape) for g in grid]
        grid = _flatten_only(grid)
    elif isinstance(grid, torch.Tensor):
        grid_sizes = sizes_to_list(grid_sizes)
        if color_grid is not None:
            color_grid_sizes = sizes_to_list(color_grid_sizes)
    else:
        raise NotImplementedError("grid should be flatten either tensor or list")
    return grid, color_grid, grid_sizes, color_grid_sizes


# --------------------------------------------------------------------------------------
# Host-side grid descriptors handed to the C-ABI by value
# --------------------------------------------------------------------------------------


@dataclass(frozen=True)
class GridDesc:
    """One grid of a grid-list: shape + start row in the flat tensor."""

    B: int
    D: int
    H: int
    W: int
    row_offset: int  # first row of this grid in the flat [rows, C] tensor

    @property
    def n_rows(self) -> int:
        return self.B * self.D * self.H * self.W

    @property
    def kind(self) -> str:
        """'voxel' (3 non-singleton dims) or 'plane' (exactly 2)."""
        ns = sum(int(s > 1) for s in (self.D, self.H, self.W))
        if ns == 3:
            return "voxel"
        if ns == 2:
            return "plane"
        raise ValueError(
            f"Unexpected n non-singular dim of input grid ({ns}); "
            "only voxel grids [B,D,H,W,C] and planes with one singleton dim are supported"
        )


def make_grid_descs(grid_sizes) -> Tuple[List[GridDesc], int, int]:
    """``(descs, n_channels, total_rows)`` for a list of ``[B, D, H, W, C]`` sizes."""
    sizes = sizes_to_list(grid_sizes)
    assert len(sizes) > 0
    C = sizes[0][4]
    B = sizes[0][0]
    descs: List[GridDesc] = []
    row = 0
    for gs in sizes:
        assert gs[4] == C, "All grids should have the same feature dimensions."
        assert gs[0] == B, "All grids should have the same batch size."
        d = GridDesc(gs[0], gs[1], gs[2], gs[3], row)
        d.kind  # validates
        descs.append(d)
        row += d.n_rows
    return descs, C, row
