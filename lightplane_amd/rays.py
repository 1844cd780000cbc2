"""Ray container and ray-side helpers of the Renderer / Splatter hot path.

Mirrors the public surface of the reference's ``lightplane/ray_utils.py``
(`Rays` :19-57, `pad_to_block_size` :109-140, `calc_harmonic_embedding` :181-212,
`calc_harmonic_embedding_dim` :215-217, `jitter_near_far` :220-229,
`_validate_rays` :232-274) with the same field names, argument meaning and
error behaviour (``AssertionError`` on malformed rays).  ``Rays.to`` is a
working version of the reference's broken one (:142-169).

The HIP kernels consume the fields exactly as laid out here:
``directions`` / ``origins`` are ``[N, 3]`` f32 row-major, ``near`` / ``far``
``[N]`` f32, ``grid_idx`` ``[N]`` integer (cast to int32 at the C-ABI),
``encoding`` ``[N, E]`` f32.
"""
from __future__ import annotations

import copy as _copy
import dataclasses
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

_TENSOR_FIELDS = ("directions", "origins", "grid_idx", "near", "far", "encoding")


def _validate_rays(directions, origins, grid_idx, near, far, encoding) -> None:
    """Shape / dtype / device checks (reference ray_utils.py:232-274)."""
    assert directions.ndim == 2, "directions has to be [n_rays, 3]"
    assert origins.ndim == 2, "origins has to be [n_rays, 3]"
    assert grid_idx.ndim == 1, "grid_idx has to be [n_rays]"
    assert near.ndim == 1, "near has to be [n_rays]"
    assert far.ndim == 1, "far has to be [n_rays]"
    assert not grid_idx.is_floating_point(), "grid_idx has to be an integer tensor"
    assert directions.shape[1] == 3 and origins.shape[1] == 3
    n_rays = directions.shape[0]
    dev = directions.device
    named = {
        "directions": directions,
        "origins": origins,
        "near": near,
        "far": far,
        "grid_idx": grid_idx,
    }
    for name, t in named.items():
        assert t.device == dev, f"{name} is on a wrong device ({t.device}, expected {dev})"
        assert (
            t.shape[0] == n_rays
        ), f"Unexpected number of elements in {name} ({t.shape[0]}, expected {n_rays})"
    if encoding is not None:
        assert encoding.ndim == 2, "encoding has to be [n_rays, C]"
        assert encoding.shape[0] == n_rays
        assert encoding.device == dev


@dataclass
class Rays:
    """A batch of rendering / splatting rays.

    A point on ray ``i`` is ``origins[i] + t * directions[i]`` with
    ``t`` in ``[near[i], far[i]]``; ``directions`` need not be normalised.
    ``grid_idx[i]`` selects the batch element (scene) of the grid-list the
    ray is rendered from / splatted into.  ``encoding`` optionally carries a
    per-ray feature vector (added to the colour-MLP input by the Renderer,
    splatted by the Splatter).
    """

    directions: torch.Tensor  # [N, 3]
    origins: torch.Tensor  # [N, 3]
    grid_idx: torch.Tensor  # [N] integer
    near: torch.Tensor  # [N]
    far: torch.Tensor  # [N]
    encoding: Optional[torch.Tensor] = None  # [N, E]

    def __post_init__(self):
        _validate_rays(
            self.directions, self.origins, self.grid_idx, self.near, self.far, self.encoding
        )

    # -- basic container behaviour -------------------------------------------------
    @property
    def device(self) -> torch.device:
        return self.directions.device

    @property
    def n_rays(self) -> int:
        return int(self.directions.shape[0])

    def _map(self, fn) -> "Rays":
        kw = {}
        for name in _TENSOR_FIELDS:
            v = getattr(self, name)
            kw[name] = None if v is None else fn(v)
        return type(self)(**kw)

    def __getitem__(self, key) -> "Rays":
        """Sub-select rays (any torch index valid on dim 0)."""
        return self._map(lambda v: v[key])

    def pad_to_block_size(self, block_size: int) -> Tuple["Rays", int]:
        """Zero-pad the ray count to a multiple of ``block_size``.

        Returns the padded rays and the number of rays added.  Kept for API
        compatibility (reference ray_utils.py:109-140); the HIP kernels mask the
        tail themselves and never need padding.
        """
        n = self.n_rays
        n_pad = (-n) % int(block_size)
        if n_pad == 0:
            return self, 0

        def _pad(v: torch.Tensor) -> torch.Tensor:
            tail = v.new_zeros((n_pad,) + tuple(v.shape[1:]))
            return torch.cat([v, tail], dim=0)

        return self._map(_pad), n_pad

    def to(self, device, copy: bool = False) -> "Rays":
        """``torch.Tensor.to`` semantics for the whole container."""
        device = torch.device(device)
        if not copy and self.device == device:
            return self
        return self._map(lambda v: v.to(device, copy=copy))

    def clone(self) -> "Rays":
        return _copy.deepcopy(self)

    def contiguous(self) -> "Rays":
        return self._map(lambda v: v.contiguous())

    def shard(self, rank: int, world_size: int) -> "Rays":
        """Contiguous ray shard ``rank`` of ``world_size`` (multi-GPU ray sharding)."""
        n = self.n_rays
        per = (n + world_size - 1) // world_size
        lo, hi = min(rank * per, n), min((rank + 1) * per, n)
        return self[lo:hi]


def calc_harmonic_embedding(directions: torch.Tensor, n_harmonic_functions: int) -> torch.Tensor:
    """Harmonic (sin / cos) embedding of ``directions`` (``[..., 3]``).

    Output layout (reference ray_utils.py:181-212): ``[sin(d*2^k) for each
    coordinate, k] ++ [cos(...)] ++ directions`` -> ``[..., 6*n + 3]``;
    cos is evaluated as ``sin(x + pi/2)``.  ``n_harmonic_functions == 0``
    returns ``directions`` unchanged.
    """
    if n_harmonic_functions == 0:
        return directions
    freqs = 2.0 ** torch.arange(
        n_harmonic_functions, dtype=directions.dtype, device=directions.device
    )
    phase = torch.tensor([0.0, 0.5 * math.pi], dtype=directions.dtype, device=directions.device)
    # [..., 3, n] -> [..., 2(phase), 3, n]
    ang = directions.unsqueeze(-1) * freqs
    ang = ang.unsqueeze(-3) + phase.view(2, 1, 1)
    emb = torch.sin(ang).flatten(start_dim=-3)
    return torch.cat([emb, directions], dim=-1)


def calc_harmonic_embedding_dim(n_harmonic_functions: int) -> int:
    """Dimension of :func:`calc_harmonic_embedding`'s output."""
    return 3 + 6 * n_harmonic_functions


def jitter_near_far(near: torch.Tensor, far: torch.Tensor, num_samples: int):
    """Shift near/far by a common uniform offset in ``[-delta, delta]``,
    ``delta = (far - near) / num_samples`` (reference ray_utils.py:220-229)."""
    delta = (far - near) / num_samples
    shift = (torch.rand_like(near) * 2.0 - 1.0) * delta
    return near + shift, far + shift
