"""HIP-graph replay of a whole training step's Renderer call (small batches are host-bound: DESIGN.md 4.7).

A NeRF-style step renders a few thousand rays; the kernels of such a call take tens of microseconds while the host side of a
forward + backward (argument checks, descriptor filling, ~10 allocations, the autograd bookkeeping) takes ~0.3 ms.  The
calls of this package have no host synchronisation (with ``config.check_inputs`` off) and launch on the current stream, so a
forward + backward captures into a HIP graph.  ``graphed_renderer`` does the capture with ``torch.cuda.make_graphed_callables``
and hands back a callable with the module's own signature for FIXED shapes: same number of rays, same grid shapes, same
keyword configuration; tensor VALUES (rays, grids, module parameters) are free to change between calls.

The reference has no counterpart (its Triton launches read sizes on the device and synchronise).
"""
from __future__ import annotations

from typing import Sequence

import torch

from . import config
from .rays import Rays


def graphed_renderer(module: torch.nn.Module, rays: Rays, feature_grid: Sequence[torch.Tensor], num_warmup_iters: int = 3,
                     **forward_kwargs):
    """Capture ``module(rays, feature_grid, **forward_kwargs)`` -- a ``LightplaneRenderer`` -- and its backward into HIP graphs.

    ``rays`` / ``feature_grid`` are sample inputs of the shapes (and requires-grad states) every later call will have; the grids
    are a LIST of ``[B, D, H, W, C]`` tensors.  Returns ``fn(rays, feature_grid) -> (ray_length, alpha, feature)``: the replayed
    forward, differentiable like the eager call (gradients reach the grids, ``rays.encoding`` if given, and the module's
    parameters).  Rays without an encoding are supported (the module computes its ray embedding inside the graph).

    ``config.check_inputs`` must be off while capturing and replaying (the ``grid_idx`` range check is a host sync); it is
    switched off for the capture and the returned callable asserts it stays off.

    Every host-side SCALAR of the call is frozen into the graph at capture time: ``num_samples``, ``gain``, ``bg_color`` given as
    numbers, the switches -- and the opacity-noise seed, which is a by-value kernel argument drawn on the host
    (``renderer.py``: ``random.randint`` when no seed is given).  A replay would therefore inject the SAME noise pattern at every
    step, which silently defeats the regulariser (eager mode and the reference draw a fresh seed per call), so capturing with
    ``inject_noise_sigma > 0`` (module attribute or keyword) is refused.
    """
    sigma = forward_kwargs.get("inject_noise_sigma")
    if sigma is None:
        sigma = getattr(module, "inject_noise_sigma", 0.0)
    if sigma and float(sigma) > 0.0:
        raise ValueError("graphed_renderer: inject_noise_sigma > 0 cannot be captured -- the noise seed is a host scalar frozen into "
                         "the graph, every replay would repeat one noise pattern; train with noise in eager mode")
    assert isinstance(feature_grid, (list, tuple)), "graphed_renderer takes the grid-list as a list of tensors"
    has_enc = rays.encoding is not None
    n_grids = len(feature_grid)

    def flat_call(directions, origins, near, far, grid_idx, *rest):
        enc = rest[0] if has_enc else None
        grids = list(rest[1 if has_enc else 0:])
        r = Rays(directions=directions, origins=origins, grid_idx=grid_idx, near=near, far=far, encoding=enc)
        return module(r, grids, **forward_kwargs)

    class _Wrap(torch.nn.Module):  # make_graphed_callables treats a Module's parameters as graph inputs too
        def __init__(self):
            super().__init__()
            self.inner = module

        def forward(self, *args):
            return flat_call(*args)

    def flatten(r: Rays, grids):
        args = [r.directions, r.origins, r.near, r.far, r.grid_idx]
        if has_enc:
            args.append(r.encoding)
        return tuple(args) + tuple(grids)

    old = config.check_inputs
    config.check_inputs = False
    try:
        graphed = torch.cuda.make_graphed_callables(_Wrap(), flatten(rays, feature_grid), num_warmup_iters=num_warmup_iters)
    finally:
        config.check_inputs = old

    def fn(r: Rays, grids):
        assert len(grids) == n_grids and (r.encoding is not None) == has_enc, "graphed_renderer: fixed input structure"
        assert not config.check_inputs, "graphed_renderer: config.check_inputs has to stay off (its range check is a host sync)"
        return graphed(*flatten(r, grids))

    return fn
