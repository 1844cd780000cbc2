"""Multi-GPU layer: ray sharding + RCCL all-reduce (one process per GPU, torch.distributed).

The reference has no distributed code at all.  Rays are independent, so the
path shards embarrassingly: every rank holds a contiguous ray shard, the grid-list and the MLP
parameters are replicated, and exactly one exchange step per direction is needed:

* Renderer backward: ``grad_feature_grid`` (+ ``grad_color_feature_grid``, ``grad_mlp_params``)
  are partial sums over the local rays -> all-reduce(sum).  ``grad_rays_enc`` stays local.
* Splatter forward: the UN-normalised feature grid and the weight grid are partial sums ->
  all-reduce(sum) of both, THEN ``feat / clamp(weight, 1e-5)`` on every rank (normalising
  before the reduce would be wrong).  Splatter backward needs no collective.

``backend="nccl"`` is RCCL on ROCm (xGMI inside a node).  The helpers also run on ``gloo``
with CPU tensors, which is how the collective *logic* is tested without GPUs.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

import torch
import torch.distributed as dist

#: all-reduce payloads up to this many bytes are flattened into one bucket (latency bound on
#: xGMI: cfg-4 sends 6.3 MB + 19 KB -> one collective instead of three)
BUCKET_BYTES = 64 << 20
#: above this size use reduce-scatter + all-gather explicitly (uses all 7 xGMI links of a GPU)
RS_AG_BYTES = 256 << 20


def is_distributed(process_group=None) -> bool:
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1


def shard_bounds(n: int, rank: int, world_size: int):
    """Contiguous shard [lo, hi) of n items for ``rank`` (keeps image-space ray coherence)."""
    per = (n + world_size - 1) // world_size
    return min(rank * per, n), min((rank + 1) * per, n)


def shard_rays(rays, process_group=None):
    """This rank's contiguous ray shard."""
    if not is_distributed(process_group):
        return rays
    return rays.shard(dist.get_rank(process_group), dist.get_world_size(process_group))


def _big_allreduce_(t: torch.Tensor, process_group) -> None:
    """reduce-scatter + all-gather on a flat tensor (pads to a multiple of world_size)."""
    ws = dist.get_world_size(process_group)
    flat = t.view(-1)
    n = flat.numel()
    per = (n + ws - 1) // ws
    if per * ws != n:
        buf = flat.new_zeros(per * ws)
        buf[:n] = flat
    else:
        buf = flat
    shard = buf.new_empty(per)
    dist.reduce_scatter_tensor(shard, buf, op=dist.ReduceOp.SUM, group=process_group)
    dist.all_gather_into_tensor(buf, shard, group=process_group)
    if buf.data_ptr() != flat.data_ptr():
        flat.copy_(buf[:n])


def allreduce_sum_(tensors: Sequence[Optional[torch.Tensor]], process_group=None) -> None:
    """In-place sum of each tensor over the ranks of ``process_group``.

    Small tensors are coalesced into one bucket (one collective); very large ones go through
    reduce-scatter + all-gather.  No-op when not distributed.
    """
    if not is_distributed(process_group):
        return
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    small: List[torch.Tensor] = []
    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if nbytes >= RS_AG_BYTES and t.is_contiguous() and _supports_rs(t):
            _big_allreduce_(t, process_group)
        elif nbytes <= BUCKET_BYTES:
            small.append(t)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
    if len(small) == 1:
        dist.all_reduce(small[0], op=dist.ReduceOp.SUM, group=process_group)
    elif small:
        flat = torch.cat([t.reshape(-1) for t in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=process_group)
        pos = 0
        for t in small:
            n = t.numel()
            t.copy_(flat[pos : pos + n].view_as(t))
            pos += n


def _supports_rs(t: torch.Tensor) -> bool:
    # gloo has no reduce_scatter_tensor; keep the CPU test path on plain all_reduce
    return t.is_cuda


class _AllReduceGrad(torch.autograd.Function):
    """Identity in forward; sums the gradient over ranks in backward.

    Wrap the *replicated* inputs of a ray-sharded Renderer call (grid, colour grid, MLP
    parameters) so that ``.grad`` on every rank is the gradient of the whole (all-rank) loss.
    """

    @staticmethod
    def forward(ctx, process_group, *tensors):
        ctx.process_group = process_group
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        grads = [None if g is None else g.contiguous() for g in grads]
        allreduce_sum_(grads, ctx.process_group)
        return (None, *grads)


def replicate_with_grad_allreduce(tensors: Iterable[torch.Tensor], process_group=None):
    """Return views of ``tensors`` whose gradients are all-reduced (sum) across ranks."""
    tensors = list(tensors)
    if not is_distributed(process_group):
        return tensors
    return list(_AllReduceGrad.apply(process_group, *tensors))
