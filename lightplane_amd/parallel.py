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
#: With ``replicate_with_grad_allreduce(..., exclusive_grads=True)`` gradients of at least this size are reduced IN PLACE, in
#: the buffer autograd hands to the all-reduce node, instead of in a private clone (cfg 5: the 2.15 GB gradient of the
#: 256^3 x 32 grid on a 19 GB peak).  That is only correct when nothing else reads that buffer: autograd hands ONE tensor to
#: both inputs of an add (``replicated + delta``), AccumulateGrad may have stolen it as somebody's ``.grad``, a hook or
#: retain_grad() may hold it -- every such reader would see the all-rank sum (or a half-reduced buffer) instead of its local
#: gradient.  The node cannot prove exclusivity from inside ``backward``, so the caller states it: the default is a clone.
INPLACE_GRAD_BYTES = RS_AG_BYTES


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


def _big_allreduce_(t: torch.Tensor, process_group, reduce_scatter=None, all_gather=None) -> None:
    """reduce-scatter + all-gather on a flat tensor, in place and without a padded copy: the part divisible by the
    world size goes through the two collectives (every rank reduces 1/world of it over all its xGMI links), the
    remainder (< world elements) through a plain all-reduce.

    ``reduce_scatter(out_shard, inp, group)`` / ``all_gather(out, in_shard, group)`` default to the torch.distributed
    tensor collectives; they are parameters so that the view / tail arithmetic can be exercised with other
    implementations (tests/test_host_logic.py runs it on gloo with the native ones and with an all-reduce emulation)."""
    if reduce_scatter is None:
        reduce_scatter = lambda out, inp, group: dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group)  # noqa: E731
    if all_gather is None:
        all_gather = lambda out, inp, group: dist.all_gather_into_tensor(out, inp, group=group)  # noqa: E731
    ws = dist.get_world_size(process_group)
    rank = dist.get_rank(process_group)
    flat = t.view(-1)
    n = flat.numel()
    per = n // ws
    if per > 0:
        head = flat[: per * ws]
        shard = head[rank * per : (rank + 1) * per]  # a view: the reduced shard lands where it belongs
        reduce_scatter(shard, head, process_group)
        all_gather(head, shard, process_group)
    if per * ws != n:
        dist.all_reduce(flat[per * ws :], op=dist.ReduceOp.SUM, group=process_group)


def _coalesced_all_reduce_(tensors: List[torch.Tensor], process_group, async_op: bool = False):
    """One launch for several all-reduces (RCCL group call), no flattening copy.  Falls back to one collective per
    tensor where the backend has no coalescing (gloo: the CPU test path)."""
    if len(tensors) == 1:
        return [dist.all_reduce(tensors[0], op=dist.ReduceOp.SUM, group=process_group, async_op=async_op)]
    backend = dist.get_backend(process_group)
    cm = getattr(dist.distributed_c10d, "_coalescing_manager", None)
    if backend == "nccl" and cm is not None and tensors[0].is_cuda:
        try:  # only the call that builds the context may fail softly (private API, its signature has moved before); an
            # error INSIDE the group call must surface -- retrying tensor by tensor would sum some of them twice
            group_call = cm(group=process_group, device=tensors[0].device, async_ops=async_op)
        except TypeError:
            group_call = None
        if group_call is not None:
            with group_call as handle:
                for t in tensors:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group)
            return [handle] if async_op else [None]
    return [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group, async_op=async_op) for t in tensors]


def allreduce_sum_(tensors: Sequence[Optional[torch.Tensor]], process_group=None, async_op: bool = False):
    """In-place sum of each tensor over the ranks of ``process_group``.

    Tensors up to ``BUCKET_BYTES`` go out together as ONE coalesced collective launch (no flatten / copy-back: the
    collective works on the tensors where they are); very large ones go through reduce-scatter + all-gather.
    No-op when not distributed.  With ``async_op`` the coalesced part is returned as a list of work handles to
    ``wait()`` on (the large tensors are reduced synchronously)."""
    if not is_distributed(process_group):
        return []
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    small: List[torch.Tensor] = []
    works = []
    for t in tensors:
        assert t.is_contiguous(), "all-reduce works in place: tensors have to be contiguous"
        nbytes = t.numel() * t.element_size()
        if nbytes >= RS_AG_BYTES and _supports_rs(t, process_group):
            _big_allreduce_(t, process_group)
        elif nbytes <= BUCKET_BYTES:
            small.append(t)
        else:
            works.append(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=process_group, async_op=async_op))
    if small:
        works += _coalesced_all_reduce_(small, process_group, async_op)
    return [w for w in works if w is not None] if async_op else []


def _supports_rs(t: torch.Tensor, process_group=None) -> bool:
    # the backend of THIS group decides (a gloo sub-group under an nccl default group, or the other way round); the
    # explicit two-step path only pays on RCCL / xGMI, every other backend keeps its plain all_reduce
    return t.is_cuda and dist.get_backend(process_group) == "nccl"


class _AllReduceGrad(torch.autograd.Function):
    """Identity in forward; sums the gradient over ranks in backward.

    Wrap the *replicated* inputs of a ray-sharded Renderer call (grid, colour grid, MLP
    parameters) so that ``.grad`` on every rank is the gradient of the whole (all-rank) loss.
    """

    @staticmethod
    def forward(ctx, process_group, exclusive_grads, *tensors):
        ctx.process_group = process_group
        ctx.exclusive_grads = bool(exclusive_grads)
        return tuple(t.view_as(t) for t in tensors)

    @staticmethod
    def backward(ctx, *grads):
        # never reduce into the buffers autograd handed us (they may be shared with the other input of an add, with a
        # stolen .grad, with hooks / retain_grad): own copies -- unless the caller declared the gradients exclusive
        # (`exclusive_grads`), and then only for INPLACE_GRAD_BYTES or more, where the clone would double a multi-GB buffer
        def own(g):
            if g is None:
                return None
            if ctx.exclusive_grads and g.is_contiguous() and g.numel() * g.element_size() >= INPLACE_GRAD_BYTES:
                _AllReduceGrad.inplace_reductions += 1
                return g
            return g.clone(memory_format=torch.contiguous_format)

        grads = [own(g) for g in grads]
        # Synchronous on the STREAM only (the host does not block).  It cannot be deferred past this node: autograd
        # accumulates the returned tensors into .grad right away, on this stream.  Overlap with the ray-embedding
        # backward comes from the node order instead: wrap the replicated tensors BEFORE the module computes the ray
        # embedding, then the embedding's backward nodes (created later) run before this one.
        allreduce_sum_(grads, ctx.process_group)
        return (None, None, *grads)


_AllReduceGrad.inplace_reductions = 0  # how many gradients were reduced without a clone (tests)


def replicate_with_grad_allreduce(tensors: Iterable[torch.Tensor], process_group=None, exclusive_grads: bool = False):
    """Return views of ``tensors`` whose gradients are all-reduced (sum) across ranks.

    ``exclusive_grads=True`` is the caller's statement that each returned view is consumed by exactly ONE differentiable
    op that writes its gradient into a buffer of its own (this package's Renderer / Splatter calls do), with no hook or
    ``retain_grad()`` on the view: gradients of ``INPLACE_GRAD_BYTES`` or more are then summed in that buffer instead of
    in a clone.  Do not set it when a view feeds an add, a sum of losses or any op that forwards its incoming gradient
    unchanged -- autograd shares one tensor between such an op's inputs."""
    tensors = list(tensors)
    if not is_distributed(process_group):
        return tensors
    return list(_AllReduceGrad.apply(process_group, exclusive_grads, *tensors))
