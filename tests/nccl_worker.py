"""Worker of tests/test_gpu_modules.py::test_rccl_collectives_on_a_one_rank_group: a ONE-rank `nccl` (= RCCL) process group on
cuda:0.  No 8-GPU node is available to the build, so this is where the RCCL-facing calls of lightplane_amd/parallel.py --
the private `_coalescing_manager` context, `reduce_scatter_tensor` / `all_gather_into_tensor` on views of one buffer, the
in-place all-reduce inside autograd's backward -- execute on ROCm at all before a multi-GPU run.  With one
rank every sum is the identity, so values must come back unchanged; what is tested is that the calls are accepted, run on
the stream and leave the buffers intact."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import lightplane_amd as lp  # noqa: E402
from lightplane_amd import parallel  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 1000))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    pg = dist.group.WORLD
    assert dist.get_backend(pg) == "nccl"
    gen = torch.Generator().manual_seed(0)

    # 1. the coalesced launch (RCCL group call through the private context manager): count that the context really was built
    calls = []
    cm = dist.distributed_c10d._coalescing_manager

    def counting_cm(*a, **k):
        calls.append(k)
        return cm(*a, **k)

    dist.distributed_c10d._coalescing_manager = counting_cm
    try:
        ts = [torch.randn(n, generator=gen).to(dev) for n in (786432 // 4, 4273, 7, 1)]
        want = [t.clone() for t in ts]
        assert parallel._coalesced_all_reduce_(ts, pg, async_op=False) == [None]
        works = parallel._coalesced_all_reduce_(ts, pg, async_op=True)
        for w in works:
            if w is not None:
                w.wait()
        torch.cuda.synchronize()
        assert len(calls) == 2 and all("async_ops" in k and "device" in k for k in calls), calls
        for a, b in zip(ts, want):
            assert torch.equal(a, b)
    finally:
        dist.distributed_c10d._coalescing_manager = cm

    # 2. reduce-scatter + all-gather in place on views of ONE buffer (what a >= 256 MB gradient takes), odd sizes included
    for n in (1, 5, 1 << 20, (1 << 20) + 3):
        t = torch.randn(n, generator=gen).to(dev)
        want = t.clone()
        parallel._big_allreduce_(t, pg)
        torch.cuda.synchronize()
        assert torch.equal(t, want), n
    assert parallel._supports_rs(t, pg)

    # 3. the real step: a ray-sharded Renderer forward + backward whose replicated grid / parameters are all-reduced by RCCL
    # inside autograd's backward (is_distributed() is false for one rank, so the step is told it is distributed), thresholds
    # lowered so that the small gradient takes the coalesced launch and the grid the reduce-scatter + all-gather path
    from tests.synth import RENDERER_CASES
    d = next(c for c in RENDERER_CASES if c.name == "triplane_plus_voxel").build()
    dec = d["decoder"]
    up = [u.to(dev) for u in d["upstream"]]

    def run(group):
        params = dec.mlp_params.to(dev).clone().requires_grad_(True)
        grids = [g.to(dev).clone().requires_grad_(True) for g in d["grids"]]
        leaves = grids + [params]
        rep = parallel.replicate_with_grad_allreduce(leaves, group, exclusive_grads=True) if group is not None else leaves
        hdec = lp.DecoderParams(rep[-1], dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
        out = lp.lightplane_renderer(d["rays"].to(dev), rep[:-1], hdec, **d["cfg"])
        ((out[0] * up[0]).sum() + (out[1] * up[1]).sum() + (out[2] * up[2]).sum()).backward()
        return [g.grad for g in grids], params.grad

    g1, p1 = run(None)
    old = parallel.is_distributed, parallel.RS_AG_BYTES, parallel.INPLACE_GRAD_BYTES, parallel.BUCKET_BYTES
    parallel.is_distributed = lambda process_group=None: True
    parallel.RS_AG_BYTES = parallel.INPLACE_GRAD_BYTES = 4096
    n0 = parallel._AllReduceGrad.inplace_reductions
    try:
        g2, p2 = run(pg)
    finally:
        parallel.is_distributed, parallel.RS_AG_BYTES, parallel.INPLACE_GRAD_BYTES, parallel.BUCKET_BYTES = old
    torch.cuda.synchronize()
    assert parallel._AllReduceGrad.inplace_reductions > n0
    for a, b in zip(g2 + [p2], g1 + [p1]):
        scale = float(b.abs().max())
        assert float((a - b).abs().max()) <= 2e-5 * scale  # (atomics reorder the sums between two runs)

    dist.barrier()
    print("RCCL_1RANK_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
