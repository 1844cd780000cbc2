"""CPU tests: host API types, the C-ABI library loads and exports every declared symbol, argument
validation at the C boundary (no kernel is launched without a GPU), multi-process collectives on gloo."""
import ctypes
import os
import re

import pytest
import torch

import lightplane_amd as lp
from lightplane_amd import _lib, grids, params


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.lib()
    assert L.lp_version() == 207
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "lightplane_hip.h")).read()
    declared = set(re.findall(r"^\s*(?:int|const char\*)\s+(lp_[a-z_0-9]+)\s*\(", hdr, flags=re.M))
    assert declared == set(_lib.EXPORTS), f"header vs binding mismatch: {declared ^ set(_lib.EXPORTS)}"
    for name in declared:
        assert hasattr(L, name), f"{name} not exported by liblightplane_hip.so"


def test_abi_struct_sizes_match():
    L = _lib.lib()
    for which, st in enumerate((_lib.LpGrid, _lib.LpGridList, _lib.LpRays, _lib.LpMarch, _lib.LpMlp,
                                _lib.LpRendererArgs, _lib.LpSplatterArgs, _lib.LpRayEmbedArgs)):
        assert L.lp_abi_sizeof(which) == ctypes.sizeof(st)
    assert L.lp_abi_sizeof(99) == -1


def _empty_renderer_args():
    a = _lib.LpRendererArgs()
    a.rays.n_rays = 0
    a.grid = _lib.make_grid_list(None, [grids.GridDesc(1, 4, 4, 4, 0)], 16, 64)
    a.march = _lib.make_march(8, 0, False, False, 1e-5)
    a.trunk = _lib.make_mlp([16, 32, 32], 0)
    a.opacity = _lib.make_mlp([32, 32, 1], params.mlp_numel([16, 32, 32]))
    a.color = _lib.make_mlp([32, 32, 16], params.mlp_numel([16, 32, 32]) + params.mlp_numel([32, 32, 1]))
    a.n_mlp_params = sum(params.mlp_numel(d) for d in ([16, 32, 32], [32, 32, 1], [32, 32, 16]))
    a.mlp_params = 0x1000  # never dereferenced: n_rays == 0
    a.color_chn = 3
    a.rays.encoding_dim = 32
    return a


def test_c_abi_argument_validation_without_gpu():
    L = _lib.lib()
    a = _empty_renderer_args()
    assert L.lp_renderer_forward(ctypes.byref(a), None) == 0  # zero rays: validated, nothing launched
    a.march.num_samples = 0
    assert L.lp_renderer_forward(ctypes.byref(a), None) == -1
    assert b"num_samples" in L.lp_last_error()
    a = _empty_renderer_args()
    a.n_mlp_params += 1
    assert L.lp_renderer_forward(ctypes.byref(a), None) == -1
    assert b"number of elements in mlp param" in L.lp_last_error()
    a = _empty_renderer_args()
    a.grid.grids[0].W = 1
    a.grid.grids[0].H = 1  # a "line" grid: neither voxel nor plane
    assert L.lp_renderer_forward(ctypes.byref(a), None) == -1
    a = _empty_renderer_args()
    a.rays.encoding_dim = 7
    assert L.lp_renderer_backward(ctypes.byref(a), None) == -1
    assert L.lp_renderer_forward(None, None) == -3
    # early termination: negative / NaN threshold rejected; the backward needs the checkpoint buffer
    a = _empty_renderer_args()
    a.stop_neg_log_t = -1.0
    assert L.lp_renderer_forward(ctypes.byref(a), None) == -1
    assert b"stop_neg_log_t" in L.lp_last_error()
    a.stop_neg_log_t = 9.0
    assert L.lp_renderer_forward(ctypes.byref(a), None) == 0
    assert L.lp_renderer_backward(ctypes.byref(a), None) == -3
    assert b"neg_log_t_ckpt" in L.lp_last_error()
    with pytest.raises(AssertionError):
        _lib.check(-1, "x")


def test_cpu_tensors_fail_loudly():
    from tests.synth import RENDERER_CASES
    d = RENDERER_CASES[0].build()
    with pytest.raises(_lib.LightplaneHipError, match="GPU only"):
        lp.lightplane_renderer(d["rays"], d["grids"], d["decoder"], **d["cfg"])
    with pytest.raises(NotImplementedError):
        lp.LightplaneRenderer(8, 3, 16, 32, use_naive_impl=True)


def test_rays_container():
    n = 21
    r = lp.Rays(torch.randn(n, 3), torch.randn(n, 3), torch.zeros(n, dtype=torch.long), torch.zeros(n),
                torch.ones(n), torch.randn(n, 5))
    p, n_pad = r.pad_to_block_size(16)
    assert n_pad == 11 and p.directions.shape == (32, 3) and p.encoding.shape == (32, 5)
    assert (p.far[n:] == 0).all()
    assert r[3:7].n_rays == 4 and r.to("cpu") is r and r.to("cpu", copy=True) is not r
    assert sum(r.shard(k, 4).n_rays for k in range(4)) == n
    with pytest.raises(AssertionError):
        lp.Rays(torch.randn(n, 3), torch.randn(n, 3), torch.zeros(n), torch.zeros(n), torch.ones(n))
    with pytest.raises(AssertionError):
        lp.Rays(torch.randn(n, 3), torch.randn(n + 1, 3), torch.zeros(n, dtype=torch.long), torch.zeros(n), torch.ones(n))
    e = lp.calc_harmonic_embedding(torch.randn(4, 3), 3)
    assert e.shape == (4, lp.calc_harmonic_embedding_dim(3)) == (4, 21)
    d = torch.randn(2, 3)
    e = lp.calc_harmonic_embedding(d, 2)
    assert torch.allclose(e[:, :6].reshape(2, 3, 2), torch.sin(d[..., None] * torch.tensor([1.0, 2.0])))
    assert torch.allclose(e[:, 6:12].reshape(2, 3, 2), torch.cos(d[..., None] * torch.tensor([1.0, 2.0])), atol=1e-6)
    assert torch.equal(e[:, 12:], d)


def test_param_layout_roundtrip():
    dec = lp.init_decoder_params("cpu", n_layers_opacity=2, n_layers_trunk=3, n_layers_color=2, input_chn=16,
                                 hidden_chn=32, color_chn=3, opacity_init_bias=-5.0)
    assert dec.mlp_params.numel() == (16 * 32 + 32 * 32 * 2 + 32 * 3) + (32 * 32 + 32 + 32 + 1) + (32 * 32 + 32 * 16 + 32 + 16)
    wt, bt, wo, bo, wc, bc = lp.flattened_decoder_params_to_list(
        dec.mlp_params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color)
    assert [tuple(w.shape) for w in wt] == [(16, 32), (32, 32), (32, 32)]
    assert tuple(wc[-1].shape) == (32, 16) and (wc[-1][:, 3:] == 0).all() and float(bo[-1]) == -5.0
    flat, nt, no, nc = lp.flatten_decoder_params(wt, bt, wo, bo, wc, bc)
    assert torch.equal(flat, dec.mlp_params) and nt.dtype == torch.int32
    assert lp.get_triton_function_input_dims(nt, no, nc) == (32, 32, 32, 3, 2, 2, 16)
    sp = lp.init_splatter_params("cpu", 3, 32, 64, 16)
    assert sp.n_hidden.tolist() == [32, 64, 64, 16]
    dec2 = lp.init_decoder_params("cpu", 2, 0, 2, input_chn=16, hidden_chn=32, use_separate_color_grid=True)
    assert dec2.n_hidden_trunk.numel() == 0 and dec2.n_hidden_opacity.tolist() == [16, 32, 1]
    with pytest.raises(AssertionError):
        lp.init_decoder_params("cpu", 2, 2, 2, use_separate_color_grid=True)


def test_grid_helpers():
    gl = [torch.randn(2, 1, 5, 7, 4), torch.randn(2, 6, 1, 7, 4), torch.randn(2, 6, 5, 1, 4)]
    flat, sizes = lp.flatten_grid(gl)
    assert flat.shape == (2 * (35 + 42 + 30), 4) and sizes.tolist() == [list(g.shape) for g in gl]
    back = lp.unflatten_grid(flat, sizes)
    assert all(torch.equal(a, b) for a, b in zip(gl, back))
    descs, C, rows = grids.make_grid_descs(sizes)
    assert [d.row_offset for d in descs] == [0, 70, 154] and rows == 214 and C == 4
    assert [d.kind for d in descs] == ["plane"] * 3
    with pytest.raises(ValueError):
        grids.make_grid_descs([[1, 1, 1, 8, 4]])
    with pytest.raises(NotImplementedError):
        grids.check_grid(tuple(gl))
    with pytest.raises(AssertionError):
        grids.check_grid(flat, None)
    g, cg, gs, cgs = grids.process_and_flatten_grid(gl, None)
    assert torch.equal(g, flat) and gs == sizes.tolist()


def test_module_state_dict_keys_match_reference():
    m = lp.LightplaneRenderer(num_samples=8, color_chn=3, grid_chn=16, mlp_hidden_chn=32, bg_color=1.0)
    assert sorted(m.state_dict().keys()) == ["bg_color", "harmonic_ray_embedding_linear.bias",
                                            "harmonic_ray_embedding_linear.weight", "mlp_params"]
    assert m.harmonic_ray_embedding_linear.weight.shape == (32, 21)
    with pytest.raises(ValueError):
        lp.LightplaneRenderer(8, 3, 16, 32, enable_direction_dependent_colors=False)
    m2 = lp.LightplaneRenderer(8, 3, 16, 32, enable_direction_dependent_colors=False, ray_embedding_num_harmonics=None)
    assert not hasattr(m2, "harmonic_ray_embedding_linear")
    s = lp.LightplaneSplatter(8, 32)
    assert len(s.state_dict()) == 0 and s.get_splatter_params() is None


def test_abi_v2_validation_without_gpu():
    """ABI 0.2 additions validate on the host: ray-embedding entry points, per-grid pointers, gradient-list consistency."""
    L = _lib.lib()
    e = _lib.LpRayEmbedArgs()
    e.n_rays, e.n_harmonics, e.out_dim = 0, 3, 32
    assert L.lp_ray_embedding_forward(ctypes.byref(e), None) == 0
    assert L.lp_ray_embedding_backward(ctypes.byref(e), None) == 0
    e.n_harmonics = 11
    assert L.lp_ray_embedding_forward(ctypes.byref(e), None) == -2 and b"n_harmonics" in L.lp_last_error()
    e.n_harmonics, e.n_rays = 3, 5
    assert L.lp_ray_embedding_forward(ctypes.byref(e), None) == -3  # NULL buffers with rays to process
    # a grid that carries its own pointer does not have to fit LpGridList.n_rows; one that does not, has to
    a = _empty_renderer_args()
    a.grid.n_rows = 1
    assert L.lp_renderer_forward(ctypes.byref(a), None) == -1 and b"outside the flat tensor" in L.lp_last_error()
    a.grid.grids[0].data = 0x1000
    assert L.lp_renderer_forward(ctypes.byref(a), None) == 0
    a = _empty_renderer_args()
    a.alpha_mode = 3
    assert L.lp_renderer_forward(ctypes.byref(a), None) == -1 and b"alpha_mode" in L.lp_last_error()


def test_backward_segments_query_without_gpu():
    """lp_renderer_backward_segments (ABI 0.2.1) looks at shapes only: blocks of LP_SEG_LEN = 8 samples (16 before 0.2.7) for a small
    batch of the default decoder shape with 16 channels, 1 everywhere else."""
    L = _lib.lib()
    a = _empty_renderer_args()
    a.rays.n_rays = 4096
    assert _lib.LP_SEG_LEN == 8
    for s, want in ((8, 1), (9, 2), (16, 2), (17, 3), (64, 8), (65, 9), (256, 32)):
        a.march.num_samples = s
        assert L.lp_renderer_backward_segments(ctypes.byref(a)) == want
    a.march.num_samples = 128
    a.rays.n_rays = 1 << 16
    assert L.lp_renderer_backward_segments(ctypes.byref(a)) == 1          # fills the GPU without it
    a.rays.n_rays = 4096
    a.march.num_samples_inf = 1
    assert L.lp_renderer_backward_segments(ctypes.byref(a)) == 1          # beyond-far samples
    a.march.num_samples_inf = 0
    a.stop_neg_log_t = 4.0
    assert L.lp_renderer_backward_segments(ctypes.byref(a)) == 1          # early termination
    a.stop_neg_log_t = 0.0
    a.kernel = _lib.LP_KERNEL_GENERIC
    assert L.lp_renderer_backward_segments(ctypes.byref(a)) == 1
    a.kernel = _lib.LP_KERNEL_AUTO
    assert L.lp_renderer_backward_segments(ctypes.byref(a)) == 16
    # zero rays with a prefix buffer: validated, nothing launched
    a.rays.n_rays = 0
    a.seg_prefix = 0x1000
    assert L.lp_renderer_forward(ctypes.byref(a), None) == 0


def test_backward_segments_python_query_covers_every_mfma_family():
    """``lp.backward_segments`` (shapes only, no GPU): the default shape with 16 / 32 channels, the flex and two-grid
    shapes and hidden width 64 (layer-looped family) march small batches in segments; the shape-generic kernels do not."""
    from tests.synth import grid_sizes_for, pinhole_rays, random_decoder, random_grids
    gen = torch.Generator().manual_seed(0)
    rays = pinhole_rays(32, 32, enc_dim=32, gen=gen)

    def q(C=16, layers=(2, 2, 2), hidden=32, sep=False, **kw):
        sizes = grid_sizes_for((1, 8, 8, 8, C), True)
        grids = random_grids(gen, sizes)
        cgrids = random_grids(gen, sizes) if sep else None
        dec = random_decoder(gen, *layers, input_chn=C, hidden_chn=hidden, color_chn=3, use_separate_color_grid=sep)
        return lp.backward_segments(rays, grids, dec, color_grid=cgrids, **dict(dict(num_samples=64), **kw))

    assert q() == 8 and q(C=32) == 8
    assert q(layers=(1, 1, 1), hidden=16) == 8          # flex
    assert q(layers=(0, 2, 2), sep=True) == 8           # two-grid decoder
    assert q(C=32, hidden=64) == 8                      # hidden 64: two-block looped kernels
    assert q(layers=(3, 2, 2)) == 8                     # layer-looped family
    assert q(num_samples=8) == 1 and q(num_samples_inf=2) == 1 and q(stop_transmittance=0.01) == 1
    assert lp.kernel_family(rays, random_grids(gen, grid_sizes_for((1, 8, 8, 8, 16), True)),
                            random_decoder(gen, 3, 2, 2, input_chn=16, hidden_chn=32, color_chn=3)) == 3


def _gloo_worker(rank, world_size, port, ret):
    import torch.distributed as dist
    from lightplane_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        # Splatter semantics: sum un-normalised features AND weights, then normalise
        feat = torch.full((6, 4), float(rank + 1))
        wgt = torch.full((6,), float(rank + 1) * 0.5)
        parallel.allreduce_sum_([feat, wgt, None])
        assert torch.allclose(feat, torch.full((6, 4), 3.0)) and torch.allclose(wgt, torch.full((6,), 1.5))
        # Renderer semantics: gradients of replicated inputs are summed over ray shards
        w = torch.ones(5, requires_grad=True)
        (wv,) = parallel.replicate_with_grad_allreduce([w])
        n = 10
        lo, hi = parallel.shard_bounds(n, rank, world_size)
        x = torch.arange(n, dtype=torch.float32)[lo:hi]
        (wv.sum() * x.sum()).backward()
        assert torch.allclose(w.grad, torch.full((5,), float(sum(range(n)))))
        # several replicated tensors (grid + parameters: one coalesced launch on RCCL, one collective each on gloo);
        # the tensors autograd hands to the all-reduce node are not modified in place (hooks see the local gradient)
        g = torch.ones(7, 3, requires_grad=True)
        p = torch.ones(4, requires_grad=True)
        gv, pv = parallel.replicate_with_grad_allreduce([g, p])
        seen = []
        gv.register_hook(lambda t: seen.append(t.clone()))
        ((gv.sum() + 2 * pv.sum()) * float(rank + 1)).backward()
        assert torch.allclose(g.grad, torch.full((7, 3), 3.0)) and torch.allclose(p.grad, torch.full((4,), 6.0))
        assert torch.allclose(seen[0], torch.full((7, 3), float(rank + 1)))
        odd = torch.full((11,), float(rank + 1))
        parallel.allreduce_sum_([odd])
        assert torch.allclose(odd, torch.full((11,), 3.0))
        # the reduce-scatter + all-gather path (payloads >= 256 MB on RCCL): its in-place view / tail arithmetic for
        # n % world in {0, 1, world - 1} and n < world, with gloo's own tensor collectives and with an emulation of the
        # reduce-scatter by an all-reduce (what a backend without reduce_scatter_tensor would be given)
        def rs_emulated(out, inp, group):
            tmp = inp.clone()
            dist.all_reduce(tmp, group=group)
            per = out.numel()
            out.copy_(tmp[dist.get_rank(group) * per:(dist.get_rank(group) + 1) * per])

        for n in (0, 1, world_size, 4 * world_size, 4 * world_size + 1, 5 * world_size - 1, 1000003):
            for rs in (None, rs_emulated):
                base = torch.arange(n, dtype=torch.float64) * 0.25
                t = (base * (rank + 1)).reshape(-1)
                parallel._big_allreduce_(t, None, reduce_scatter=rs)
                want = base * sum(r + 1 for r in range(world_size))
                assert torch.equal(t, want), (n, rs is None)
        # ... and through allreduce_sum_ when the thresholds select it (the backend check is per process group)
        old = parallel.RS_AG_BYTES, parallel._supports_rs
        parallel.RS_AG_BYTES, parallel._supports_rs = 64, (lambda t, pg=None: dist.get_backend(pg) == "gloo")
        try:
            big = torch.full((3, 7), float(rank + 1))
            parallel.allreduce_sum_([big, None, torch.full((2,), 1.0)])
            assert torch.allclose(big, torch.full((3, 7), 3.0))
        finally:
            parallel.RS_AG_BYTES, parallel._supports_rs = old
        # large gradients of replicated tensors: reduced in place (no clone) ONLY when the caller declares them exclusive.
        # (gv2 * w).sum() hands the node a contiguous buffer of its own (a .sum() alone gives a stride-0 expand, which takes
        # the clone path whatever the flag says)
        old_inplace = parallel.INPLACE_GRAD_BYTES
        parallel.INPLACE_GRAD_BYTES = 16
        try:
            wgt2 = torch.arange(18, dtype=torch.float32).reshape(9, 2) + 1.0
            n0 = parallel._AllReduceGrad.inplace_reductions
            g2 = torch.ones(9, 2, requires_grad=True)
            (gv2,) = parallel.replicate_with_grad_allreduce([g2], exclusive_grads=True)
            ((gv2 * wgt2).sum() * float(rank + 1)).backward()
            assert torch.allclose(g2.grad, wgt2 * 3.0)
            assert parallel._AllReduceGrad.inplace_reductions == n0 + 1       # no clone happened
            # default: the buffer is aliased here (autograd hands ONE tensor to both inputs of the add); the other consumer
            # has to keep its LOCAL gradient
            g3 = torch.ones(9, 2, requires_grad=True)
            delta = torch.zeros(9, 2, requires_grad=True)
            (gv3,) = parallel.replicate_with_grad_allreduce([g3])
            (((gv3 + delta) * wgt2).sum() * float(rank + 1)).backward()
            assert parallel._AllReduceGrad.inplace_reductions == n0 + 1       # cloned
            assert torch.allclose(g3.grad, wgt2 * 3.0) and torch.allclose(delta.grad, wgt2 * float(rank + 1))
        finally:
            parallel.INPLACE_GRAD_BYTES = old_inplace
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_ray_shard_allreduce_logic_on_gloo():
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs) and len(ret) == 2


def _gloo_worker8(rank, world_size, port, ret):
    """World size 8 (one node's worth of ranks): the reduce-scatter + all-gather path with tails that do not divide by 8 and the
    ray-shard bounds of 8 x 1080p -- the launch shape of BASELINE configs[3] / [4] (round-5 review, next 8)."""
    import torch.distributed as dist
    from lightplane_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world_size)
    try:
        tri = sum(r + 1 for r in range(world_size))
        for n in (0, 5, 8, 8 * 1000 + 7, 8 * 1000 + 1, 123457):
            base = torch.arange(n, dtype=torch.float64) * 0.5 - 3.0
            t = base * (rank + 1)
            parallel._big_allreduce_(t, None)
            assert torch.equal(t, base * tri), n
        old = parallel.RS_AG_BYTES, parallel._supports_rs
        parallel.RS_AG_BYTES, parallel._supports_rs = 64, (lambda t, pg=None: True)
        try:  # through allreduce_sum_, a [rows, C] grid whose element count is not a multiple of 8, next to a small parameter vector
            grid = torch.full((1001, 3), float(rank + 1))
            par = torch.full((19,), 2.0 * (rank + 1))
            parallel.allreduce_sum_([grid, par, None])
            assert torch.equal(grid, torch.full((1001, 3), float(tri))) and torch.equal(par, torch.full((19,), 2.0 * tri))
        finally:
            parallel.RS_AG_BYTES, parallel._supports_rs = old
        # 8 x 1080p rays over 8 ranks: contiguous, disjoint, complete; a count that does not divide leaves the last shard short
        for n in (8 * 1920 * 1080, 8 * 1920 * 1080 - 5, 3):
            lo, hi = parallel.shard_bounds(n, rank, world_size)
            sizes = torch.zeros(world_size, dtype=torch.int64)
            sizes[rank] = hi - lo
            dist.all_reduce(sizes)
            assert int(sizes.sum()) == n and int(sizes.max()) == -(-n // world_size)
            bounds = torch.zeros(world_size, 2, dtype=torch.int64)
            bounds[rank, 0], bounds[rank, 1] = lo, hi
            dist.all_reduce(bounds)
            assert bool((bounds[1:, 0] == bounds[:-1, 1]).all()) and int(bounds[0, 0]) == 0 and int(bounds[-1, 1]) == n
        # gradients of a replicated tensor summed over the 8 ray shards
        w = torch.ones(6, requires_grad=True)
        (wv,) = parallel.replicate_with_grad_allreduce([w])
        (wv.sum() * float(rank + 1)).backward()
        assert torch.equal(w.grad, torch.full((6,), float(tri)))
        ret[rank] = True
    finally:
        dist.destroy_process_group()


def test_ray_shard_allreduce_logic_on_gloo_world_size_8():
    import torch.multiprocessing as mp
    port = 31500 + (os.getpid() % 2000)
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_gloo_worker8, args=(r, 8, port, ret)) for r in range(8)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
    assert all(p.exitcode == 0 for p in procs) and len(ret) == 8


def test_build_info_names_what_the_binary_was_built_from():
    """lp_build_info(): the source hash of the tree the library was compiled from, its flags and the arithmetic of every backward
    family -- what bench.py copies into its record (`arithmetic`, `build`) instead of a hand-written string."""
    from lightplane_amd.csrc import build as B
    info = _lib.build_info()
    assert info["version"] == 207 and info["test_hooks"] in (0, 1)
    assert info["src_hash"] == B.source_hash(), "liblightplane_hip.so was built from other sources than this tree: run lightplane_amd/csrc/build.py"
    assert _lib.build_matches_tree() is True
    assert info["tuned_bwd"]["dx_limbs"] in (2, 3) and "v_mfma_f32_16x16x" in info["tuned_bwd"]["dw"]
    assert "-DLP_LOOP_DW_FP32" in info["flags"]["per_file"]["lp_renderer_loop_shallow.hip"]
    import bench
    text = bench.arithmetic_string(info)
    assert ("two-limb" in text) == (info["tuned_bwd"]["dx_limbs"] == 2)
    assert ("v_mfma_f32_16x16x32_bf16" in text) == ("bf16" in info["tuned_bwd"]["dw"])


def test_arithmetic_selection_and_dump_words_without_gpu():
    """LpRendererArgs.arithmetic: LP_ARITH_FP32 keeps the tuned family where it has such instantiations and selects the generic
    fp32 kernels elsewhere; invalid values are refused; lp_renderer_relu_dump_words per family."""
    from tests.synth import RENDERER_CASES
    from lightplane_amd.renderer import relu_dump_words
    by = {c.name: c for c in RENDERER_CASES}
    d = by["triplane_basic"].build()
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"]) == 1
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"], arithmetic=_lib.LP_ARITH_FP32) == 1
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"], arithmetic=_lib.LP_ARITH_FP32, num_samples_inf=65) == 0
    assert lp.kernel_family(d["rays"], d["grids"], d["decoder"], kernel=_lib.LP_KERNEL_GENERIC) == 0
    assert lp.backward_segments(d["rays"], d["grids"], d["decoder"], num_samples=64) > 1
    assert lp.backward_segments(d["rays"], d["grids"], d["decoder"], num_samples=64, arithmetic=_lib.LP_ARITH_FP32) == 1
    hooks = int(_lib.build_info()["test_hooks"])
    assert relu_dump_words(d["rays"], d["grids"], d["decoder"]) == (5 if hooks else 0)
    assert relu_dump_words(d["rays"], d["grids"], d["decoder"], arithmetic=_lib.LP_ARITH_FP32) == 0
    assert relu_dump_words(d["rays"], d["grids"], d["decoder"], kernel=_lib.LP_KERNEL_GENERIC) == (5 if hooks else 0)  # the generic twin
    dd = by["triplane_deep444"].build()      # 4 + 3 + 3 sites of one word
    assert lp.kernel_family(dd["rays"], dd["grids"], dd["decoder"]) == 3
    assert lp.kernel_family(dd["rays"], dd["grids"], dd["decoder"], arithmetic=_lib.LP_ARITH_FP32) == 0
    assert relu_dump_words(dd["rays"], dd["grids"], dd["decoder"]) == (11 if hooks else 0)
    dw = by["triplane_h64_c32"].build()      # 2 + 1 + 1 sites of two words
    assert relu_dump_words(dw["rays"], dw["grids"], dw["decoder"]) == (9 if hooks else 0)
    assert relu_dump_words(dw["rays"], dw["grids"], dw["decoder"], kernel=_lib.LP_KERNEL_GENERIC) == (9 if hooks else 0)  # ceil(64 / 32) words
    a = _empty_renderer_args()
    a.arithmetic = 7
    assert _lib.lib().lp_renderer_forward(ctypes.byref(a), None) == -1 and b"arithmetic" in _lib.lib().lp_last_error()


def test_march_order_heuristic_without_gpu():
    """march_order "auto": neighbouring rays (an image in scanline order) march rays per wavefront; unrelated rays -- random rays, and
    random PIXELS of one camera -- samples per wavefront; explicit orders are taken as given; no look when check_inputs is off."""
    from lightplane_amd.renderer import check_inputs_and_choose_march
    from tests.synth import pinhole_rays, random_rays
    gen = torch.Generator().manual_seed(0)
    img = pinhole_rays(96, 128, enc_dim=32, gen=gen)
    shuffled = img[torch.randperm(img.n_rays, generator=gen)]
    rnd = random_rays(gen, 2048, 3, 32)
    R, S = _lib.LP_MARCH_RAYS_PER_WAVE, _lib.LP_MARCH_SAMPLES_PER_WAVE
    gi = lambda r: r.grid_idx.to(torch.int32)  # noqa: E731
    assert lp.config.check_inputs
    assert check_inputs_and_choose_march(img, gi(img), 1) == R
    assert check_inputs_and_choose_march(shuffled, gi(shuffled), 1) == S
    assert check_inputs_and_choose_march(rnd, gi(rnd), 3) == S
    assert check_inputs_and_choose_march(rnd, gi(rnd), 3, "rays") == R and check_inputs_and_choose_march(img, gi(img), 1, "samples") == S
    with pytest.raises(AssertionError, match="out of bounds"):
        check_inputs_and_choose_march(rnd, gi(rnd), 2)
    with pytest.raises(AssertionError, match="march_order"):
        check_inputs_and_choose_march(rnd, gi(rnd), 3, "diagonal")
    try:
        lp.config.check_inputs = False
        assert check_inputs_and_choose_march(rnd, gi(rnd), 3) == R
    finally:
        lp.config.check_inputs = True


def test_forcer_reports_what_it_forced():
    """oracle.relu_mask_forcer measures what a forced-oracle proof is allowed to rest on: the largest relative |pre-activation| of a
    unit forced against the oracle's own sign, and how many units sit that close to zero."""
    from oracle import lightplane_oracle as O
    x = torch.tensor([[1.0, -2.0, 1e-7, -3e-7, 0.5, 0.0]], dtype=torch.float64)
    own = x > 0
    m = own.clone()
    m[0, 2] = False    # a near tie decided the other way
    with O.relu_mask_forcer([m], near_eps=1e-6) as f:
        y = O._relu(x)
    assert f.n_forced == 1 and abs(f.max_forced_margin - 1e-7 / 2.0) < 1e-12 and f.n_near_units == 2 and f.n_units == 6
    assert torch.equal(y, x * m.double())
    m[0, 0] = False    # a unit with a CLEAR sign forced: the margin says so
    with O.relu_mask_forcer([m], near_eps=1e-6) as f:
        O._relu(x)
    assert f.n_forced == 2 and f.max_forced_margin == 0.5


def test_mlp_splatter_host_checks():
    """lightplane_mlp_splatter validates shapes like the reference (asserts) and never runs on CPU."""
    from tests.synth import SPLATTER_CASES
    d = SPLATTER_CASES[4].build()
    with pytest.raises(_lib.LightplaneHipError, match="GPU only"):
        lp.lightplane_mlp_splatter(d["rays"], d["out_sizes"], d["mlp"], d["in_grids"], **d["cfg"])
    bad = lp.SplatterParams(d["mlp"].mlp_params[:-1], d["mlp"].n_hidden)
    with pytest.raises(AssertionError, match="number of elements in mlp param"):
        lp.lightplane_mlp_splatter(d["rays"], d["out_sizes"], bad, d["in_grids"], **d["cfg"])
    with pytest.raises(AssertionError):
        lp.lightplane_mlp_splatter(d["rays"], [[2, 6, 5, 7, 16]], d["mlp"], d["in_grids"], **d["cfg"])
    m = lp.LightplaneMLPSplatter(num_samples=8, grid_chn=16, input_grid_chn=32, mlp_hidden_chn=64, mlp_n_layers=3)
    assert set(m.state_dict().keys()) == {"mlp_params"}  # n_hidden is a non-persistent buffer (reference :232)
    assert m.get_splatter_params().n_hidden.tolist() == [32, 64, 64, 16]
    with pytest.raises(NotImplementedError):
        lp.LightplaneMLPSplatter(8, 16, use_naive_impl=True)


def test_mlp_splatter_c_abi_argument_errors():
    L = _lib.lib()
    a = _lib.LpSplatterArgs()
    a.rays.n_rays = 0
    a.march = _lib.make_march(4, 0, False, False, 1e-5)
    from lightplane_amd.grids import make_grid_descs
    descs, C, rows = make_grid_descs([[1, 4, 4, 4, 16]])
    a.out = _lib.make_grid_list(None, descs, C, rows)
    a.rays.encoding_dim = 32
    a.mlp = _lib.make_mlp([32, 32, 8], 0)  # output width 8 != 16 grid channels
    in_descs, Ci, rows_i = make_grid_descs([[1, 3, 3, 3, 32]])
    a.input_grid = _lib.make_grid_list(None, in_descs, Ci, rows_i)
    a.n_mlp_params = 32 * 32 + 32 * 8 + 32 + 8
    assert L.lp_splatter_forward(ctypes.byref(a), None) == -1
    assert b"output grid channels" in L.lp_last_error()
    a.mlp = _lib.make_mlp([32, 32, 16], 0)
    a.n_mlp_params = 5
    assert L.lp_splatter_forward(ctypes.byref(a), None) == -1
    assert b"number of elements in mlp param" in L.lp_last_error()
    a.n_mlp_params = 32 * 32 + 32 * 16 + 32 + 16
    assert L.lp_splatter_forward(ctypes.byref(a), None) == -3  # mlp_params NULL


def _family_with_color_grid(dec, sizes, csizes):
    """kernel_family for a flat-tensor colour grid given by its sizes only (no tensors needed for the query)."""
    from lightplane_amd.grids import make_grid_descs
    from lightplane_amd.params import mlp_numel
    from lightplane_amd.renderer import _decoder_dims
    descs, channels, n_rows = make_grid_descs(sizes)
    cdescs, _, c_rows = make_grid_descs(csizes)
    dims_t, dims_o, dims_c = _decoder_dims(dec)
    a = _lib.LpRendererArgs()
    a.grid = _lib.make_grid_list(None, descs, channels, n_rows)
    a.color_grid = _lib.make_grid_list(None, cdescs, channels, c_rows)
    n_t, n_o = mlp_numel(dims_t), mlp_numel(dims_o)
    a.trunk, a.opacity, a.color = _lib.make_mlp(dims_t, 0), _lib.make_mlp(dims_o, n_t), _lib.make_mlp(dims_c, n_t + n_o)
    a.color_chn = int(dec.color_chn)
    return int(_lib.lib().lp_renderer_kernel_family(ctypes.byref(a)))


def test_kernel_family_selection():
    """LP_KERNEL_AUTO picks the MFMA families for the shapes they are built for (no GPU needed): the headline
    configuration must never fall back to the shape-generic kernels silently."""
    from lightplane_amd.renderer import kernel_family
    from tests.synth import RENDERER_CASES, grid_sizes_for, random_decoder
    want = {"voxel_basic": 1, "triplane_basic": 1, "triplane_c32": 1, "voxel_c32_color1": 1, "triplane_h64_c32": 3,
            "voxel_h64_c16_scaffold": 3, "triplane_colorgrid": 3, "colorgrid_c32_h16_voxel": 3, "colorgrid_heads1_inf": 3, "colorgrid_c32_mixed": 3,
            "voxel_deep": 3, "triplane_h16_c32": 3, "color16": 3, "triplane_deep444": 3, "colorgrid_deep044": 3,
            "triplane_242_c32_color4": 3, "voxel_deep342_h64_c32": 0, "color16_deep323_h16": 3,
            # (2/2/2 x 64: the two-block looped kernels since 0.2.4 -- the fp32-MFMA family 2 was retired)
            # shallow decoders other than the tuned 2/2/2 x 32 shape: the layer-looped family's two-waves-per-SIMD backward
            # (family 1's fp32-MFMA flex / two-grid kernels were retired in round 4)
            "nb2_like_t2_o1_c1": 3, "nb1_like_h16_111": 3, "flex_121_h16_c32_noise": 3, "flex_212_c32_scaffold": 3,
            "triplane_c64_h32": 3, "voxel_c64_h64_112_scaffold": 3,
            "colorgrid_h64_c32_triplane": 3, "colorgrid_h64_c16_o1c2_inf": 3}  # two-grid x 64: since 0.2.4
    for c in RENDERER_CASES:
        if c.name in want:
            d = c.build()
            assert kernel_family(d["rays"], d["grids"], d["decoder"], color_grid=d["color_grids"]) == want[c.name], c.name
    # BASELINE cfg 2 and cfg 4 shapes
    gen = torch.Generator().manual_seed(0)
    for C, G in ((16, 64), (32, 128)):
        dec = random_decoder(gen, 2, 2, 2, C, 32, 3)
        assert kernel_family(None, None, dec, grid_sizes=grid_sizes_for((1, G, G, G, C), True)) == 1
    # every decoder shape of the reference's own sweep (tests/test_renderer_with_autograd.py:35-56: grid [3,16,12,8,16], hidden
    # 32, 3 colour channels, 2 or 4 layers per MLP, trunk 0 with a separate colour grid [4,3,9]) runs on the matrix cores
    import itertools
    for color_grid, tri, nt, no, nc in itertools.product([None, [4, 3, 9]], [False, True], [2, 4], [2, 4], [2, 4]):
        sep = color_grid is not None
        dec = random_decoder(gen, 0 if sep else nt, no, nc, 16, 32, 3, use_separate_color_grid=sep)
        sizes = grid_sizes_for((3, 16, 12, 8, 16), tri)
        csizes = grid_sizes_for((3, *color_grid, 16), tri) if sep else None
        fam = _family_with_color_grid(dec, sizes, csizes) if sep else kernel_family(None, None, dec, grid_sizes=sizes)
        assert fam in (1, 3), (color_grid, tri, nt, no, nc, fam)
    # splatter families through the C ABI
    L = _lib.lib()
    from lightplane_amd.grids import make_grid_descs
    a = _lib.LpSplatterArgs()
    descs, C, rows = make_grid_descs([[1, 128, 128, 128, 32]])
    a.out = _lib.make_grid_list(None, descs, C, rows)
    assert L.lp_splatter_kernel_family(ctypes.byref(a)) == 1
    a.mlp = _lib.make_mlp([32, 32, 32], 0)
    in_descs, Ci, rows_i = make_grid_descs([[1, 64, 64, 64, 32]])
    a.input_grid = _lib.make_grid_list(None, in_descs, Ci, rows_i)
    assert L.lp_splatter_kernel_family(ctypes.byref(a)) == 3  # (the two-layer fp32-MFMA family 2 was retired in round 4)
    a.mlp = _lib.make_mlp([32, 64, 64, 32], 0)
    assert L.lp_splatter_kernel_family(ctypes.byref(a)) == 3  # layer-looped family
    # every MLP shape of the reference's own Splatter sweep (tests/test_splatter_with_autograd.py:38-53: hidden 64, 3 / 4 layers,
    # 32 / 64 input features, 32 output channels) runs on the matrix cores
    for n_layers in (3, 4):
        for feat in (32, 64):
            a.mlp = _lib.make_mlp([feat] + [64] * (n_layers - 1) + [32], 0)
            in_descs, Ci, rows_i = make_grid_descs([[2, 10, 14, 16, feat]])
            a.input_grid = _lib.make_grid_list(None, in_descs, Ci, rows_i)
            assert L.lp_splatter_kernel_family(ctypes.byref(a)) == 3, (n_layers, feat)
    a.mlp = _lib.make_mlp([32, 64, 48, 32], 0)  # hidden widths differ: shape-generic kernels
    assert L.lp_splatter_kernel_family(ctypes.byref(a)) == 0
    # grid-lists of 4 GB and more stay on the matrix cores (lp_host.h grid_list_rows_ok: rows are 32-bit, bytes are not;
    # tests/test_gpu_large_grid.py runs them): a batch of four 256^3 x 32 scenes (8.6 GB), the layer-looped family on 17 GB of 16
    # channels, an MLP-Splatter reading 8.6 GB; 2^31 rows and more are the shape-generic kernels'
    dec = random_decoder(gen, 2, 2, 2, 32, 32, 3)
    assert kernel_family(None, None, dec, grid_sizes=[[4, 256, 256, 256, 32]]) == 1
    assert kernel_family(None, None, dec, grid_sizes=grid_sizes_for((40, 1024, 1024, 1024, 32), True)) == 1   # 3 x 5.4 GB of planes
    assert kernel_family(None, None, random_decoder(gen, 3, 2, 2, 16, 16, 3), grid_sizes=[[16, 256, 256, 256, 16]]) == 3
    assert kernel_family(None, None, random_decoder(gen, 2, 2, 2, 16, 32, 3), grid_sizes=[[128, 256, 256, 256, 16]]) == 0  # 2^31 rows
    from lightplane_amd.params import SplatterParams
    sp = SplatterParams(torch.zeros(1), torch.tensor([32, 32, 32]))
    assert lp.mlp_splatter_kernel_family([[4, 64, 64, 64, 32]], sp, [[4, 256, 256, 256, 32]]) == 3


def test_extension_and_absent_names():
    """stop_transmittance is validated on the host; the reference names that are deliberately absent say why."""
    from tests.synth import RENDERER_CASES
    d = RENDERER_CASES[0].build()
    with pytest.raises(AssertionError, match="stop_transmittance"):
        lp.lightplane_renderer(d["rays"], d["grids"], d["decoder"], stop_transmittance=1.5, **d["cfg"])
    assert lp.config.stop_transmittance == 0.0  # off by default: the exact march
    for name in ("lightplane_renderer_naive", "lightplane_splatter_naive", "lightplane_mlp_splatter_naive",
                 "visualize_rays_plotly"):
        with pytest.raises(AttributeError, match="deliberately not provided"):
            getattr(lp, name)
    assert _lib.n_nlt_ckpt(128, 0) == 2 * (4 + 1) and _lib.n_nlt_ckpt(33, 5) == 2 * (2 + 5 + 1)


def test_c_abi_header_is_plain_c(tmp_path):
    """include/lightplane_hip.h is the drop-in boundary: it has to compile as C99 (and C++11) on its own, and the struct
    sizes a C compiler sees must be the ones the library (hipcc) and the ctypes mirror use."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "abi.c"
    src.write_text('#include <stdio.h>\n#include "lightplane_hip.h"\nint main(void){printf("%zu %zu %zu %zu\\n", '
                   'sizeof(LpGridList), sizeof(LpRays), sizeof(LpRendererArgs), sizeof(LpSplatterArgs)); return 0;}\n')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(root, "include"),
                    str(src), "-o", str(exe)], check=True)
    subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-x", "c++", "-I", os.path.join(root, "include"), str(src)], check=True)
    sizes = [int(v) for v in subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE).stdout.split()]
    L = _lib.lib()
    assert sizes == [L.lp_abi_sizeof(1), L.lp_abi_sizeof(2), L.lp_abi_sizeof(5), L.lp_abi_sizeof(6)]
    assert sizes == [ctypes.sizeof(_lib.LpGridList), ctypes.sizeof(_lib.LpRays), ctypes.sizeof(_lib.LpRendererArgs),
                     ctypes.sizeof(_lib.LpSplatterArgs)]


def test_plain_c_binding_example(tmp_path):
    """examples/c_abi_example.c binds the library from C99 through the public header alone (dlopen), fills the
    benchmark decoder's descriptor and gets the MFMA family + clean argument validation back -- no GPU needed."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "c_abi_example"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-I", os.path.join(root, "include"),
                    os.path.join(root, "examples", "c_abi_example.c"), "-o", str(exe), "-ldl"], check=True)
    r = subprocess.run([str(exe), _lib.LIB_PATH], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()
    assert b"kernel family 1" in r.stdout and b"rc = -1" in r.stdout


def test_bench_counter_lookup_is_by_the_kernel_that_ran(tmp_path):
    """bench.py's `roofline.binding` / `traffic` must come from the counters of the kernel instantiation that ran, never
    from whichever summary file sorts last (round-3 review, weak 5: the LP_LOOP experiment's counters ended up in the line)."""
    import importlib.util
    import json

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lp_bench", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)

    bf3 = "lp::renderer_bwd_bf3<16, 1, true, 3, 4, false>"
    loop = "lp::renderer_bwd_loop<16, 1, false, 4, 3, false, 1>"
    # the committed summaries: the tuned kernel's entry is found by its exact name; experiment files are not consulted
    v, src, k = bench.pmc_entry("cfg2", bf3)
    assert v is not None and k == bf3 and re.search(r"r\d+_pmc_summary\.json$", src)
    assert bench.pmc_entry("cfg2", loop)[0] is None          # lives in r03loop_pmc_summary.json only: not a default-command file
    assert bench.pmc_entry("cfg2", "lp::renderer_bwd")[0] is None and bench.pmc_entry("cfg2", None)[0] is None  # no substring matches
    # observed kernels -> dominant by device time per step
    obs = {bf3: {"launches_per_step": 1, "mean_ms": 2.05}, "lp::renderer_fwd_bf3<16, 1, 2, 3, false>": {"launches_per_step": 1, "mean_ms": 0.47},
           "lp::renderer_bwd_combine": {"launches_per_step": 1, "mean_ms": 0.01}}
    assert bench.dominant_kernel(obs, "renderer_bwd") == bf3
    assert bench.dominant_kernel(obs, "splat_fwd_walk") is None

    class WL(bench.RendererWorkload):
        def __init__(self):
            from tests.synth import random_decoder
            self.n_rays, self.S, self.C, self.hidden, self.layers = 65536, 128, 16, 32, (2, 2, 2)
            self.dec_c = random_decoder(torch.Generator().manual_seed(0), 2, 2, 2, 16, 32, 3)

    b = bench.binding_ceiling("cfg2", WL(), 0.48, 2.07, obs)
    assert b["kernel"] == bf3 and 0.5 < b["frac_issue"] <= 1.0, b
    assert WL().dw_f32_mfma_per_launch() == 2048 * 128 * 112
    assert bench.binding_ceiling("cfg2", WL(), 0.48, 2.07, {loop: {"launches_per_step": 1, "mean_ms": 3.8}}) is None
    # newest round wins, an `rNNxyz_` experiment file never does
    e = {"FETCH_SIZE": 1.0, "WRITE_SIZE": 1.0, "hbm_bytes_per_launch": 2048, "SQ_INSTS_VALU": 4.0, "SQ_INSTS_MFMA": 1.0}
    for name, tag in (("r03_pmc_summary.json", 3), ("r04_pmc_summary.json", 4), ("r04zz_pmc_summary.json", 99), ("r10_pmc_summary.json", 10)):
        json.dump({"cfg2: " + bf3: dict(e, tag=tag)}, open(tmp_path / name, "w"))
    assert bench.pmc_entry("cfg2", bf3, profiles_dir=str(tmp_path))[0]["tag"] == 10


def _load_bench():
    import importlib.util
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("lp_bench", os.path.join(repo, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench, repo


def test_bench_binding_ceiling_for_every_extras_workload():
    """Round 4's driver run died on an `assert` in a helper of an EXTRAS leg as soon as counters for a hidden-64 kernel were
    committed.  For every workload bench.py can measure and every kernel the committed counter summaries hold for it:
    `binding_ceiling` + `relabel` run through on a stub workload with that workload's real decoder shape -- a dict or None,
    never an exception; the fp32 dW-MFMA count follows the decoder's layer list."""
    import glob
    import json
    from tests.synth import random_decoder

    bench, repo = _load_bench()
    kernels = {}
    for f in glob.glob(os.path.join(repo, "profiles", "r*_pmc_summary.json")):
        for k in json.load(open(f)):
            if ": " not in k:
                continue  # (round 1's summary has bare kernel names: never matched by a workload lookup)
            w, kn = k.split(": ", 1)
            kernels.setdefault(w, set()).add(kn)

    def stub(name):
        if name == "cfg3":
            wl = bench.SplatterWorkload.__new__(bench.SplatterWorkload)
            wl.n_rays, wl.S, wl.C, wl.G = 65536, 256, 32, 128
            return wl
        wl = bench.RendererWorkload.__new__(bench.RendererWorkload)
        H, W, S, C, G, wl.desc = bench.RENDER_CFGS[name]
        wl.name, wl.n_rays, wl.S, wl.C = name, H * W, S, C
        wl.layers, wl.hidden = bench.DECODER_SHAPES.get(name, ((2, 2, 2), bench.HIDDEN))
        wl.dec_c = random_decoder(torch.Generator().manual_seed(0), *wl.layers, C, wl.hidden, bench.COLOR)
        return wl

    names = list(bench.RENDER_CFGS) + ["cfg3"]
    assert {"h64_222", "h64_example_112", "cfg4", "1080p_s128", "small"} <= set(names)
    checked = 0
    for name in names:
        wl = stub(name)
        fwd_b, bwd_b = wl.algorithmic_bytes()
        assert fwd_b > 0 and bwd_b > 0
        for kn in sorted(kernels.get(name, ())) + ["lp::renderer_bwd_loop<99, 9, false, 9, 9, false, 9>", None]:
            obs = {kn: {"launches_per_step": 1, "mean_ms": 2.0}} if kn else {}
            roof = wl.roofline(1.0, 2.0)
            roof["binding"] = bench.binding_ceiling(name, wl, 1.0, 2.0, obs)   # must not raise
            bench.relabel(roof)
            b = roof["binding"]
            assert b is None or (isinstance(b, dict) and "error" not in b), (name, kn, b)
            if b is not None:
                checked += 1
                assert roof["bound"] == b["kind"] and roof["nominal_hbm_frac"] == roof["frac"] and roof["binding_frac"] > 0, roof
            else:
                assert roof["bound"] == "hbm"
            json.dumps(roof)
    assert checked >= 5, checked  # cfg2, 1080p, cfg4, small, h64_222, cfg3 have committed counters
    # the dW-MFMA count comes from the layer list (no shape assertion): 2/2/2 x 32 on C=16: 112, 2/2/2 x 64 on C=32: 448 per wave-sample
    assert stub("cfg2").dw_f32_mfma_per_launch() == 2048 * 128 * 112
    assert stub("h64_222").dw_f32_mfma_per_launch() == 2048 * 128 * 448
    assert stub("h64_example_112").dw_f32_mfma_per_launch() == 2048 * 128 * (32 * 64 + 64 * 64) // 32
    broken = stub("cfg2")
    broken.dec_c = None
    assert broken.dw_f32_mfma_per_launch() is None
    v = {"SQ_INSTS_VALU": 1000.0, "SQ_INSTS_MFMA": 202.0}
    assert bench.issue_bound(v, "lp::renderer_bwd_loop<1>", None) == (4000.0 + 112.0 * 32.0) / bench.N_SIMD  # the ratio fallback


def test_bench_extras_fail_soft_and_cover_every_leg():
    """Every extras leg runs inside `soft`: an exception becomes {"error": ...} and the other legs still run."""
    import json
    bench, _ = _load_bench()

    def boom(dev, k):
        raise AssertionError("leg failed")

    out = bench.run_extras(None, 0, legs=(("a", boom), ("b", lambda dev, k: {"ok": 1})))
    assert out["b"] == {"ok": 1} and "AssertionError" in out["a"]["error"]
    json.dumps(out)
    keys = [k for k, _ in bench.EXTRAS]
    assert len(keys) == len(set(keys)) and {"renderer_h64_222", "renderer_h64_example_112", "splatter_cfg3", "joint_cfg5_one_gpu"} <= set(keys)
    src = open(bench.__file__).read()
    main_src = src[src.index("def main():"):]
    assert "finally:" in main_src and main_src.index("finally:") < main_src.rindex("print(json.dumps(res), flush=True)")
    assert "assert self.hidden" not in src


def test_graphed_renderer_refuses_to_freeze_the_noise_seed():
    """The opacity-noise seed is a host scalar: captured into a HIP graph it would repeat one noise pattern at every replay
    (ADVICE round 3).  graphed_renderer refuses, before it touches the GPU."""
    from lightplane_amd.graphs import graphed_renderer
    from tests.synth import grid_sizes_for, random_grids, random_rays

    g = torch.Generator().manual_seed(0)
    grids = random_grids(g, grid_sizes_for((1, 4, 4, 4, 16), True))
    rays = random_rays(g, 32, 1, None)
    noisy = lp.LightplaneRenderer(num_samples=8, color_chn=3, grid_chn=16, mlp_hidden_chn=32, inject_noise_sigma=0.5)
    with pytest.raises(ValueError, match="inject_noise_sigma"):
        graphed_renderer(noisy, rays, grids)
    quiet = lp.LightplaneRenderer(num_samples=8, color_chn=3, grid_chn=16, mlp_hidden_chn=32)
    with pytest.raises(ValueError, match="inject_noise_sigma"):
        graphed_renderer(quiet, rays, grids, inject_noise_sigma=0.1)


def test_reference_submodule_import_paths():
    """`import lightplane_amd as lightplane` users also write `from lightplane.mlp_utils import DecoderParams`
    (reference tests/renderer_speed_benchmark.py:30), `lightplane.misc_utils.flatten_grid`, `lightplane.ray_utils.Rays`:
    the alias modules are re-exports of the same objects, and loading the two that are named like their functions leaves the
    package attributes callable (as in the reference's own __init__)."""
    import importlib

    for mod, names in (("mlp_utils", ["DecoderParams", "SplatterParams", "init_decoder_params", "flatten_decoder_params",
                                      "flattened_decoder_params_to_list", "get_triton_function_input_dims", "init_splatter_params"]),
                       ("misc_utils", ["flatten_grid", "unflatten_grid", "process_and_flatten_grid", "check_grid_and_color_grid"]),
                       ("ray_utils", ["Rays", "calc_harmonic_embedding", "jitter_near_far"]),
                       ("renderer_module", ["LightplaneRenderer"]), ("splatter_module", ["LightplaneSplatter", "LightplaneMLPSplatter"]),
                       ("lightplane_renderer", ["lightplane_renderer", "LightplaneFunction"]),
                       ("lightplane_splatter", ["lightplane_splatter", "lightplane_mlp_splatter", "LightplaneSplatterFunction"])):
        m = importlib.import_module("lightplane_amd." + mod)
        for n in names:
            assert getattr(m, n) is getattr(lp, n, getattr(m, n)), (mod, n)
            assert n in m.__all__
    import lightplane_amd.lightplane_renderer  # noqa: F401
    import lightplane_amd.lightplane_splatter  # noqa: F401
    assert callable(lp.lightplane_renderer) and callable(lp.lightplane_splatter)
    assert lp.mlp_utils.DecoderParams is lp.DecoderParams and lp.misc_utils.flatten_grid is lp.flatten_grid


def test_kernel_family_rejects_unknown_keywords():
    """``kernel_family`` / ``backward_segments`` take the render call's keywords (and ignore the ones that do not influence a
    shape-only answer) -- but a typo must not pass silently."""
    from tests.synth import grid_sizes_for, random_decoder, random_grids, random_rays

    g = torch.Generator().manual_seed(0)
    grids = random_grids(g, grid_sizes_for((1, 4, 4, 4, 16), True))
    rays = random_rays(g, 32, 1, 32)
    dec = random_decoder(g, 2, 2, 2, 16, 32, 3)
    assert lp.kernel_family(rays, grids, dec, num_samples=8, gain=1.0, mask_out_of_bounds_samples=True, inject_noise_sigma=0.1) == 1
    with pytest.raises(TypeError, match="num_sample_inf"):
        lp.kernel_family(rays, grids, dec, num_sample_inf=3)
    with pytest.raises(TypeError, match="num_sample"):
        lp.backward_segments(rays, grids, dec, num_samples=64, num_sample_inf=3)
    assert lp.backward_segments(rays, grids, dec, num_samples=64, gain=1.0) >= 1


def test_limb_image_layout_matches_the_bank_model(tmp_path):
    """scripts/lds_bank_model.py argues about LDS bank conflicts with a Python restatement of `rm_off` (lp_bf3.h: the layout of the
    row-major limb images and of the limb tiles of the weight-gradient products).  The shipped header and the model must be the same
    function: a host program compiled from the header prints rm_off for every (row, column); the model's algebra checks (bijection,
    16 contiguous bytes per backward lane, 8 per forward supplier lane) then hold for the shipped layout."""
    import shutil
    import subprocess
    import sys
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "rm_off_dump.hip"
    src.write_text('#include <cstdio>\n#include "lp_bf3.h"\nint main() {\n  for (int k = 0; k < 32; ++k)\n    for (int m = 0; m < 32; ++m) '
                   'printf("%d %d %d\\n", k, m, lp::rm_off(k, m));\n  printf("bytes %d %d\\n", lp::rm_bytes(16), lp::rm_bytes(32));\n  return 0;\n}\n')
    exe = tmp_path / "rm_off_dump"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(repo, "lightplane_amd", "csrc"), str(src), "-o", str(exe)],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = subprocess.run([str(exe)], check=True, stdout=subprocess.PIPE, timeout=60).stdout.decode().split("\n")
    sys.path.insert(0, os.path.join(repo, "scripts"))
    import lds_bank_model as M
    got = {}
    for line in out:
        p = line.split()
        if len(p) == 3 and p[0] != "bytes":
            got[(int(p[0]), int(p[1]))] = int(p[2])
    assert len(got) == 32 * 32 and all(M.rm_off(k, m) == v for (k, m), v in got.items())
    assert [l for l in out if l.startswith("bytes")] == [f"bytes {16 * 64 + 4 * 16} {32 * 64 + 8 * 16}"]
    # and the model's verdicts for that layout: forward transposed reads and backward ds_read_b128 conflict-free
    base = 1440
    for c in (0, 1):
        assert M.extra(M.G64, lambda l: base + M.rm_off(16 * c + 4 * (l >> 5) + ((l & 15) >> 2), (l & 16) + 4 * (l & 3)), 8) == 0
        assert M.extra(M.G128, lambda l: base + M.rm_off(l & 31, 16 * c + 4 * (l >> 5)), 16) == 0
        assert M.extra(M.G64, lambda l: base + M.rm_off_r4(16 * c + 4 * (l >> 5) + ((l & 15) >> 2), (l & 16) + 4 * (l & 3)), 8) == 2  # rounds 2-4
