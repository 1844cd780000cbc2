"""Grid-lists of 4 GB and more stay on the MFMA kernels.

The MFMA kernels index grid ROWS with 32-bit integers and address them as 64-bit base + row * C * 4: the byte size of a grid-list is
not limited (lp_host.h grid_list_rows_ok; rounds 2-6 sent every grid-list of 4 GB or more to the shape-generic kernels, ~50x slower,
because ONE scatter walk formed a 32-bit byte offset).  A batch of 256^3 x 32 grids is 2.1 GB per element -- batches of them are what
288 GB of HBM are for.

Property used (no CPU oracle can hold these tensors): batch element b of a grid-list [B, D, H, W, C] is an independent scene.  Rays
that point at the LAST element of a batch whose earlier elements fill the first 4 GB have to
  * produce bit-identical outputs to the same rays on a grid-list that holds only that element (which is below 4 GB and is held to the
    oracle by the rest of the suite) -- the forward is deterministic;
  * leave the same gradient in that element (to the order of the fp32 atomics) and EXACT zeros everywhere else -- an address that wrapped
    at 2^32 lands in an earlier element.
"""
import pytest
import torch

import lightplane_amd as lp
from lightplane_amd import _lib
from tests.synth import pinhole_rays, random_decoder, random_splatter_mlp

pytestmark = pytest.mark.gpu
GB4 = 1 << 32


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _batch_for(shape_1, pad=1):
    """smallest batch whose LAST element starts at or beyond 4 GB (+ pad more elements)"""
    per = 4
    for v in shape_1:
        per *= v
    return (GB4 + per - 1) // per + pad


def _big_and_small(gen, shape_1, batch, dev):
    """[1, ...] random scene and the [batch, ...] tensor whose last element is that scene, everything else zero"""
    small = torch.randn(1, *shape_1, generator=gen).to(dev)
    big = torch.zeros(batch, *shape_1, device=dev)
    big[batch - 1].copy_(small[0])
    return big, small


def _rays(h, w, b, enc_dim, dev, az=25.0, el=20.0):
    gen = torch.Generator().manual_seed(5)
    r = pinhole_rays(h, w, enc_dim=enc_dim, gen=gen, grid_idx=b, azimuth_deg=az, elevation_deg=el)
    return lp.Rays(directions=r.directions.to(dev), origins=r.origins.to(dev), grid_idx=r.grid_idx.to(dev).int(), near=r.near.to(dev),
                   far=r.far.to(dev), encoding=r.encoding.to(dev).clone().requires_grad_(True))


def _render(rays, grids, dec, dev, cfg, up, **extra):
    params = dec.mlp_params.to(dev).clone().requires_grad_(True)
    hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
    gs = [g.clone().requires_grad_(True) for g in grids] if isinstance(grids, list) else grids.clone().requires_grad_(True)
    out = lp.lightplane_renderer(rays, gs, hdec, **cfg, **extra)
    ((out[0] * up[0]).sum() + (out[1] * up[1]).sum() + (out[2] * up[2]).sum()).backward()
    return out, params.grad, rays.encoding.grad, ([g.grad for g in gs] if isinstance(gs, list) else gs.grad)


def _close(name, got, want, tol=2e-5):
    got, want = got.detach(), want.detach()
    sc = float(want.abs().max()) + 1e-30
    e = float((got - want).abs().max()) / sc
    assert e <= tol, f"{name}: {e:.3e} of the largest entry"


CFG = dict(num_samples=96, gain=1.0, num_samples_inf=0, mask_out_of_bounds_samples=True, contract_coords=False,
           inject_noise_sigma=0.0, inject_noise_seed=0)

RENDER_CASES = {
    # name: (shapes of ONE batch element, decoder (layers, hidden), expected kernel family)
    # voxel column walk (lp_splat_walk.h) of the tuned family
    "tuned_voxel128_c32": ([(128, 128, 128, 32)], ((2, 2, 2), 32), 1),
    # two planes = a grid-list that is neither a canonical triplane nor a voxel grid: the per-slot walk (flush_run), the one that wrapped
    "tuned_two_planes_c32": ([(1, 768, 768, 32), (64, 1, 64, 32)], ((2, 2, 2), 32), 1),
    # layer-looped family, 16 channels
    "looped_voxel128_c16_h16": ([(128, 128, 128, 16)], ((1, 1, 1), 16), 3),
}


@pytest.mark.parametrize("name", list(RENDER_CASES))
def test_renderer_beyond_4gb(name):
    dev = _dev()
    shapes, (layers, hidden), family = RENDER_CASES[name]
    C = shapes[0][-1]
    batch = max(_batch_for(s) for s in shapes[:1])
    gen = torch.Generator().manual_seed(11)
    pairs = [_big_and_small(gen, s, batch, dev) for s in shapes]
    big, small = [p[0] for p in pairs], [p[1] for p in pairs]
    assert big[0].numel() * 4 >= GB4 and (batch - 1) * small[0].numel() * 4 >= GB4
    dec = random_decoder(gen, *layers, C, hidden, 3, std=0.15)
    h, w = 48, 64
    up = tuple(t.to(dev) for t in (torch.randn(h * w, generator=gen), torch.randn(h * w, generator=gen), torch.randn(h * w, 3, generator=gen)))
    r_big, r_small = _rays(h, w, batch - 1, hidden, dev), _rays(h, w, 0, hidden, dev)
    assert lp.kernel_family(r_big, big, dec) == family, "a grid-list of 4 GB or more left the MFMA family"
    assert lp.kernel_family(r_small, small, dec) == family
    o_b, gp_b, ge_b, gg_b = _render(r_big, big, dec, dev, CFG, up)
    o_s, gp_s, ge_s, gg_s = _render(r_small, small, dec, dev, CFG, up)
    for i, (a, b) in enumerate(zip(o_b, o_s)):
        assert torch.equal(a, b), f"output {i} differs between the element beyond 4 GB and the same scene alone"
    assert float(o_s[2].detach().abs().max()) > 0
    _close("grad_mlp_params", gp_b, gp_s)
    _close("grad_encoding", ge_b, ge_s)
    for i, (gb, gs) in enumerate(zip(gg_b, gg_s)):
        assert float(gs.abs().max()) > 0
        _close(f"grad_grid{i}[last element]", gb[batch - 1], gs[0])
        assert float(gb[:batch - 1].abs().max()) == 0.0, f"grad_grid{i}: a write landed in an element the rays do not look at"


def test_renderer_flat_triplane_beyond_4gb():
    """The flat [rows, C] form: ONE tensor of 4.4 GB, the yz plane of the last batch element ends at its end (canonical triplane:
    scatter_triplane of the tuned backward)."""
    dev = _dev()
    C, S = 32, 512
    per = 3 * S * S * C * 4
    batch = (GB4 + per - 1) // per + 1
    gen = torch.Generator().manual_seed(12)
    shapes = [(1, S, S, C), (S, 1, S, C), (S, S, 1, C)]
    pairs = [_big_and_small(gen, s, batch, dev) for s in shapes]
    flat_big, sizes_big = lp.flatten_grid([p[0] for p in pairs])
    flat_small, sizes_small = lp.flatten_grid([p[1] for p in pairs])
    del pairs
    assert flat_big.numel() * 4 >= GB4
    dec = random_decoder(gen, 2, 2, 2, C, 32, 3, std=0.15)
    h, w = 48, 64
    up = tuple(t.to(dev) for t in (torch.randn(h * w, generator=gen), torch.randn(h * w, generator=gen), torch.randn(h * w, 3, generator=gen)))
    r_big, r_small = _rays(h, w, batch - 1, 32, dev), _rays(h, w, 0, 32, dev)
    assert lp.kernel_family(r_big, flat_big, dec, grid_sizes=sizes_big.tolist()) == 1
    o_b, gp_b, ge_b, gg_b = _render(r_big, flat_big, dec, dev, CFG, up, grid_sizes=sizes_big.tolist())
    o_s, gp_s, ge_s, gg_s = _render(r_small, flat_small, dec, dev, CFG, up, grid_sizes=sizes_small.tolist())
    for i, (a, b) in enumerate(zip(o_b, o_s)):
        assert torch.equal(a, b), f"output {i}"
    _close("grad_mlp_params", gp_b, gp_s)
    _close("grad_encoding", ge_b, ge_s)
    # rows of the last batch element of each plane inside the flat tensors
    rows_1 = S * S
    total = 0.0
    for g in range(3):
        lo_b = g * batch * rows_1 + (batch - 1) * rows_1
        _close(f"grad plane {g}", gg_b[lo_b:lo_b + rows_1], gg_s[g * rows_1:(g + 1) * rows_1])
        total += float(gg_b[lo_b:lo_b + rows_1].abs().sum())
        assert float(gg_b[g * batch * rows_1:lo_b].abs().max()) == 0.0, f"plane {g}: a write landed in another batch element"
    assert total > 0


def test_splatter_beyond_4gb():
    """Plain Splatter into an output grid-list of 4.6 GB (the walks address rows through 64-bit bases since round 3; this pins it)."""
    dev = _dev()
    shape = (128, 128, 128, 32)
    batch = _batch_for(shape)
    h, w = 48, 64
    cfg = dict(num_samples=96, num_samples_inf=0, mask_out_of_bounds_samples=True, contract_coords=False)
    r_big, r_small = _rays(h, w, batch - 1, 32, dev), _rays(h, w, 0, 32, dev)
    out_b = lp.lightplane_splatter(r_big, [(batch,) + shape], **cfg)[0]
    out_s = lp.lightplane_splatter(r_small, [(1,) + shape], **cfg)[0]
    assert out_b.numel() * 4 >= GB4
    _close("splatted element", out_b[batch - 1], out_s[0], tol=1e-5)
    assert float(out_s.abs().max()) > 0
    assert float(out_b[:batch - 1].abs().max()) == 0.0
    gen = torch.Generator().manual_seed(13)
    up = torch.randn(1, *shape, generator=gen).to(dev)
    (out_s * up).sum().backward()
    (out_b[batch - 1:] * up).sum().backward()
    _close("grad_encoding", r_big.encoding.grad, r_small.encoding.grad, tol=1e-5)


def test_mlp_splatter_input_grid_beyond_4gb():
    """MLP-Splatter: the INPUT grid-list (gathered in the forward, its gradient scattered by the per-slot walk in the backward) beyond 4 GB"""
    dev = _dev()
    shape_in = (128, 128, 128, 16)
    shape_out = (32, 32, 32, 16)
    batch = _batch_for(shape_in)
    gen = torch.Generator().manual_seed(14)
    big, small = _big_and_small(gen, shape_in, batch, dev)
    mlp = random_splatter_mlp(gen, 2, 16, 16, 16, std=0.3)
    h, w = 48, 64
    cfg = dict(num_samples=64, num_samples_inf=0, mask_out_of_bounds_samples=True, contract_coords=False)
    res = []
    for grid, b, B in ((big, batch - 1, batch), (small, 0, 1)):
        rays = _rays(h, w, b, 16, dev)
        params = mlp.mlp_params.to(dev).clone().requires_grad_(True)
        sp = lp.SplatterParams(params, mlp.n_hidden)
        g = grid.clone().requires_grad_(True)
        assert lp.mlp_splatter_kernel_family([(B,) + shape_out], sp, [(B,) + shape_in]) == 3
        out = lp.lightplane_mlp_splatter(rays, [(B,) + shape_out], sp, [g], **cfg)[0]
        up = torch.randn(1, *shape_out, generator=torch.Generator().manual_seed(15)).to(dev)
        (out[b:b + 1] * up).sum().backward()
        res.append((out[b], params.grad, rays.encoding.grad, g.grad, b))
    (o_b, gp_b, ge_b, gg_b, b), (o_s, gp_s, ge_s, gg_s, _) = res
    _close("splatted element", o_b, o_s, tol=1e-5)
    _close("grad_mlp_params", gp_b, gp_s)
    _close("grad_encoding", ge_b, ge_s)
    assert float(gg_s.abs().max()) > 0
    _close("grad_input_grid[last element]", gg_b[b], gg_s[0])
    assert float(gg_b[:b].abs().max()) == 0.0


def test_renderer_forward_on_a_136gb_grid_list():
    """Rows are 32-bit: the largest grid-list the MFMA kernels take is just below 2^31 rows -- 127 scenes of 256^3 x 16 channels = 136 GB
    in ONE tensor (an MI355X has 288 GB).  Forward only (its gradient would be another 136 GB): rays at the last scene give the outputs
    of that scene alone, bit for bit."""
    dev = _dev()
    shape = (256, 256, 256, 16)
    batch = 127
    need = batch * 256 ** 3 * 16 * 4
    free, _ = torch.cuda.mem_get_info(dev)
    if free < need + (8 << 30):
        pytest.skip(f"needs {need / 2**30:.0f} GiB of free HBM, {free / 2**30:.0f} GiB are free")
    gen = torch.Generator().manual_seed(21)
    big, small = _big_and_small(gen, shape, batch, dev)
    assert big.numel() // 16 < (1 << 31) <= (batch + 1) * 256 ** 3
    dec = random_decoder(gen, 2, 2, 2, 16, 32, 3, std=0.15)
    params = dec.mlp_params.to(dev)
    hdec = lp.DecoderParams(params, dec.n_hidden_trunk, dec.n_hidden_opacity, dec.n_hidden_color, dec.color_chn)
    h, w = 48, 64
    r_big, r_small = _rays(h, w, batch - 1, 32, dev), _rays(h, w, 0, 32, dev)
    assert lp.kernel_family(r_big, [big], dec) == 1
    with torch.no_grad():
        o_b = lp.lightplane_renderer(r_big, [big], hdec, **CFG)
        o_s = lp.lightplane_renderer(r_small, [small], hdec, **CFG)
    for i, (a, b) in enumerate(zip(o_b, o_s)):
        assert torch.equal(a, b), f"output {i}"
    assert float(o_s[2].abs().max()) > 0
    del big
    torch.cuda.empty_cache()
