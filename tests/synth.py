"""Seeded synthetic inputs shared by the golden generator, the parity tests and smoke().

Recipes follow the reference's test fixtures (tests/utils.py:230-268 `random_rays`,
:283-324 `random_grid`, :327-376 `random_mlp_decoder_params` -- Xavier init then
overwritten with N(0, 0.01)), generated on the CPU so that the oracle and the
GPU see identical bits.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from lightplane_amd import DecoderParams, Rays, SplatterParams
from lightplane_amd.params import flatten_decoder_params, flatten_splatter_params


def grid_sizes_for(base, is_triplane: bool):
    """[B, D, H, W, C] -> list of sizes (one voxel grid or the three planes)."""
    base = list(base)
    if not is_triplane:
        return [base]
    out = []
    for ax in (1, 2, 3):
        s = list(base)
        s[ax] = 1
        out.append(s)
    return out


def random_rays(gen, n_rays, batch_size, enc_dim, one_ray=False) -> Rays:
    def rn(c):
        if one_ray:
            return torch.randn(1, c, generator=gen).repeat(n_rays, 1)
        return torch.randn(n_rays, c, generator=gen)

    grid_idx = torch.randint(0, batch_size, (n_rays,), generator=gen, dtype=torch.long)
    origins = rn(3) / 3.0
    directions = -origins + rn(3) * 0.1
    near = rn(1)[:, 0] * 0.1 + 0.1
    far = rn(1)[:, 0].abs() * 0.1 + 3.0
    enc = None if enc_dim is None else rn(enc_dim)
    return Rays(directions=directions, origins=origins, grid_idx=grid_idx, near=near, far=far, encoding=enc)


def random_grids(gen, sizes) -> List[torch.Tensor]:
    return [torch.randn(*s, generator=gen) for s in sizes]


def _rand_mlp(gen, n_layers, d_in, d_hidden, d_out, std):
    ws, bs = [], []
    for l in range(n_layers):
        i = d_in if l == 0 else d_hidden
        o = d_out if l == n_layers - 1 else d_hidden
        ws.append(torch.randn(i, o, generator=gen) * std)
        bs.append(torch.randn(o, generator=gen) * std)
    return ws, bs


def random_decoder(gen, n_layers_trunk, n_layers_opacity, n_layers_color, input_chn, hidden_chn,
                   color_chn, use_separate_color_grid=False, std=0.01, pad_color=True) -> DecoderParams:
    """N(0, std) weights AND biases in the reference's flat layout."""
    if use_separate_color_grid:
        n_layers_trunk = 0
    wt, bt = _rand_mlp(gen, n_layers_trunk, input_chn, hidden_chn, hidden_chn, std) if n_layers_trunk else ([], [])
    head_in = input_chn if use_separate_color_grid else hidden_chn
    wo, bo = _rand_mlp(gen, n_layers_opacity, head_in, hidden_chn, 1, std)
    wc, bc = _rand_mlp(gen, n_layers_color, head_in, hidden_chn, color_chn, std)
    flat, nt, no, nc = flatten_decoder_params(wt, bt, wo, bo, wc, bc, pad_color)
    return DecoderParams(flat, nt, no, nc, color_chn)


def random_splatter_mlp(gen, n_layers, input_chn, hidden_chn, out_chn, std=0.01) -> SplatterParams:
    w, b = _rand_mlp(gen, n_layers, input_chn, hidden_chn, out_chn, std)
    return SplatterParams(*flatten_splatter_params(w, b))


def pinhole_rays(height, width, cam_dist=2.7, enc_dim=None, gen=None, grid_idx=0,
                 azimuth_deg=0.0, elevation_deg=0.0) -> Rays:
    """Pinhole camera at distance ``cam_dist`` looking at the origin so that the
    [-1,1]^3 cube fills the frame (SURVEY.md 8(d) cfg 2 recipe).  near/far bracket
    the cube's bounding sphere.  Rays are emitted in row-major pixel order."""
    import math

    half = 1.0 / (cam_dist - 1.0)  # tan(fov/2): the cube's front face fills the frame
    ys = (torch.arange(height, dtype=torch.float32) + 0.5) / height * 2 - 1
    xs = (torch.arange(width, dtype=torch.float32) + 0.5) / width * 2 - 1
    yy, xx = torch.meshgrid(ys, xs, indexing="ij")
    d_cam = torch.stack([xx * half, yy * half, -torch.ones_like(xx)], dim=-1).reshape(-1, 3)
    az, el = math.radians(azimuth_deg), math.radians(elevation_deg)
    # camera-to-world rotation: first elevate about x, then rotate about y
    rx = torch.tensor([[1, 0, 0], [0, math.cos(el), -math.sin(el)], [0, math.sin(el), math.cos(el)]], dtype=torch.float32)
    ry = torch.tensor([[math.cos(az), 0, math.sin(az)], [0, 1, 0], [-math.sin(az), 0, math.cos(az)]], dtype=torch.float32)
    rot = ry @ rx
    dirs = d_cam @ rot.T
    origin = (rot @ torch.tensor([0.0, 0.0, cam_dist])).expand_as(dirs).contiguous()
    n = dirs.shape[0]
    r = math.sqrt(3.0)
    near = torch.full((n,), cam_dist - r)
    far = torch.full((n,), cam_dist + r)
    enc = None
    if enc_dim is not None:
        enc = torch.randn(n, enc_dim, generator=gen)
    return Rays(directions=dirs.contiguous(), origins=origin, grid_idx=torch.full((n,), grid_idx, dtype=torch.long),
                near=near, far=far, encoding=enc)


def cat_rays(parts) -> Rays:
    """Concatenate ray batches (e.g. two camera images looking at two batch entries)."""
    kw = {}
    for f in ("directions", "origins", "grid_idx", "near", "far", "encoding"):
        vals = [getattr(p, f) for p in parts]
        kw[f] = None if vals[0] is None else torch.cat(vals, dim=0)
    return Rays(**kw)


def pinhole_crop(height, width, y0, x0, h, w, **kw) -> Rays:
    """The ``h x w`` block of pixels at ``(y0, x0)`` of a ``height x width`` pinhole image, row-major: rays as
    dense as in the full image (neighbouring pixels), but few enough for the CPU oracle."""
    full = pinhole_rays(height, width, **kw)
    ys = torch.arange(y0, y0 + h)
    xs = torch.arange(x0, x0 + w)
    idx = (ys[:, None] * width + xs[None, :]).reshape(-1)
    return full[idx]


@dataclass
class RendererCase:
    """One seeded Renderer test case (inputs + config)."""

    name: str
    seed: int = 0
    n_rays: int = 24
    grid_base: tuple = (2, 5, 6, 7, 16)
    is_triplane: bool = False
    extra_voxel: bool = False  # triplane + one extra voxel grid in the same grid-list
    n_layers: tuple = (2, 2, 2)  # trunk, opacity, color
    hidden: int = 32
    color_chn: int = 3
    num_samples: int = 9
    num_samples_inf: int = 0
    gain: float = 1.0
    mask_oob: bool = False
    contract: bool = False
    scaffold_size: Optional[tuple] = None  # (D, H, W)
    separate_color_grid: bool = False
    color_grid_base: Optional[tuple] = None  # colour grid-list of another size / type (default: like the grid-list)
    color_is_triplane: Optional[bool] = None
    noise_sigma: float = 0.0
    noise_seed: int = 0
    param_std: float = 0.2
    one_ray: bool = False

    def build(self):
        gen = torch.Generator().manual_seed(self.seed)
        B, C = self.grid_base[0], self.grid_base[-1]
        sizes = grid_sizes_for(self.grid_base, self.is_triplane)
        if self.extra_voxel:
            sizes = sizes + [[B, 4, 3, 5, C]]
        grids = random_grids(gen, sizes)
        color_grids = None
        if self.separate_color_grid:
            c_base = self.color_grid_base if self.color_grid_base is not None else self.grid_base
            c_tri = self.color_is_triplane if self.color_is_triplane is not None else self.is_triplane
            color_grids = random_grids(gen, sizes if self.color_grid_base is None and self.color_is_triplane is None
                                       else grid_sizes_for(c_base, c_tri))
        dec = random_decoder(gen, *self.n_layers, input_chn=C, hidden_chn=self.hidden, color_chn=self.color_chn,
                             use_separate_color_grid=self.separate_color_grid, std=self.param_std)
        enc_dim = int(dec.n_hidden_color[0])
        rays = random_rays(gen, self.n_rays, B, enc_dim, one_ray=self.one_ray)
        scaffold = None
        if self.scaffold_size is not None:
            scaffold = (torch.rand(B, *self.scaffold_size, generator=gen) > 0.4).float()
        cfg = dict(num_samples=self.num_samples, gain=self.gain, num_samples_inf=self.num_samples_inf,
                   mask_out_of_bounds_samples=self.mask_oob, contract_coords=self.contract,
                   inject_noise_sigma=self.noise_sigma, inject_noise_seed=self.noise_seed)
        # upstream gradients for the backward check
        g_len = torch.randn(self.n_rays, generator=gen)
        g_nlt = torch.randn(self.n_rays, generator=gen)
        g_feat = torch.randn(self.n_rays, self.color_chn, generator=gen)
        return dict(rays=rays, grids=grids, color_grids=color_grids, decoder=dec, scaffold=scaffold, cfg=cfg,
                    sizes=sizes, upstream=(g_len, g_nlt, g_feat))


@dataclass
class SplatterCase:
    name: str
    seed: int = 0
    n_rays: int = 24
    out_base: tuple = (2, 6, 5, 7, 32)
    is_triplane: bool = False
    num_samples: int = 9
    num_samples_inf: int = 0
    mask_oob: bool = False
    contract: bool = False
    # MLP splatter
    use_mlp: bool = False
    in_base: tuple = (2, 4, 6, 5, 32)
    in_triplane: bool = False
    n_layers: int = 3
    hidden: int = 32
    feat_dim: int = 32

    def build(self):
        gen = torch.Generator().manual_seed(self.seed)
        B = self.out_base[0]
        out_sizes = grid_sizes_for(self.out_base, self.is_triplane)
        enc_dim = self.feat_dim if self.use_mlp else self.out_base[-1]
        rays = random_rays(gen, self.n_rays, B, enc_dim)
        rays.encoding = torch.rand(self.n_rays, enc_dim, generator=gen)
        mlp, in_grids, in_sizes = None, None, None
        if self.use_mlp:
            in_base = list(self.in_base)
            in_base[-1] = self.feat_dim
            in_sizes = grid_sizes_for(in_base, self.in_triplane)
            in_grids = random_grids(gen, in_sizes)
            mlp = random_splatter_mlp(gen, self.n_layers, self.feat_dim, self.hidden, self.out_base[-1], std=0.2)
        cfg = dict(num_samples=self.num_samples, num_samples_inf=self.num_samples_inf,
                   mask_out_of_bounds_samples=self.mask_oob, contract_coords=self.contract)
        up = [torch.randn(*s, generator=gen) for s in out_sizes]
        return dict(rays=rays, out_sizes=out_sizes, mlp=mlp, in_grids=in_grids, in_sizes=in_sizes, cfg=cfg, upstream=up)


RENDERER_CASES = [
    RendererCase("voxel_basic"),
    RendererCase("triplane_basic", seed=1, is_triplane=True),
    RendererCase("triplane_plus_voxel", seed=2, is_triplane=True, extra_voxel=True, gain=3.0),
    RendererCase("voxel_inf_contract", seed=3, num_samples_inf=4, contract=True),
    RendererCase("triplane_mask", seed=4, is_triplane=True, mask_oob=True, num_samples_inf=3),
    RendererCase("voxel_scaffold", seed=5, scaffold_size=(6, 4, 5), gain=3.0),
    RendererCase("triplane_colorgrid", seed=6, is_triplane=True, separate_color_grid=True, n_layers=(0, 2, 2)),
    RendererCase("voxel_noise", seed=7, noise_sigma=1.0, noise_seed=1234),
    RendererCase("voxel_deep", seed=8, n_layers=(4, 2, 4), n_rays=17),
    RendererCase("triplane_h16_c32", seed=9, is_triplane=True, grid_base=(1, 6, 5, 4, 32), hidden=16, n_layers=(1, 3, 2)),
    RendererCase("single_sample", seed=10, num_samples=1, n_rays=5),
    RendererCase("one_ray_repeated", seed=11, one_ray=True, is_triplane=True, n_rays=33),
    RendererCase("color16", seed=12, color_chn=16, n_rays=8),
    RendererCase("triplane_c32", seed=13, is_triplane=True, grid_base=(2, 6, 5, 4, 32), n_rays=40, num_samples_inf=2),
    RendererCase("voxel_c32_color1", seed=14, grid_base=(1, 4, 5, 6, 32), color_chn=1, n_rays=70, num_samples=37),
    # hidden width 64 (the reference's own example configuration): two-block kernels of the layer-looped family (until 0.2.3: a width-64 fp32-MFMA family)
    RendererCase("triplane_h64_c32", seed=15, is_triplane=True, grid_base=(1, 6, 5, 4, 32), hidden=64, n_rays=40,
                 num_samples_inf=2, param_std=0.15),
    RendererCase("voxel_h64_c16_scaffold", seed=16, grid_base=(2, 6, 5, 7, 16), hidden=64, scaffold_size=(6, 4, 5),
                 gain=3.0, n_rays=70, num_samples=21, param_std=0.15),
    RendererCase("triplane_plus_voxel_h64", seed=17, is_triplane=True, extra_voxel=True, hidden=64, mask_oob=True,
                 param_std=0.15),
    # the shapes of the reference's notebooks: heads without a hidden layer, one trunk layer, hidden width 16
    RendererCase("nb2_like_t2_o1_c1", seed=18, is_triplane=True, n_layers=(2, 1, 1), n_rays=40),
    RendererCase("nb1_like_h16_111", seed=19, hidden=16, n_layers=(1, 1, 1), n_rays=70, num_samples=21, param_std=0.3),
    RendererCase("flex_121_h16_c32_noise", seed=20, is_triplane=True, grid_base=(2, 6, 5, 4, 32), hidden=16,
                 n_layers=(1, 2, 1), noise_sigma=0.5, noise_seed=77, num_samples_inf=3, contract=True, param_std=0.3),
    RendererCase("flex_212_c32_scaffold", seed=21, grid_base=(2, 6, 5, 7, 32), n_layers=(2, 1, 2),
                 scaffold_size=(6, 4, 5), gain=3.0, mask_oob=True, n_rays=33),
    # two-grid decoder (separate colour grid-list, no trunk) on the MFMA family
    RendererCase("colorgrid_c32_h16_voxel", seed=24, grid_base=(2, 5, 6, 7, 32), separate_color_grid=True,
                 color_grid_base=(2, 4, 3, 9, 32), n_layers=(0, 2, 2), hidden=16, scaffold_size=(6, 4, 5), n_rays=50),
    RendererCase("colorgrid_heads1_inf", seed=25, is_triplane=True, separate_color_grid=True,
                 color_grid_base=(2, 3, 4, 5, 16), color_is_triplane=False, n_layers=(0, 1, 1), num_samples_inf=3,
                 contract=True, noise_sigma=0.5, noise_seed=77, n_rays=40),
    RendererCase("colorgrid_c32_mixed", seed=26, grid_base=(1, 6, 5, 4, 32), is_triplane=True,
                 separate_color_grid=True, n_layers=(0, 2, 1), mask_oob=True, n_rays=70, num_samples=21),
    # deep decoders: the layer counts of the reference's own sweep (tests/test_renderer_with_autograd.py:49-51: 2 or 4
    # layers per MLP), hidden widths 16 / 32 / 64, up to 16 colour channels -- the layer-looped MFMA family
    RendererCase("triplane_deep444", seed=27, is_triplane=True, n_layers=(4, 4, 4), n_rays=70, num_samples=21),
    RendererCase("voxel_deep342_h64_c32", seed=28, grid_base=(2, 5, 6, 7, 32), hidden=64, n_layers=(3, 4, 2), n_rays=40,
                 param_std=0.15, scaffold_size=(6, 4, 5), gain=3.0),
    RendererCase("colorgrid_deep044", seed=29, is_triplane=True, separate_color_grid=True, n_layers=(0, 4, 4), n_rays=40,
                 mask_oob=True, num_samples_inf=2),
    RendererCase("color16_deep323_h16", seed=30, hidden=16, color_chn=16, n_layers=(3, 2, 3), n_rays=33, param_std=0.3,
                 noise_sigma=0.5, noise_seed=99, contract=True, num_samples_inf=3),
    RendererCase("triplane_242_c32_color4", seed=31, is_triplane=True, grid_base=(2, 6, 5, 4, 32), n_layers=(2, 4, 2),
                 color_chn=4, n_rays=130, num_samples=33),
    # 64 grid channels (the reference takes any power of two >= 16, lightplane_renderer.py:411-416): the two-block
    # instantiation of the layer-looped family, hidden 32 and the reference example's 1/1/2 x 64 decoder depth
    RendererCase("triplane_c64_h32", seed=32, is_triplane=True, grid_base=(2, 6, 5, 4, 64), n_rays=70, num_samples=21,
                 mask_oob=True, param_std=0.15),
    RendererCase("voxel_c64_h64_112_scaffold", seed=33, grid_base=(2, 5, 6, 7, 64), hidden=64, n_layers=(1, 1, 2), n_rays=40,
                 scaffold_size=(6, 4, 5), gain=3.0, mask_oob=True, param_std=0.1),
    # two-grid decoder with hidden width 64 (use_separate_color_grid + mlp_hidden_chn 64): two-block looped kernels since 0.2.4
    RendererCase("colorgrid_h64_c32_triplane", seed=34, is_triplane=True, grid_base=(2, 6, 5, 4, 32), separate_color_grid=True,
                 n_layers=(0, 2, 2), hidden=64, n_rays=70, num_samples=21, mask_oob=True, param_std=0.15),
    RendererCase("colorgrid_h64_c16_o1c2_inf", seed=35, grid_base=(2, 5, 6, 7, 16), separate_color_grid=True,
                 color_grid_base=(2, 4, 3, 9, 16), n_layers=(0, 1, 2), hidden=64, n_rays=40, num_samples_inf=3, contract=True,
                 scaffold_size=(6, 4, 5), gain=3.0, param_std=0.15),
]

SPLATTER_CASES = [
    SplatterCase("voxel_basic"),
    SplatterCase("triplane_basic", seed=1, is_triplane=True),
    SplatterCase("voxel_inf_contract", seed=2, num_samples_inf=4, contract=True),
    SplatterCase("triplane_mask", seed=3, is_triplane=True, mask_oob=True, num_samples_inf=3),
    SplatterCase("mlp_voxel", seed=4, use_mlp=True),
    SplatterCase("mlp_triplane_in_triplane", seed=5, use_mlp=True, is_triplane=True, in_triplane=True, n_layers=4,
                 hidden=64, feat_dim=64, mask_oob=True),
    SplatterCase("single_ray", seed=6, n_rays=1),
    # two-layer hidden-32 MLPs (LightplaneMLPSplatter's default shape): MFMA MLP-Splatter family
    SplatterCase("mlp2_voxel", seed=7, use_mlp=True, n_layers=2, n_rays=70, num_samples=11),
    SplatterCase("mlp2_triplane_c16", seed=8, use_mlp=True, n_layers=2, feat_dim=16, out_base=(2, 6, 5, 7, 16),
                 is_triplane=True, in_triplane=True, mask_oob=True, num_samples_inf=2, n_rays=40),
    SplatterCase("mlp2_voxel_in16_out32", seed=9, use_mlp=True, n_layers=2, feat_dim=16, in_base=(2, 4, 6, 5, 16),
                 contract=True, num_samples_inf=3, n_rays=33),
    # the MLP shapes of the reference's own sweep (tests/test_splatter_with_autograd.py:49-51: hidden 64, 3-4 layers,
    # feature width 32 / 64) and a 64-channel plain splat (the reference's speed benchmark shape)
    SplatterCase("mlp3_voxel_h64_f32", seed=10, use_mlp=True, n_layers=3, hidden=64, feat_dim=32, n_rays=70, num_samples=11),
    SplatterCase("mlp4_voxel_h64_f64_c16", seed=11, use_mlp=True, n_layers=4, hidden=64, feat_dim=64, out_base=(2, 6, 5, 7, 16),
                 num_samples_inf=2, contract=True, n_rays=40),
    SplatterCase("voxel_c64", seed=12, out_base=(2, 6, 5, 7, 64), n_rays=70, num_samples=11, mask_oob=True),
    SplatterCase("triplane_c64", seed=13, out_base=(1, 6, 5, 7, 64), is_triplane=True, n_rays=40),
]


def baseline_cfg1(seed=0):
    """BASELINE.json configs[0] (SURVEY 8(d) cfg 1) exactly: 1 000 random rays (the reference's ``random_rays`` recipe with
    near 0.1 / far 3.0), one 32^3 x 16-channel voxel grid, 64 samples, 2/2/2-layer x 32-hidden decoder, 3 colour channels.
    Same dictionary as ``RendererCase.build``.  Used by bench.py's ``cpu_baseline`` leg, by the golden generator
    (tests/golden/renderer__baseline_cfg1.npz) and by the GPU parity test of this configuration."""
    gen = torch.Generator().manual_seed(seed)
    sizes = [[1, 32, 32, 32, 16]]
    grids = random_grids(gen, sizes)
    dec = random_decoder(gen, 2, 2, 2, 16, 32, 3, std=0.15)
    rays = random_rays(gen, 1000, 1, 32)
    rays.near = torch.full((1000,), 0.1)
    rays.far = torch.full((1000,), 3.0)
    cfg = dict(num_samples=64, gain=1.0, num_samples_inf=0, mask_out_of_bounds_samples=False, contract_coords=False,
               inject_noise_sigma=0.0, inject_noise_seed=0)
    up = (torch.randn(1000, generator=gen), torch.randn(1000, generator=gen), torch.randn(1000, 3, generator=gen))
    return dict(rays=rays, grids=grids, color_grids=None, decoder=dec, scaffold=None, cfg=cfg, sizes=sizes, upstream=up)


def cfg2_inputs(height=256, width=256, channels=16, grid=64, num_samples=128, rank=0):
    """BASELINE.json configs[1] (cfg 2) as bench.py builds it: pinhole image, canonical triplane grid^2 x channels, 2/2/2 x 32
    decoder with N(0, 0.15) parameters, 32-wide ray encoding, random upstream gradients."""
    gen = torch.Generator().manual_seed(0)
    sizes = grid_sizes_for((1, grid, grid, grid, channels), True)
    grids = random_grids(gen, sizes)
    dec = random_decoder(gen, 2, 2, 2, channels, 32, 3, std=0.15)
    gen_r = torch.Generator().manual_seed(100 + rank)
    rays = pinhole_rays(height, width, enc_dim=32, gen=gen_r)
    n = height * width
    up = (torch.randn(n, generator=gen_r), torch.randn(n, generator=gen_r), torch.randn(n, 3, generator=gen_r))
    cfg = dict(num_samples=num_samples, gain=1.0, num_samples_inf=0, mask_out_of_bounds_samples=False, contract_coords=False,
               inject_noise_sigma=0.0, inject_noise_seed=0)
    return dict(rays=rays, grids=grids, color_grids=None, decoder=dec, scaffold=None, cfg=cfg, sizes=sizes, upstream=up)


# ---- Module level: configurations of the reference's LightplaneRenderer module whose outputs and
# parameter gradients are pinned in tests/golden/module_renderer__*.npz (made by tests/golden/make_golden.py from the
# reference module with use_naive_impl=True)
MODULE_RENDERER_CASES = {
    "triplane_c16_alpha": dict(ctor=dict(num_samples=17, color_chn=3, grid_chn=16, mlp_hidden_chn=32, bg_color=0.3, gain=2.0,
                                         opacity_init_bias=-1.0, ray_embedding_num_harmonics=3),
                               grid_base=(2, 6, 5, 7, 16), triplane=True, n_rays=70, seed=31),
    "voxel_c32_logt_mask": dict(ctor=dict(num_samples=11, color_chn=3, grid_chn=32, mlp_hidden_chn=32, bg_color=(0.1, 0.5, 0.9),
                                          gain=3.0, opacity_init_bias=-0.5, ray_embedding_num_harmonics=2,
                                          return_log_transmittance=True, mask_out_of_bounds_samples=True, num_samples_inf=2),
                                grid_base=(1, 5, 6, 4, 32), triplane=False, n_rays=45, seed=32),
}


def module_renderer_inputs(spec):
    """(grid sizes, grids, rays without encoding, upstream gradients, generator for the module parameters)."""
    gen = torch.Generator().manual_seed(spec["seed"])
    sizes = grid_sizes_for(spec["grid_base"], spec["triplane"])
    grids = [0.7 * g for g in random_grids(gen, sizes)]
    rays = random_rays(gen, spec["n_rays"], spec["grid_base"][0], None)
    n = spec["n_rays"]
    up = (torch.randn(n, generator=gen), torch.randn(n, generator=gen), torch.randn(n, 3, generator=gen))
    pgen = torch.Generator().manual_seed(spec["seed"] + 1000)
    return sizes, grids, rays, up, pgen
