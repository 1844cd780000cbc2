"""``torch.nn.Module`` front-ends: LightplaneRenderer / LightplaneSplatter / LightplaneMLPSplatter.

Constructor / forward arguments, ``state_dict`` keys (``mlp_params``,
``harmonic_ray_embedding_linear.{weight,bias}``, buffer ``bg_color``; non-persistent
``n_hidden_*``) and return values follow the reference's
``lightplane/renderer_module.py`` (`LightplaneRenderer` :38-601) and
``lightplane/splatter_module.py`` (`LightplaneSplatter` :25-161, `LightplaneMLPSplatter`
:164-331), so checkpoints interchange.  ``LightplaneRenderer.forward`` is two HIP launches: the harmonic ray
embedding + its Linear layer (``lp_ray_embedding_forward``) and the march with the background / alpha epilogue
fused in (``LpRendererArgs.bg_color`` / ``alpha``); ``config.fused_module_ops = False`` restores the reference's
PyTorch op chain around the functional renderer (same numbers to fp32 round-off, ~17 launches).

``use_naive_impl=True`` is rejected: the pure-PyTorch implementation is this repository's
test oracle (``oracle/``), not part of the product path.
"""
from __future__ import annotations

import copy
import logging
from dataclasses import asdict
from typing import Optional, Tuple

import torch

from .grids import if_not_none_else
from .params import (DecoderParams, SplatterParams, flattened_decoder_params_to_list, init_decoder_params,
                     init_splatter_params)
from .rays import Rays, calc_harmonic_embedding, calc_harmonic_embedding_dim, jitter_near_far
import ctypes

from . import _lib, config
from .renderer import _render, lightplane_renderer
from .splatter import lightplane_mlp_splatter, lightplane_splatter

logger = logging.getLogger(__name__)

_NAIVE_MSG = (
    "use_naive_impl=True is not available in lightplane_amd: the pure-PyTorch implementation is "
    "kept as the CPU test oracle under oracle/ and is never part of the product path."
)


class _RayEmbeddingFunction(torch.autograd.Function):
    """``Linear(harmonic_embedding(normalize(directions)))`` as one HIP kernel per direction (reference
    renderer_module.py:578-601 is normalize -> mul/add/sin/flatten/cat -> Linear: eight launches, and as many again in
    autograd's backward).  Gradients flow to the Linear layer's weight and bias; ray directions are not
    differentiable, as everywhere else in the Renderer (lightplane_renderer.py:724-756)."""

    @staticmethod
    def forward(ctx, directions, weight, bias, n_harmonics: int):
        dev = directions.device
        stream = _lib.current_stream(dev)
        directions, weight, bias = directions.contiguous(), weight.contiguous(), bias.contiguous()
        _lib.check_tensors(dev, {"rays.directions": directions, "harmonic_ray_embedding_linear.weight": weight,
                                 "harmonic_ray_embedding_linear.bias": bias})
        n, e = directions.shape[0], weight.shape[0]
        assert weight.shape[1] == 3 + 6 * n_harmonics
        out = torch.empty(n, e, device=dev, dtype=torch.float32)
        a = _lib.LpRayEmbedArgs()
        a.n_rays, a.directions, a.n_harmonics, a.out_dim = n, _lib.ptr(directions), int(n_harmonics), e
        a.weight, a.bias, a.out = _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(out)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lp_ray_embedding_forward(ctypes.byref(a), stream), "lp_ray_embedding_forward")
        ctx.save_for_backward(directions, weight)
        ctx.n_harmonics = int(n_harmonics)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        directions, weight = ctx.saved_tensors
        need_w, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        if not (need_w or need_b):
            return None, None, None, None
        dev = directions.device
        stream = _lib.current_stream(dev)
        grad_out = grad_out.contiguous()
        gw = torch.zeros_like(weight) if need_w else None
        gb = torch.zeros(weight.shape[0], device=dev, dtype=torch.float32) if need_b else None
        a = _lib.LpRayEmbedArgs()
        a.n_rays, a.directions, a.n_harmonics, a.out_dim = directions.shape[0], _lib.ptr(directions), ctx.n_harmonics, weight.shape[0]
        a.grad_out, a.grad_weight, a.grad_bias = _lib.ptr(grad_out), _lib.ptr(gw), _lib.ptr(gb)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().lp_ray_embedding_backward(ctypes.byref(a), stream), "lp_ray_embedding_backward")
        return None, gw, gb, None


def _fused_embedding_supported(n_harmonics: int, out_dim: int) -> bool:
    # limits of lp_ray_embedding.hip (LDS staging of the backward)
    # (out_dim: check_ray_embed's LP_MAX_WIDTH; beyond it the PyTorch op chain runs, as the docstring promises)
    return (0 <= n_harmonics <= 10 and 1 <= out_dim <= 128
            and 256 * (6 * n_harmonics + 3 + 1 + out_dim + 1) * 4 <= 150 * 1024)


class LightplaneRenderer(torch.nn.Module):
    """Module wrapper of :func:`lightplane_amd.lightplane_renderer`.

    Owns the decoder parameters (``mlp_params``), the optional harmonic ray-direction
    embedding (``harmonic_ray_embedding_linear``) and the background colour.
    """

    def __init__(
        self,
        num_samples: int,
        color_chn: int,
        grid_chn: int,
        mlp_hidden_chn: int,
        mlp_n_layers_opacity: int = 2,
        mlp_n_layers_trunk: int = 2,
        mlp_n_layers_color: int = 2,
        use_separate_color_grid: bool = False,
        opacity_init_bias: float = -5.0,
        gain: float = 1.0,
        bg_color=0.0,
        enable_direction_dependent_colors: bool = True,
        ray_embedding_num_harmonics: Optional[int] = 3,
        num_samples_inf: int = 0,
        mask_out_of_bounds_samples: bool = False,
        contract_coords: bool = False,
        disparity_at_inf: float = 1e-5,
        inject_noise_sigma: float = 0.0,
        inject_noise_seed: Optional[int] = None,
        rays_jitter_near_far: bool = False,
        return_log_transmittance: bool = False,
        triton_block_size: int = 16,
        triton_num_warps: int = 4,
        use_naive_impl: bool = False,
    ) -> None:
        super().__init__()
        if use_naive_impl:
            raise NotImplementedError(_NAIVE_MSG)
        self.num_samples = num_samples
        self.color_chn = color_chn
        self.opacity_init_bias = opacity_init_bias
        self.gain = gain
        self.num_samples_inf = num_samples_inf
        self.mask_out_of_bounds_samples = mask_out_of_bounds_samples
        self.contract_coords = contract_coords
        self.disparity_at_inf = disparity_at_inf
        self.inject_noise_sigma = inject_noise_sigma
        self.inject_noise_seed = inject_noise_seed
        self.rays_jitter_near_far = rays_jitter_near_far
        self.return_log_transmittance = return_log_transmittance
        self.triton_block_size = triton_block_size
        self.triton_num_warps = triton_num_warps
        self.use_naive_impl = False
        self.enable_direction_dependent_colors = enable_direction_dependent_colors
        self.ray_embedding_num_harmonics = ray_embedding_num_harmonics

        if use_separate_color_grid and mlp_n_layers_trunk > 0:
            logger.warning("Auto-setting mlp_n_layers_trunk=0 because a separate feature grid"
                           " for colors is used (use_separate_color_grid=True).")
            mlp_n_layers_trunk = 0

        dec = init_decoder_params(
            device="cpu", n_layers_opacity=mlp_n_layers_opacity, n_layers_trunk=mlp_n_layers_trunk,
            n_layers_color=mlp_n_layers_color, input_chn=grid_chn, hidden_chn=mlp_hidden_chn,
            color_chn=color_chn, opacity_init_bias=opacity_init_bias,
            pad_color_channels_to_min_block_size=True, use_separate_color_grid=use_separate_color_grid,
        )
        self.rays_encoding_dim = int(dec.n_hidden_color[0])
        self.mlp_params = torch.nn.Parameter(dec.mlp_params)

        if ray_embedding_num_harmonics is not None:
            if not enable_direction_dependent_colors:
                raise ValueError(
                    "LightplaneRenderer's viewpoint dependent colors are disabled,"
                    " (enable_direction_dependent_colors=False), but `ray_embedding_num_harmonics` is set."
                    " Set LightplaneRender.ray_embedding_num_harmonics = None if you intended to disable"
                    " viewpoint dependent colors.")
            self.harmonic_ray_embedding_linear = torch.nn.Linear(
                calc_harmonic_embedding_dim(ray_embedding_num_harmonics), self.rays_encoding_dim)
            torch.nn.init.xavier_uniform_(self.harmonic_ray_embedding_linear.weight)
            self.harmonic_ray_embedding_linear.bias.data.zero_()

        for name in ("n_hidden_trunk", "n_hidden_opacity", "n_hidden_color"):
            self.register_buffer(name, getattr(dec, name), persistent=False)
        self.register_buffer("bg_color", self._process_bg_color(bg_color))

    # -- parameter access ----------------------------------------------------------------
    def get_decoder_params(self) -> DecoderParams:
        return DecoderParams(self.mlp_params, self.n_hidden_trunk, self.n_hidden_opacity,
                             self.n_hidden_color, color_chn=self.color_chn)

    def get_decoder_params_list(self):
        return flattened_decoder_params_to_list(self.mlp_params, self.n_hidden_trunk, self.n_hidden_opacity,
                                                self.n_hidden_color)

    def _process_bg_color(self, bg_color) -> torch.Tensor:
        if bg_color is None:
            return self.bg_color
        if isinstance(bg_color, (float, int)):
            bg_color = torch.full((self.color_chn,), float(bg_color))
        elif not torch.is_tensor(bg_color):
            bg_color = torch.tensor(bg_color, dtype=torch.float)
        assert len(bg_color) == self.color_chn
        return bg_color

    # -- point evaluation (reference renderer_module.py:183-417) ---------------------------
    # The reference evaluates these with its naive PyTorch decoder; here they go through the HIP
    # Renderer itself: a "ray" with near = far = 0 and ONE sample sits exactly at its origin, the
    # interval length of a single-sample march is 1 (naive_renderer.py:252-256), so the returned
    # -log T is the opacity at the point and, with a huge gain (T -> 0), the rendered feature is
    # the colour at the point.
    def _render_points(self, pts, pts_to_grid_idx, feature_grid, color_feature_grid, scaffold, gain,
                       mask_out_of_bounds_samples, contract_coords, grid_sizes, rays_encoding):
        n_rays, n_pts, pts_dim = pts.shape
        assert pts_dim == 3
        assert pts_to_grid_idx.shape == (n_rays,)
        n = n_rays * n_pts
        dev = pts.device
        origins = pts.reshape(n, 3).to(torch.float32).contiguous()
        zeros = torch.zeros(n, device=dev, dtype=torch.float32)
        if rays_encoding is None:
            enc = torch.zeros(n, self.rays_encoding_dim, device=dev, dtype=torch.float32)
        else:
            enc = rays_encoding[:, None, :].expand(-1, n_pts, -1).reshape(n, -1).contiguous()
        rays = Rays(directions=torch.zeros_like(origins), origins=origins,
                    grid_idx=pts_to_grid_idx.to(torch.long)[:, None].expand(-1, n_pts).reshape(n).contiguous(),
                    near=zeros, far=zeros, encoding=enc)
        if color_feature_grid is None and int(self.n_hidden_trunk.numel()) == 0:
            color_feature_grid = feature_grid  # opacity-only evaluation in the two-grid mode: colours unused
        return lightplane_renderer(
            rays, feature_grid, self.get_decoder_params(), num_samples=1, gain=gain,
            mask_out_of_bounds_samples=mask_out_of_bounds_samples, contract_coords=contract_coords,
            scaffold=scaffold, color_grid=color_feature_grid, grid_sizes=grid_sizes,
            color_grid_sizes=grid_sizes if color_feature_grid is not None else None)

    def eval_opacity_at_points(self, pts, pts_to_grid_idx, feature_grid, scaffold=None, gain=None,
                               mask_out_of_bounds_samples=None, grid_sizes=None):
        """Opacities ``[n_rays, n_pts]`` of the decoder at ``pts [n_rays, n_pts, 3]`` (reference :302-347)."""
        out = self._render_points(
            pts, pts_to_grid_idx, feature_grid, None, scaffold, if_not_none_else(gain, self.gain),
            if_not_none_else(mask_out_of_bounds_samples, self.mask_out_of_bounds_samples), False, grid_sizes, None)
        return out[1].reshape(pts.shape[0], pts.shape[1])

    def eval_decoder_at_points(self, pts, pts_to_grid_idx, rays_encoding, feature_grid, color_feature_grid=None,
                               scaffold=None, gain=None, mask_out_of_bounds_samples=None, contract_coords=None,
                               directions=None):
        """(opacity ``[n_rays, n_pts]``, colour ``[n_rays, n_pts, color_chn]``) at ``pts`` (reference
        :183-241).

        Implementation: two single-sample renders through the HIP Renderer (a ray with ``near = far = 0`` and one
        sample sits at its origin; the interval length of a single-sample march is 1, so ``-log T`` is the opacity;
        with gain 1e30 the compositing weight ``1 - exp(-x)`` is 1, so the rendered feature is the colour).
        Documented deviations from the reference's naive decoder:
        * colour is 0 (not ``sigmoid(raw)``) at a point whose ``softplus(raw opacity)`` underflows to exactly 0 in
          fp32 (raw < -103: the point is empty to 1e-45), and at points the scaffold / out-of-bounds mask removes;
        * the colour has ``color_chn`` channels, not the padded width (>= 16) of the reference's colour head;
        * autograd through the returned opacity does not see the colour branch (two separate renders), and grids
          have to be passed as a list (flat tensors + sizes are not accepted here);
        * cost: two marches (2 x 12 gathers + 2 decoder evaluations per point) -- meant for scaffolds and
          diagnostics, not for inner loops."""
        n_rays, n_pts, _ = pts.shape
        if rays_encoding is not None:
            assert tuple(rays_encoding.shape) == (n_rays, self.rays_encoding_dim)
        else:
            assert directions is not None, "Must pass one of (rays_encoding, directions)"
            assert directions.shape == (n_rays, 3)
        enc = self._get_ray_encoding(rays_encoding, directions)
        mask = if_not_none_else(mask_out_of_bounds_samples, self.mask_out_of_bounds_samples)
        contract = if_not_none_else(contract_coords, self.contract_coords)
        args = (pts, pts_to_grid_idx, feature_grid, color_feature_grid, scaffold)
        opacity = self._render_points(*args, if_not_none_else(gain, self.gain), mask, contract, None, enc)[1]
        color = self._render_points(*args, 1e30, mask, contract, None, enc)[2]
        return opacity.reshape(n_rays, n_pts), color.reshape(n_rays, n_pts, self.color_chn)

    @torch.no_grad()
    def calculate_scaffold(self, feature_grid, scaffold_size, device, threshold: float = 1e-7, grid_sizes=None,
                           dilate_scaffold: int = 2):
        """Occupancy scaffold ``[B, D, H, W]`` (0/1 floats): the decoder's opacity on the regular lattice
        ``x = linspace(-1, 1, W)``, ``y = linspace(-1, 1, H)``, ``z = linspace(-1, 1, D)``, dilated by a
        max-pool of ``2 * dilate_scaffold + 1`` and thresholded (reference :349-417, which walks the lattice
        slice by slice through the naive decoder; here one HIP launch per batch element)."""
        B, D, H, W = (int(v) for v in scaffold_size)
        lin = lambda n: torch.linspace(0, 1, n, device=device) * 2.0 - 1.0  # noqa: E731
        zz, yy, xx = torch.meshgrid(lin(D), lin(H), lin(W), indexing="ij")
        pts = torch.stack([xx, yy, zz], dim=-1).reshape(1, D * H * W, 3)
        scaffold = torch.empty(B, D, H, W, device=device)
        for b in range(B):
            idx = torch.full((1,), b, dtype=torch.long, device=device)
            scaffold[b] = self.eval_opacity_at_points(pts, idx, feature_grid, grid_sizes=grid_sizes).reshape(D, H, W)
        if dilate_scaffold > 0:
            ks = dilate_scaffold * 2 + 1
            scaffold = torch.nn.functional.max_pool3d(scaffold, kernel_size=ks, padding=dilate_scaffold, stride=1)
        return (scaffold > threshold) * 1.0

    # -- ray encoding ---------------------------------------------------------------------
    def _get_ray_embedding(self, ray_directions: torch.Tensor) -> torch.Tensor:
        if not self.enable_direction_dependent_colors:
            return ray_directions.new_zeros(ray_directions.shape[0], self.rays_encoding_dim)
        assert self.ray_embedding_num_harmonics is not None
        lin = self.harmonic_ray_embedding_linear
        if (config.fused_module_ops and ray_directions.is_cuda and ray_directions.dtype == torch.float32
                and not ray_directions.requires_grad
                and _fused_embedding_supported(self.ray_embedding_num_harmonics, lin.weight.shape[0])):
            return _RayEmbeddingFunction.apply(ray_directions, lin.weight, lin.bias, self.ray_embedding_num_harmonics)
        emb = calc_harmonic_embedding(torch.nn.functional.normalize(ray_directions, dim=-1),
                                      self.ray_embedding_num_harmonics)
        return lin(emb)

    def _get_ray_encoding(self, ray_encoding, directions) -> torch.Tensor:
        if ray_encoding is not None:
            assert not self.enable_direction_dependent_colors or self.ray_embedding_num_harmonics is None
            return ray_encoding
        return self._get_ray_embedding(directions)

    def forward(
        self,
        rays: Rays,
        feature_grid,
        color_feature_grid=None,
        scaffold: Optional[torch.Tensor] = None,
        grid_sizes=None,
        color_grid_sizes=None,
        bg_color=None,
        num_samples: Optional[int] = None,
        gain: Optional[float] = None,
        num_samples_inf: Optional[int] = None,
        mask_out_of_bounds_samples: Optional[bool] = None,
        contract_coords: Optional[bool] = None,
        disparity_at_inf: Optional[float] = None,
        inject_noise_sigma: Optional[float] = None,
        inject_noise_seed: Optional[int] = None,
        rays_jitter_near_far: Optional[bool] = None,
        return_log_transmittance: Optional[bool] = None,
        regenerate_code: Optional[bool] = None,
    ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """Render; every keyword overrides the module default for this call.

        Returns ``(ray_length_render, alpha, feature_render)`` with ``alpha = 1 - T`` (or
        ``log T`` if ``return_log_transmittance``) and ``feature_render`` composited over
        ``bg_color`` (reference renderer_module.py:552-561).
        """
        device = rays.device
        num_samples = if_not_none_else(num_samples, self.num_samples)
        rays_jitter = if_not_none_else(rays_jitter_near_far, self.rays_jitter_near_far)
        return_log_t = if_not_none_else(return_log_transmittance, self.return_log_transmittance)
        bg = self._process_bg_color(if_not_none_else(bg_color, self.bg_color)).to(device)

        _check_renderer_ray_encoding_input(rays.encoding, self.ray_embedding_num_harmonics,
                                           self.rays_encoding_dim, self.enable_direction_dependent_colors)
        r = copy.copy(rays)
        r.encoding = self._get_ray_encoding(rays.encoding, rays.directions)
        if rays_jitter:
            r.near, r.far = jitter_near_far(r.near, r.far, num_samples)

        kw = dict(
            num_samples=num_samples,
            gain=if_not_none_else(gain, self.gain),
            num_samples_inf=if_not_none_else(num_samples_inf, self.num_samples_inf),
            mask_out_of_bounds_samples=if_not_none_else(mask_out_of_bounds_samples, self.mask_out_of_bounds_samples),
            contract_coords=if_not_none_else(contract_coords, self.contract_coords),
            disparity_at_inf=if_not_none_else(disparity_at_inf, self.disparity_at_inf),
            inject_noise_sigma=if_not_none_else(inject_noise_sigma, self.inject_noise_sigma),
            inject_noise_seed=if_not_none_else(inject_noise_seed, self.inject_noise_seed),
            scaffold=scaffold, color_grid=color_feature_grid, grid_sizes=grid_sizes,
            color_grid_sizes=color_grid_sizes,
        )
        if config.fused_module_ops and not bg.requires_grad:
            # background compositing and alpha inside the render kernel (and its backward): reference :552-561
            # (a background colour that needs a gradient takes the op chain below: the kernels do not produce one)
            ray_length, _, feature, alpha = _render(r, feature_grid, self.get_decoder_params(),
                                                    bg_color=bg.to(torch.float32), alpha_mode=2 if return_log_t else 1, **kw)
            return ray_length, alpha, feature
        ray_length, nlt, feature = lightplane_renderer(r, feature_grid, self.get_decoder_params(), **kw)
        transmittance = torch.exp(-nlt)
        feature = feature + transmittance[..., None] * bg
        alpha = -nlt if return_log_t else 1 - transmittance
        return ray_length, alpha, feature


def _check_renderer_ray_encoding_input(ray_encoding, ray_embedding_num_harmonics, ray_encoding_dim,
                                       enable_direction_dependent_colors) -> None:
    """Same decision table (and ``ValueError``s) as reference renderer_module.py:604-667."""
    if ray_encoding is not None and ray_encoding.shape[1] != ray_encoding_dim:
        raise ValueError(f"Ray encoding has a wrong dimension. Expected: {ray_encoding_dim},"
                         f" got: {ray_encoding.shape[1]}")
    if not enable_direction_dependent_colors:
        if ray_encoding is not None:
            raise ValueError("LightplaneRenderer's viewpoint dependent colors are disabled"
                             " (enable_direction_dependent_colors=False), but the `encoding` field of"
                             " `rays` is set. Set rays.encoding=None to disable viewpoint dependent colors.")
        if ray_embedding_num_harmonics is not None:
            raise ValueError("LightplaneRenderer's viewpoint dependent colors are disabled"
                             " (enable_direction_dependent_colors=False), but `ray_embedding_num_harmonics`"
                             " is set. Set it to None to disable viewpoint dependent colors.")
        return
    have_h, have_e = ray_embedding_num_harmonics is not None, ray_encoding is not None
    if have_h != have_e:
        return  # exactly one source of ray encodings: fine
    if not have_e:
        msg = ("rays.encoding is unset (=None), but the Lightplane module is not configured to compute"
               " harmonic ray embeddings (self.ray_embedding_num_harmonics is unset = None).")
    else:
        msg = ("rays.encoding is set, but the Lightplane module is configured to also compute harmonic ray"
               " embeddings (self.ray_embedding_num_harmonics is set).")
    raise ValueError(msg + " Either set ray_embedding_num_harmonics to an integer and rays.encoding=None,"
                           " or pass your own [n_rays, ray_encoding_dim] rays.encoding and set"
                           " ray_embedding_num_harmonics=None.")


class LightplaneSplatter(torch.nn.Module):
    """Module wrapper of :func:`lightplane_amd.lightplane_splatter` (reference splatter_module.py:25-161)."""

    def __init__(
        self,
        num_samples: int,
        grid_chn: int,
        num_samples_inf: int = 0,
        mask_out_of_bounds_samples: bool = False,
        contract_coords: bool = False,
        disparity_at_inf: float = 1e-5,
        rays_jitter_near_far: bool = False,
        triton_block_size: int = 16,
        triton_num_warps: int = 4,
        use_naive_impl: bool = False,
    ):
        super().__init__()
        if use_naive_impl:
            raise NotImplementedError(_NAIVE_MSG)
        self.num_samples = num_samples
        self.num_samples_inf = num_samples_inf
        self.mask_out_of_bounds_samples = mask_out_of_bounds_samples
        self.contract_coords = contract_coords
        self.disparity_at_inf = disparity_at_inf
        self.rays_jitter_near_far = rays_jitter_near_far
        self.triton_block_size = triton_block_size
        self.triton_num_warps = triton_num_warps
        self.use_naive_impl = False
        self.rays_encoding_dim = grid_chn

    def get_splatter_params(self) -> Optional[SplatterParams]:
        return None

    def forward(
        self,
        rays: Rays,
        grid_size,
        num_samples: Optional[int] = None,
        num_samples_inf: Optional[int] = None,
        mask_out_of_bounds_samples: Optional[bool] = None,
        contract_coords: Optional[bool] = None,
        disparity_at_inf: Optional[float] = None,
        rays_jitter_near_far: Optional[bool] = None,
        return_list: bool = True,
        regenerate_code: bool = False,
    ):
        num_samples = if_not_none_else(num_samples, self.num_samples)
        _check_splatter_ray_encoding_input(rays.encoding, self.rays_encoding_dim)
        r = copy.copy(rays)
        if if_not_none_else(rays_jitter_near_far, self.rays_jitter_near_far):
            r.near, r.far = jitter_near_far(r.near, r.far, num_samples)
        return lightplane_splatter(
            r, grid_size, num_samples=num_samples,
            num_samples_inf=if_not_none_else(num_samples_inf, self.num_samples_inf),
            mask_out_of_bounds_samples=if_not_none_else(mask_out_of_bounds_samples, self.mask_out_of_bounds_samples),
            contract_coords=if_not_none_else(contract_coords, self.contract_coords),
            disparity_at_inf=if_not_none_else(disparity_at_inf, self.disparity_at_inf),
            return_list=return_list,
        )


class LightplaneMLPSplatter(torch.nn.Module):
    """Module wrapper of :func:`lightplane_amd.lightplane_mlp_splatter` (reference
    splatter_module.py:164-331): owns the MLP (``mlp_params`` parameter, ``n_hidden`` buffer)."""

    def __init__(
        self,
        num_samples: int,
        grid_chn: int,
        input_grid_chn: int = 32,
        mlp_hidden_chn: int = 32,
        mlp_n_layers: int = 2,
        num_samples_inf: int = 0,
        mask_out_of_bounds_samples: bool = False,
        contract_coords: bool = False,
        disparity_at_inf: float = 1e-5,
        rays_jitter_near_far: bool = False,
        triton_block_size: int = 16,
        triton_num_warps: int = 4,
        use_naive_impl: bool = False,
    ):
        super().__init__()
        if use_naive_impl:
            raise NotImplementedError(_NAIVE_MSG)
        self.num_samples = num_samples
        self.num_samples_inf = num_samples_inf
        self.mask_out_of_bounds_samples = mask_out_of_bounds_samples
        self.contract_coords = contract_coords
        self.disparity_at_inf = disparity_at_inf
        self.rays_jitter_near_far = rays_jitter_near_far
        self.triton_block_size = triton_block_size
        self.triton_num_warps = triton_num_warps
        self.use_naive_impl = False
        assert input_grid_chn is not None, "input_grid_chn must be provided"
        sp = init_splatter_params(device="cpu", n_layers=mlp_n_layers, input_chn=input_grid_chn,
                                  hidden_chn=mlp_hidden_chn, out_chn=grid_chn)
        self.mlp_params = torch.nn.Parameter(sp.mlp_params)
        self.register_buffer("n_hidden", sp.n_hidden, persistent=False)
        self.rays_encoding_dim = input_grid_chn

    def get_splatter_params(self) -> SplatterParams:
        return SplatterParams(self.mlp_params, self.n_hidden)

    def forward(
        self,
        rays: Rays,
        grid_size,
        input_grid,
        num_samples: Optional[int] = None,
        num_samples_inf: Optional[int] = None,
        mask_out_of_bounds_samples: Optional[bool] = None,
        contract_coords: Optional[bool] = None,
        disparity_at_inf: Optional[float] = None,
        input_grid_sizes=None,
        rays_jitter_near_far: Optional[bool] = None,
        return_list: bool = True,
        regenerate_code: bool = False,
    ):
        num_samples = if_not_none_else(num_samples, self.num_samples)
        _check_splatter_ray_encoding_input(rays.encoding, self.rays_encoding_dim)
        assert input_grid is not None, "input_grid must be provided"
        r = copy.copy(rays)
        if if_not_none_else(rays_jitter_near_far, self.rays_jitter_near_far):
            r.near, r.far = jitter_near_far(r.near, r.far, num_samples)
        return lightplane_mlp_splatter(
            r, grid_size, self.get_splatter_params(), input_grid, num_samples=num_samples,
            num_samples_inf=if_not_none_else(num_samples_inf, self.num_samples_inf),
            mask_out_of_bounds_samples=if_not_none_else(mask_out_of_bounds_samples, self.mask_out_of_bounds_samples),
            contract_coords=if_not_none_else(contract_coords, self.contract_coords),
            disparity_at_inf=if_not_none_else(disparity_at_inf, self.disparity_at_inf),
            input_grid_sizes=input_grid_sizes, return_list=return_list,
        )


def _check_splatter_ray_encoding_input(ray_encoding, ray_encoding_dim) -> None:
    if ray_encoding is None:
        raise ValueError("rays.encoding has to be set: it is the feature the splatter pushes into the grid.")
    if ray_encoding.shape[1] != ray_encoding_dim:
        raise ValueError(f"Ray encoding has a wrong dimension. Expected: {ray_encoding_dim},"
                         f" got: {ray_encoding.shape[1]}")
