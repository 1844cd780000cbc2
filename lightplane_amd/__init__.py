"""lightplane_amd -- MI355X-native (gfx950) Renderer / Splatter hot path of Lightplane.

Public surface mirrors the reference's ``lightplane/__init__.py:8-31`` for the
hot path: ``Rays``, ``DecoderParams`` / ``SplatterParams`` and their helpers,
the functional ``lightplane_renderer`` / ``lightplane_splatter`` /
``lightplane_mlp_splatter`` and the ``LightplaneRenderer`` / ``LightplaneSplatter``
/ ``LightplaneMLPSplatter`` modules.  The compute path is hand-written HIP behind
a C-ABI shared library (``lightplane_amd/csrc`` -> ``liblightplane_hip.so``); there
is no CPU or PyTorch fallback: a missing library raises at first use.
"""
from .grids import flatten_grid, unflatten_grid
from .params import (
    DecoderParams,
    SplatterParams,
    flatten_decoder_params,
    flatten_splatter_params,
    flattened_decoder_params_to_list,
    flattened_triton_decoder_to_list,
    get_triton_function_input_dims,
    init_decoder_params,
    init_splatter_params,
)
from .rays import Rays, calc_harmonic_embedding, calc_harmonic_embedding_dim, jitter_near_far
from . import config  # noqa: E402
from .renderer import LightplaneFunction, backward_segments, kernel_family, lightplane_renderer  # noqa: E402
from .splatter import (LightplaneMLPSplatterFunction, LightplaneSplatterFunction, lightplane_mlp_splatter,  # noqa: E402
                       lightplane_splatter, mlp_splatter_kernel_family)
from .modules import LightplaneMLPSplatter, LightplaneRenderer, LightplaneSplatter  # noqa: E402
# The reference's sub-module import paths (`from lightplane.mlp_utils import DecoderParams`, tests/renderer_speed_benchmark.py:30)
# exist as alias modules (re-exports only).  Two of them are named like the functions they hold, exactly as in the reference
# (lightplane/__init__.py:8-9): load them first, then bind the FUNCTIONS to the package attributes, so that a later
# `import lightplane_amd.lightplane_renderer` does not turn `lightplane_amd.lightplane_renderer(...)` into a module.
import importlib as _importlib  # noqa: E402
for _m in ("lightplane_renderer", "lightplane_splatter", "misc_utils", "mlp_utils", "ray_utils", "renderer_module", "splatter_module"):
    _importlib.import_module(__name__ + "." + _m)  # (`from . import name` would skip a name the package already has)
del _m
from .renderer import lightplane_renderer  # noqa: E402,F811
from .splatter import lightplane_splatter  # noqa: E402,F811

__all__ = [
    "lightplane_renderer", "lightplane_splatter", "lightplane_mlp_splatter", "LightplaneRenderer",
    "LightplaneSplatter", "LightplaneMLPSplatter", "LightplaneFunction", "LightplaneSplatterFunction",
    "LightplaneMLPSplatterFunction", "config", "kernel_family", "mlp_splatter_kernel_family", "backward_segments",
    "Rays", "DecoderParams", "SplatterParams", "init_decoder_params", "init_splatter_params",
    "flatten_decoder_params", "flatten_splatter_params", "flattened_decoder_params_to_list",
    "flattened_triton_decoder_to_list", "get_triton_function_input_dims", "flatten_grid",
    "unflatten_grid", "calc_harmonic_embedding", "calc_harmonic_embedding_dim", "jitter_near_far",
]

_NOT_PROVIDED = {
    "lightplane_renderer_naive": "the reference's pure-PyTorch path is this repository's test oracle (oracle/), not product code",
    "lightplane_splatter_naive": "the reference's pure-PyTorch path is this repository's test oracle (oracle/), not product code",
    "lightplane_mlp_splatter_naive": "the reference's pure-PyTorch path is this repository's test oracle (oracle/), not product code",
    "visualize_rays_plotly": "plotting helpers are outside the scope of the MI355X hot path (DESIGN.md section 7)",
}


def __getattr__(name):
    if name in _NOT_PROVIDED:
        raise AttributeError(f"lightplane_amd.{name} is deliberately not provided: {_NOT_PROVIDED[name]}")
    raise AttributeError(f"module 'lightplane_amd' has no attribute '{name}'")
