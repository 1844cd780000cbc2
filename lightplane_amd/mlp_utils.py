"""Alias of the reference's ``lightplane/mlp_utils.py`` import path (re-exports only; the code lives in ``params.py``).

``from lightplane.mlp_utils import DecoderParams`` is what the reference's own speed benchmark does
(``tests/renderer_speed_benchmark.py:30``); with ``import lightplane_amd as lightplane`` / ``sys.modules`` aliasing the same line
resolves here.
"""
from .params import (  # noqa: F401
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

__all__ = [
    "DecoderParams", "SplatterParams", "flatten_decoder_params", "flatten_splatter_params", "flattened_decoder_params_to_list",
    "flattened_triton_decoder_to_list", "get_triton_function_input_dims", "init_decoder_params", "init_splatter_params",
]
