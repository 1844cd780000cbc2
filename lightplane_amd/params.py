"""MLP parameter containers and the flat parameter wire format.

Mirrors the reference's ``lightplane/mlp_utils.py`` public names
(`DecoderParams` :20-128, `SplatterParams` :131-185, `init_decoder_params`
:188-295, `init_splatter_params` :298-339, `get_triton_function_input_dims`
:342-382, `flatten_decoder_params` :390-456, `flatten_splatter_params` :459-486,
`flattened_decoder_params_to_list` :489-560, `flattened_triton_decoder_to_list`
:563-605).

Wire format read by the HIP kernels (identical to the reference, bit for bit):
``mlp_params`` is a 1-D f32 tensor ::

    [trunk W_0 .. W_{n-1} (each [in, out] row-major), trunk b_0 .. b_{n-1}]
    ++ [opacity W.., b..] ++ [color W.., b..]

with ``y = x @ W + b``.  ``n_hidden_*`` are int32 ``[n_layers + 1]`` vectors
(input width followed by every layer's output width).  The colour head's last
layer may be zero-padded to ``MIN_BLOCK_SIZE`` (=16) outputs; ``color_chn`` is
the real channel count.  The kernels only evaluate the real channels.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

import torch

#: Triton tile minimum of the reference (triton_src/shared/const.py:15); only
#: kept because it defines the colour-head padding of the wire format.
MIN_BLOCK_SIZE = 16


@dataclass
class DecoderParams:
    """Parameters of the Renderer's decoder (trunk / opacity / colour MLPs)."""

    mlp_params: torch.Tensor
    n_hidden_trunk: torch.Tensor
    n_hidden_opacity: torch.Tensor
    n_hidden_color: torch.Tensor
    color_chn: int


@dataclass
class SplatterParams:
    """Parameters of the MLP-Splatter's MLP."""

    mlp_params: torch.Tensor
    n_hidden: torch.Tensor


# --------------------------------------------------------------------------------------
# helpers on one MLP
# --------------------------------------------------------------------------------------


def int_list_of(t):
    """Python ints of a small integer tensor (layer widths, grid sizes).  Such tensors usually live on the GPU
    (module buffers), where reading them is a device sync per call: the list is remembered on the tensor object
    itself, per in-place version.  (The reference pays `.item()` / `.tolist()` syncs here on every call.)"""
    if not torch.is_tensor(t):
        return [int(v) for v in t]
    cached = getattr(t, "_lp_ints", None)
    if cached is None or cached[0] != t._version:
        v = t.tolist()
        cached = (t._version, v)
        try:
            t._lp_ints = cached
        except AttributeError:  # pragma: no cover
            pass
    return [list(r) for r in cached[1]] if cached[1] and isinstance(cached[1][0], list) else list(cached[1])


def _layer_dims(n_hidden: torch.Tensor | Sequence[int]) -> List[Tuple[int, int]]:
    dims = [int(v) for v in int_list_of(n_hidden)]
    return [(dims[i], dims[i + 1]) for i in range(len(dims) - 1)]


def mlp_numel(n_hidden: torch.Tensor | Sequence[int]) -> int:
    """Number of floats one MLP occupies in the flat layout."""
    return sum(i * o + o for i, o in _layer_dims(n_hidden))


def _n_hidden_of(weights: Sequence[torch.Tensor], device=None) -> torch.Tensor:
    if len(weights) == 0:
        return torch.zeros(0, dtype=torch.int32, device=device)
    dims = [int(weights[0].shape[0])] + [int(w.shape[1]) for w in weights]
    return torch.tensor(dims, dtype=torch.int32, device=device)


def _split_one_mlp(flat: torch.Tensor, n_hidden, transpose: bool = False):
    """Inverse of the per-MLP flattening: all weights first, then all biases."""
    dims = _layer_dims(n_hidden)
    n_w = sum(i * o for i, o in dims)
    n_b = sum(o for _, o in dims)
    assert flat.numel() == n_w + n_b, (
        f"MLP parameter vector has {flat.numel()} elements, expected {n_w + n_b}"
    )
    weights, biases = [], []
    pos = 0
    for i, o in dims:
        w = flat[pos : pos + i * o].reshape(i, o)
        weights.append(w.t().contiguous() if transpose else w)
        pos += i * o
    for _, o in dims:
        biases.append(flat[pos : pos + o])
        pos += o
    return weights, biases


# kept under the reference's private name because naive_splatter-style callers use it
_flattened_one_mlp_params_to_list = _split_one_mlp


def _check_mlp_chain(weights: Sequence[torch.Tensor], biases: Sequence[torch.Tensor]) -> None:
    assert len(weights) == len(biases)
    prev_out = None
    for w, b in zip(weights, biases):
        assert w.ndim == 2 and b.ndim == 1
        assert w.device == b.device
        assert w.shape[1] == b.shape[0]
        if prev_out is not None:
            assert w.shape[0] == prev_out, "consecutive layers do not chain"
        prev_out = w.shape[1]


def _xavier_mlp(
    n_layers: int, in_chn: int, hidden_chn: int, out_chn: int, device, last_bias: float = 0.0
):
    """Xavier-uniform (ReLU gain) weights, zero biases, optional last-layer bias."""
    gain = torch.nn.init.calculate_gain("relu")
    weights, biases = [], []
    for layer in range(n_layers):
        fan_in = in_chn if layer == 0 else hidden_chn
        fan_out = out_chn if layer == n_layers - 1 else hidden_chn
        w = torch.empty(fan_in, fan_out, device=device)
        torch.nn.init.xavier_uniform_(w, gain=gain)
        weights.append(w)
        if layer == n_layers - 1:
            biases.append(torch.full((fan_out,), float(last_bias), device=device))
        else:
            biases.append(torch.zeros(fan_out, device=device))
    return weights, biases


# --------------------------------------------------------------------------------------
# decoder (Renderer)
# --------------------------------------------------------------------------------------


def flatten_decoder_params(
    weights_trunk,
    biases_trunk,
    weights_opacity,
    biases_opacity,
    weights_color,
    biases_color,
    pad_color_channels_to_min_block_size: bool = True,
):
    """Pack the three MLPs into the flat wire format.

    Returns ``(mlp_params, n_hidden_trunk, n_hidden_opacity, n_hidden_color)``.
    With ``pad_color_channels_to_min_block_size`` the colour head's last layer
    is zero-padded to at least 16 output channels (reference mlp_utils.py:414-424).
    """
    weights_color, biases_color = list(weights_color), list(biases_color)
    if pad_color_channels_to_min_block_size and len(biases_color) > 0:
        n_pad = max(MIN_BLOCK_SIZE - biases_color[-1].numel(), 0)
        if n_pad > 0:
            weights_color[-1] = torch.nn.functional.pad(weights_color[-1], [0, n_pad])
            biases_color[-1] = torch.nn.functional.pad(biases_color[-1], [0, n_pad])

    groups = [weights_trunk, biases_trunk, weights_opacity, biases_opacity, weights_color, biases_color]
    for w, b in ((weights_trunk, biases_trunk), (weights_opacity, biases_opacity), (weights_color, biases_color)):
        _check_mlp_chain(w, b)
    mlp_params = torch.cat([t.reshape(-1) for g in groups for t in g], dim=0).contiguous()
    dev = mlp_params.device
    n_hidden = tuple(_n_hidden_of(w, dev) for w in (weights_trunk, weights_opacity, weights_color))
    assert mlp_params.dtype == torch.float32
    assert mlp_params.numel() == sum(mlp_numel(nh) for nh in n_hidden)
    return (mlp_params, *n_hidden)


def flattened_decoder_params_to_list(
    mlp_params: torch.Tensor,
    n_hidden_trunk: torch.Tensor,
    n_hidden_opacity: torch.Tensor,
    n_hidden_color: torch.Tensor,
    transpose: bool = False,
):
    """Unpack the flat vector into
    ``(weights_trunk, biases_trunk, weights_opacity, biases_opacity, weights_color, biases_color)``."""
    sizes = [mlp_numel(nh) for nh in (n_hidden_trunk, n_hidden_opacity, n_hidden_color)]
    assert mlp_params.numel() == sum(sizes), (
        f"The number of elements in mlp param should be {sum(sizes)}."
        f" Got {mlp_params.numel()} instead."
    )
    out = []
    pos = 0
    for size, nh in zip(sizes, (n_hidden_trunk, n_hidden_opacity, n_hidden_color)):
        w, b = _split_one_mlp(mlp_params[pos : pos + size], nh, transpose)
        out += [w, b]
        pos += size
    return tuple(out)


def flattened_triton_decoder_to_list(
    mlp_params: torch.Tensor,
    n_layers_trunk: int,
    n_layers_opacity: int,
    n_layers_color: int,
    input_chn: int,
    hidden_chn: int,
    color_chn: int,
):
    """Unpack given layer counts and widths instead of ``n_hidden_*`` vectors."""

    def widths(d_in, d_out, n_layers):
        if n_layers == 0:
            return torch.zeros(0, dtype=torch.int32)
        return torch.tensor([d_in] + [hidden_chn] * (n_layers - 1) + [d_out], dtype=torch.int32)

    return flattened_decoder_params_to_list(
        mlp_params,
        widths(input_chn, hidden_chn, n_layers_trunk),
        widths(hidden_chn, 1, n_layers_opacity),
        widths(hidden_chn, color_chn, n_layers_color),
    )


def init_decoder_params(
    device,
    n_layers_opacity: int,
    n_layers_trunk: int,
    n_layers_color: int,
    input_chn: int = 32,
    hidden_chn: int = 32,
    color_chn: int = 3,
    opacity_init_bias: float = 0.0,
    pad_color_channels_to_min_block_size: bool = True,
    use_separate_color_grid: bool = False,
) -> DecoderParams:
    """Xavier-initialise the decoder (argument order as reference mlp_utils.py:188-199).

    Without a separate colour grid: trunk ``input_chn -> hidden``, heads
    ``hidden -> 1`` / ``hidden -> color_chn``.  With a separate colour grid
    there is no trunk and both heads take ``input_chn`` inputs.
    """
    if n_layers_trunk > 0:
        assert not use_separate_color_grid, (
            "Cannot use trunk MLP with a separate color grid. Please set n_layers_trunk==0."
        )
        w_t, b_t = _xavier_mlp(n_layers_trunk, input_chn, hidden_chn, hidden_chn, device)
    else:
        w_t, b_t = [], []
    head_in = input_chn if use_separate_color_grid else hidden_chn
    w_o, b_o = _xavier_mlp(n_layers_opacity, head_in, hidden_chn, 1, device, last_bias=opacity_init_bias)
    w_c, b_c = _xavier_mlp(n_layers_color, head_in, hidden_chn, color_chn, device)
    mlp_params, nh_t, nh_o, nh_c = flatten_decoder_params(
        w_t, b_t, w_o, b_o, w_c, b_c, pad_color_channels_to_min_block_size
    )
    return DecoderParams(mlp_params, nh_t, nh_o, nh_c, color_chn)


def get_triton_function_input_dims(
    n_hidden_trunk: torch.Tensor, n_hidden_opacity: torch.Tensor, n_hidden_color: torch.Tensor
):
    """``(dim_hidden_trunk, dim_hidden_opacity, dim_hidden_color, n_layers_trunk,
    n_layers_opacity, n_layers_color, num_render_channels)`` -- the name is the
    reference's (mlp_utils.py:342-382); the HIP path uses the same summary.

    All hidden layers of one MLP must share a width (same restriction as the
    reference); ``AssertionError`` otherwise.
    """
    nh_t = [int(v) for v in n_hidden_trunk.tolist()]
    nh_o = [int(v) for v in n_hidden_opacity.tolist()]
    nh_c = [int(v) for v in n_hidden_color.tolist()]
    if len(nh_t) == 0:
        n_layers_trunk, dim_trunk = 0, 0
    else:
        n_layers_trunk, dim_trunk = len(nh_t) - 1, nh_t[1]
        assert all(v == dim_trunk for v in nh_t[1:]), "trunk layers must share one width"
    dim_opacity, dim_color = nh_o[1], nh_c[1]
    assert all(v == dim_opacity for v in nh_o[1:-1]), "opacity hidden layers must share one width"
    assert all(v == dim_color for v in nh_c[1:-1]), "color hidden layers must share one width"
    return (
        dim_trunk,
        dim_opacity,
        dim_color,
        n_layers_trunk,
        len(nh_o) - 1,
        len(nh_c) - 1,
        nh_c[-1],
    )


# --------------------------------------------------------------------------------------
# splatter MLP
# --------------------------------------------------------------------------------------


def flatten_splatter_params(weights, biases):
    """Pack one MLP: ``(mlp_params, n_hidden)``."""
    _check_mlp_chain(weights, biases)
    mlp_params = torch.cat([t.reshape(-1) for g in (weights, biases) for t in g], dim=0).contiguous()
    return mlp_params, _n_hidden_of(weights, mlp_params.device)


def init_splatter_params(
    device, n_layers: int, input_chn: int = 32, hidden_chn: int = 32, out_chn: int = 16
) -> SplatterParams:
    """Xavier-initialise the MLP-Splatter's MLP (reference mlp_utils.py:298-339)."""
    w, b = _xavier_mlp(n_layers, input_chn, hidden_chn, out_chn, device)
    return SplatterParams(*flatten_splatter_params(w, b))
