"""ctypes binding of ``liblightplane_hip.so`` (C ABI: ``include/lightplane_hip.h``).

The structures below mirror the header field for field.  There is NO fallback: if
the shared library is missing, ``lib()`` raises -- build it with
``python lightplane_amd/csrc/build.py`` (or ``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from .grids import GridDesc

LP_MAX_GRIDS = 8
LP_MAX_LAYERS = 8
LP_MAX_WIDTH = 128
LP_NLT_CKPT = 32
LP_SEG_LEN = 8   # samples per state record of a small batch's segment-parallel march (lightplane_hip.h; 16 before ABI 0.2.7)


def n_nlt_ckpt(num_samples: int, num_samples_inf: int) -> int:
    """Floats per ray of the -log T checkpoint buffer: one (hi, lo) float pair per checkpoint plus the
    closing pair (see LP_NLT_CKPT in the header)."""
    c = LP_NLT_CKPT
    return 2 * ((num_samples + c - 1) // c + num_samples_inf + 1)

LP_KERNEL_AUTO, LP_KERNEL_GENERIC, LP_KERNEL_MFMA = 0, 1, 2
LP_ARITH_DEFAULT, LP_ARITH_FP32 = 0, 1  # LpRendererArgs.arithmetic (include/lightplane_hip.h)
LP_MARCH_RAYS_PER_WAVE, LP_MARCH_SAMPLES_PER_WAVE = 0, 1  # LpRendererArgs.march_order

_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class LpGrid(C.Structure):
    _fields_ = [("B", C.c_int32), ("D", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("row_offset", C.c_int64), ("data", C.c_void_p)]


class LpGridList(C.Structure):
    _fields_ = [("data", C.c_void_p), ("n_grids", C.c_int32), ("channels", C.c_int32),
                ("n_rows", C.c_int64), ("grids", LpGrid * LP_MAX_GRIDS)]


class LpRays(C.Structure):
    _fields_ = [("n_rays", C.c_int64), ("directions", C.c_void_p), ("origins", C.c_void_p),
                ("grid_idx", C.c_void_p), ("near_t", C.c_void_p), ("far_t", C.c_void_p),
                ("encoding", C.c_void_p), ("encoding_dim", C.c_int32), ("row_length", C.c_int32)]


class LpMarch(C.Structure):
    _fields_ = [("num_samples", C.c_int32), ("num_samples_inf", C.c_int32),
                ("mask_out_of_bounds", C.c_int32), ("contract_coords", C.c_int32),
                ("disparity_at_inf", C.c_double)]


class LpMlp(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * (LP_MAX_LAYERS + 1)),
                ("offset", C.c_int64)]


class LpRendererArgs(C.Structure):
    _fields_ = [
        ("rays", LpRays), ("grid", LpGridList), ("color_grid", LpGridList),
        ("scaffold", C.c_void_p), ("scaffold_shape", LpGrid), ("march", LpMarch),
        ("mlp_params", C.c_void_p), ("n_mlp_params", C.c_int64),
        ("trunk", LpMlp), ("opacity", LpMlp), ("color", LpMlp),
        ("color_chn", C.c_int32), ("gain", C.c_float), ("noise_sigma", C.c_float),
        ("noise_seed", C.c_int32), ("kernel", C.c_int32), ("seg_forward_off", C.c_int32),
        ("ray_length", C.c_void_p), ("neg_log_t", C.c_void_p), ("feature", C.c_void_p),
        ("neg_log_t_ckpt", C.c_void_p),
        ("grad_ray_length", C.c_void_p), ("grad_neg_log_t", C.c_void_p), ("grad_feature", C.c_void_p),
        ("grad_grid", C.c_void_p), ("grad_color_grid", C.c_void_p), ("grad_mlp_params", C.c_void_p),
        ("grad_encoding", C.c_void_p),
        ("grad_grid_list", C.c_void_p * LP_MAX_GRIDS), ("grad_color_grid_list", C.c_void_p * LP_MAX_GRIDS),
        ("bg_color", C.c_void_p), ("alpha", C.c_void_p), ("grad_alpha", C.c_void_p), ("alpha_mode", C.c_int32),
        ("stop_neg_log_t", C.c_float), ("seg_prefix", C.c_void_p), ("arithmetic", C.c_int32), ("march_order", C.c_int32),
    ]


class LpSplatterArgs(C.Structure):
    _fields_ = [
        ("rays", LpRays), ("march", LpMarch), ("out", LpGridList),
        ("out_feature", C.c_void_p), ("out_weight", C.c_void_p),
        ("input_grid", LpGridList), ("mlp_params", C.c_void_p), ("n_mlp_params", C.c_int64),
        ("mlp", LpMlp), ("kernel", C.c_int32), ("march_order", C.c_int32),
        ("grad_out", C.c_void_p), ("weight", C.c_void_p), ("grad_encoding", C.c_void_p),
        ("grad_input_grid", C.c_void_p), ("grad_mlp_params", C.c_void_p),
        ("grad_input_grid_list", C.c_void_p * LP_MAX_GRIDS),
    ]


class LpRayEmbedArgs(C.Structure):
    _fields_ = [
        ("n_rays", C.c_int64), ("directions", C.c_void_p), ("n_harmonics", C.c_int32), ("out_dim", C.c_int32),
        ("weight", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("grad_out", C.c_void_p),
        ("grad_weight", C.c_void_p), ("grad_bias", C.c_void_p),
    ]


_LIB = None
LIB_PATH = os.environ.get("LIGHTPLANE_AMD_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "liblightplane_hip.so")

#: every symbol include/lightplane_hip.h declares
EXPORTS = (
    "lp_version", "lp_last_error", "lp_abi_sizeof", "lp_renderer_forward", "lp_renderer_backward",
    "lp_splatter_forward", "lp_splatter_normalize", "lp_splatter_backward", "lp_hash_randn",
    "lp_renderer_corner_rows", "lp_renderer_kernel_family", "lp_splatter_kernel_family",
    "lp_renderer_backward_segments", "lp_renderer_backward_relu_dump", "lp_renderer_relu_dump_words", "lp_build_info",
    "lp_ray_embedding_forward", "lp_ray_embedding_backward",
)


class LightplaneHipError(RuntimeError):
    """A C-ABI call returned a non-zero code."""


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise loudly if it has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise LightplaneHipError(
            f"{LIB_PATH} not found: the HIP extension has not been built. Run "
            "`python lightplane_amd/csrc/build.py` (there is no CPU / PyTorch fallback)."
        )
    L = C.CDLL(LIB_PATH)
    L.lp_version.restype = C.c_int
    if L.lp_version() < 0 and os.environ.get("LIGHTPLANE_AMD_ALLOW_EXPERIMENTAL") != "1":
        raise LightplaneHipError(f"{LIB_PATH} was built with -DLP_EXPERIMENTS (A/B timing switches, lp_version() = {L.lp_version()}): "
                                 "not a product build; set LIGHTPLANE_AMD_ALLOW_EXPERIMENTAL=1 to time it anyway")
    L.lp_last_error.restype = C.c_char_p
    for name in ("lp_renderer_forward", "lp_renderer_backward"):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(LpRendererArgs), C.c_void_p]
    for name in ("lp_splatter_forward", "lp_splatter_backward"):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(LpSplatterArgs), C.c_void_p]
    L.lp_splatter_normalize.restype = C.c_int
    L.lp_splatter_normalize.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    L.lp_hash_randn.restype = C.c_int
    L.lp_hash_randn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    L.lp_renderer_corner_rows.restype = C.c_int
    L.lp_renderer_corner_rows.argtypes = [C.POINTER(LpRendererArgs), C.c_void_p, C.c_void_p]
    L.lp_renderer_backward_relu_dump.restype = C.c_int
    L.lp_renderer_backward_relu_dump.argtypes = [C.POINTER(LpRendererArgs), C.c_void_p, C.c_int64, C.c_void_p]
    L.lp_renderer_relu_dump_words.restype = C.c_int
    L.lp_renderer_relu_dump_words.argtypes = [C.POINTER(LpRendererArgs)]
    L.lp_build_info.restype = C.c_char_p
    L.lp_renderer_kernel_family.restype = C.c_int
    L.lp_renderer_kernel_family.argtypes = [C.POINTER(LpRendererArgs)]
    L.lp_renderer_backward_segments.restype = C.c_int
    L.lp_renderer_backward_segments.argtypes = [C.POINTER(LpRendererArgs)]
    L.lp_splatter_kernel_family.restype = C.c_int
    L.lp_splatter_kernel_family.argtypes = [C.POINTER(LpSplatterArgs)]
    for name in ("lp_ray_embedding_forward", "lp_ray_embedding_backward"):
        fn = getattr(L, name)
        fn.restype = C.c_int
        fn.argtypes = [C.POINTER(LpRayEmbedArgs), C.c_void_p]
    L.lp_abi_sizeof.restype = C.c_int
    L.lp_abi_sizeof.argtypes = [C.c_int]
    for which, st in enumerate((LpGrid, LpGridList, LpRays, LpMarch, LpMlp, LpRendererArgs, LpSplatterArgs,
                                LpRayEmbedArgs)):
        if L.lp_abi_sizeof(which) != C.sizeof(st):
            raise LightplaneHipError(
                f"ABI mismatch: sizeof({st.__name__}) is {C.sizeof(st)} in the ctypes binding but "
                f"{L.lp_abi_sizeof(which)} in {LIB_PATH}; rebuild the library"
            )
    _LIB = L
    return L


def build_info() -> dict:
    """``lp_build_info()`` of the loaded library, parsed: version, ``src_hash`` (lightplane_amd/csrc/build.py ``source_hash()`` of the
    sources it was compiled from), compiler flags, and the arithmetic every backward family was compiled with."""
    import json
    return json.loads(lib().lp_build_info().decode())


def build_matches_tree() -> Optional[bool]:
    """True / False: the loaded library was built from the csrc/ + header of this tree (``src_hash``); None when the sources are
    not there to compare with (an installed binary)."""
    try:
        from .csrc import build as _b
        want = _b.source_hash()
    except Exception:
        return None
    return build_info().get("src_hash") == want


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().lp_last_error().decode(errors="replace")
        kind = {-1: "invalid argument", -2: "unsupported shape", -3: "NULL pointer"}.get(rc, f"hipError {rc}")
        err = LightplaneHipError(f"{what} failed ({kind}): {msg}")
        if rc in (-1, -3):
            # argument errors surface like the reference's Python-side asserts
            raise AssertionError(str(err))
        raise err


# --------------------------------------------------------------------------------------
# struct builders
# --------------------------------------------------------------------------------------


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    assert t.is_contiguous(), "tensors handed to the HIP library must be contiguous"
    return t.data_ptr()


def check_tensors(device: torch.device, float32: dict, any_dtype: Optional[dict] = None) -> None:
    """Every tensor whose raw pointer goes to the HIP library has to live on ``device`` (the kernels dereference
    the pointers on that GPU: a CPU tensor or a tensor of another GPU would be a memory fault, not an exception)
    and -- for the ``float32`` group -- be fp32 (the kernels reinterpret the bytes).  ``None`` entries are skipped.
    Raises ``AssertionError`` like the reference's Python-side checks (lightplane_renderer.py:402-468)."""
    for group, need_f32 in ((float32, True), (any_dtype or {}, False)):
        for name, t in group.items():
            if t is None:
                continue
            assert torch.is_tensor(t), f"{name} has to be a tensor"
            assert t.device == device, (
                f"{name} is on {t.device} but the grid is on {device}: every tensor of a lightplane_amd call has to "
                f"live on the same GPU (forgot .to(device)?)")
            if need_f32:
                assert t.dtype == torch.float32, f"{name} has to be float32 (got {t.dtype})"


def current_stream(device: torch.device) -> Optional[int]:
    if device.type != "cuda":
        raise LightplaneHipError(
            f"lightplane_amd kernels run on the GPU only (got tensors on '{device}'); "
            "there is no CPU path -- the CPU oracle lives under oracle/ for tests."
        )
    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)  # what Triton's launcher uses: no Stream object built
    if raw is not None:
        return raw(device.index if device.index is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(device).cuda_stream


def make_rays(directions, origins, grid_idx_i32, near, far, encoding, row_length: int = 0) -> LpRays:
    r = LpRays()
    r.row_length = int(row_length)
    r.n_rays = directions.shape[0]
    r.directions, r.origins = ptr(directions), ptr(origins)
    r.grid_idx, r.near_t, r.far_t = ptr(grid_idx_i32), ptr(near), ptr(far)
    r.encoding = ptr(encoding)
    r.encoding_dim = 0 if encoding is None else encoding.shape[1]
    return r


def make_march(num_samples, num_samples_inf, mask_out_of_bounds_samples, contract_coords, disparity_at_inf) -> LpMarch:
    m = LpMarch()
    m.num_samples, m.num_samples_inf = int(num_samples), int(num_samples_inf)
    m.mask_out_of_bounds = int(bool(mask_out_of_bounds_samples))
    m.contract_coords = int(bool(contract_coords))
    m.disparity_at_inf = float(disparity_at_inf)
    return m


def make_grid_list(data, descs: Sequence[GridDesc], channels: int, n_rows: int) -> LpGridList:
    """``data``: the flat ``[rows, C]`` tensor (grids at their ``row_offset``), or a LIST of per-grid tensors (zero-copy
    grid-lists: every grid in its own allocation, addressed from row 0 of its own tensor), or ``None`` (shapes only)."""
    gl = LpGridList()
    assert len(descs) <= LP_MAX_GRIDS, f"at most {LP_MAX_GRIDS} grids per grid-list are supported"
    per_grid = isinstance(data, (list, tuple))
    gl.data = None if per_grid else ptr(data)
    gl.n_grids = len(descs)
    gl.channels = int(channels)
    gl.n_rows = int(n_rows)
    for i, d in enumerate(descs):
        if per_grid:
            gl.grids[i] = LpGrid(d.B, d.D, d.H, d.W, 0, ptr(data[i]))
        else:
            gl.grids[i] = LpGrid(d.B, d.D, d.H, d.W, d.row_offset, None)
    return gl


def fill_ptr_list(field, tensors) -> None:
    """``field``: a ``c_void_p * LP_MAX_GRIDS`` array of an argument struct; entry g = pointer of ``tensors[g]``."""
    for i in range(LP_MAX_GRIDS):
        field[i] = ptr(tensors[i]) if (tensors is not None and i < len(tensors) and tensors[i] is not None) else None


def make_mlp(dims: Sequence[int], offset: int) -> LpMlp:
    m = LpMlp()
    dims = [int(v) for v in dims]
    n_layers = max(len(dims) - 1, 0)
    assert n_layers <= LP_MAX_LAYERS, f"at most {LP_MAX_LAYERS} layers per MLP are supported"
    m.n_layers = n_layers
    for i, v in enumerate(dims):
        m.dims[i] = v
    m.offset = int(offset)
    return m
