// lp_api.hip -- the extern "C" surface of liblightplane_hip.so (see include/lightplane_hip.h).
// Argument validation, kernel selection and error reporting; no device code here.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "lp_host.h"
#include "lp_mfma_common.h"
#if __has_include("build/lp_build_gen.h")
#include "build/lp_build_gen.h"  // written by build.py: LP_BUILD_SRC_HASH, LP_BUILD_FLAGS_JSON
#endif
#ifndef LP_BUILD_SRC_HASH
#define LP_BUILD_SRC_HASH "unknown (not built by lightplane_amd/csrc/build.py)"
#define LP_BUILD_FLAGS_JSON "null"
#endif

namespace lp {

static thread_local char g_err[512] = "";
thread_local uint32_t* g_relu_dump = nullptr;  // test hook, see lp_host.h
const char* volatile g_last_backward = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return LP_OK;
  return set_error((int)e, "%s: %s", what, hipGetErrorString(e));
}

static int64_t mlp_numel(const LpMlp& m) {
  int64_t n = 0;
  for (int l = 0; l < m.n_layers; ++l) n += (int64_t)m.dims[l] * m.dims[l + 1] + m.dims[l + 1];
  return n;
}

static int check_mlp(const char* name, const LpMlp& m, bool may_be_empty) {
  if (m.n_layers < 0 || m.n_layers > LP_MAX_LAYERS)
    return set_error(LP_EINVAL, "%s MLP: n_layers %d outside [0, %d]", name, m.n_layers, LP_MAX_LAYERS);
  if (m.n_layers == 0 && !may_be_empty) return set_error(LP_EINVAL, "%s MLP has no layers", name);
  for (int l = 0; l <= m.n_layers && m.n_layers > 0; ++l)
    if (m.dims[l] < 1 || m.dims[l] > LP_MAX_WIDTH)
      return set_error(LP_EUNSUPPORTED, "%s MLP: width %d of layer %d outside [1, %d]", name, m.dims[l], l,
                       LP_MAX_WIDTH);
  return LP_OK;
}

static int check_grid_list(const char* name, const LpGridList& gl, bool required) {
  if (gl.n_grids < 0 || gl.n_grids > LP_MAX_GRIDS)
    return set_error(LP_EINVAL, "%s: n_grids %d outside [0, %d]", name, gl.n_grids, LP_MAX_GRIDS);
  if (gl.n_grids == 0) return required ? set_error(LP_EINVAL, "%s: empty grid-list", name) : LP_OK;
  if (gl.channels < 1 || gl.channels > LP_MAX_WIDTH)
    return set_error(LP_EUNSUPPORTED, "%s: %d channels outside [1, %d]", name, gl.channels, LP_MAX_WIDTH);
  const int B = gl.grids[0].B;
  for (int g = 0; g < gl.n_grids; ++g) {
    const LpGrid& d = gl.grids[g];
    if (d.B < 1 || d.D < 1 || d.H < 1 || d.W < 1)
      return set_error(LP_EINVAL, "%s[%d]: non-positive extent [%d,%d,%d,%d]", name, g, d.B, d.D, d.H, d.W);
    if (d.B != B) return set_error(LP_EINVAL, "%s[%d]: batch %d != %d", name, g, d.B, B);
    const int ns = (d.D > 1) + (d.H > 1) + (d.W > 1);
    if (ns < 2)
      return set_error(LP_EINVAL, "%s[%d]: Unexpected n non-singular dim of input grid (%d)", name, g, ns);
    const int64_t rows = (int64_t)d.B * d.D * d.H * d.W;
    if (d.row_offset < 0) return set_error(LP_EINVAL, "%s[%d]: negative row_offset", name, g);
    // a grid without its own pointer lives in the flat tensor: it has to fit
    if (!d.data && d.row_offset + rows > gl.n_rows)
      return set_error(LP_EINVAL, "%s[%d]: rows [%lld, %lld) outside the flat tensor (%lld rows)", name, g,
                       (long long)d.row_offset, (long long)(d.row_offset + rows), (long long)gl.n_rows);
  }
  return LP_OK;
}

// every grid gets an explicit base pointer: its own (zero-copy list) or the flat tensor's
static bool normalize_grid_list(LpGridList& gl) {
  bool all = true;
  for (int g = 0; g < gl.n_grids; ++g) {
    if (!gl.grids[g].data) gl.grids[g].data = gl.data;
    all = all && gl.grids[g].data != nullptr;
  }
  for (int g = gl.n_grids > 0 ? gl.n_grids : 0; g < LP_MAX_GRIDS; ++g) gl.grids[g].data = nullptr;
  return all;
}
// gradient buffers: entry g = its own buffer or the flat one; either every grid has one or none has
static int normalize_grad_list(const char* name, float** list, float* flat, int n_grids) {
  int have = 0;
  for (int g = 0; g < LP_MAX_GRIDS; ++g) {
    if (g >= n_grids) { list[g] = nullptr; continue; }
    if (!list[g]) list[g] = flat;
    have += list[g] != nullptr;
  }
  if (have != 0 && have != n_grids)
    return set_error(LP_EINVAL, "%s: gradient buffers given for %d of %d grids (all or none)", name, have, n_grids);
  return LP_OK;
}

static int check_rays(const LpRays& r, bool need_encoding) {
  if (r.n_rays < 0) return set_error(LP_EINVAL, "n_rays %lld < 0", (long long)r.n_rays);
  if (r.n_rays == 0) return LP_OK;
  if (!r.directions || !r.origins || !r.grid_idx || !r.near_t || !r.far_t)
    return set_error(LP_ENULL, "rays: directions/origins/grid_idx/near/far must be non-NULL");
  if (need_encoding && !r.encoding) return set_error(LP_ENULL, "rays.encoding is NULL");
  if (r.row_length < 0) return set_error(LP_EINVAL, "rays.row_length %d < 0 (0 = unknown)", r.row_length);
  if (r.encoding_dim < 0 || r.encoding_dim > LP_MAX_WIDTH)
    return set_error(LP_EUNSUPPORTED, "rays.encoding_dim %d outside [0, %d]", r.encoding_dim, LP_MAX_WIDTH);
  return LP_OK;
}

static int check_march(const LpMarch& m) {
  if (m.num_samples < 1) return set_error(LP_EINVAL, "num_samples %d < 1", m.num_samples);
  if (m.num_samples_inf < 0) return set_error(LP_EINVAL, "num_samples_inf %d < 0", m.num_samples_inf);
  return LP_OK;
}

static int check_renderer(const LpRendererArgs& a, bool backward) {
  int rc;
  if ((rc = check_rays(a.rays, true))) return rc;
  if ((rc = check_march(a.march))) return rc;
  if ((rc = check_grid_list("grid", a.grid, true))) return rc;
  if ((rc = check_grid_list("color_grid", a.color_grid, false))) return rc;
  const bool two = a.color_grid.n_grids > 0;
  if (two) {
    if (a.color_grid.channels != a.grid.channels || a.color_grid.grids[0].B != a.grid.grids[0].B)
      return set_error(LP_EINVAL, "color_grid must share batch size and channel count with grid");
    if (a.trunk.n_layers != 0)
      return set_error(LP_EINVAL, "mlp_n_layers_trunk has to be 0 when use_separate_color_grid");
  }
  if (!(a.stop_neg_log_t >= 0.0f)) return set_error(LP_EINVAL, "stop_neg_log_t must be >= 0 (0 = no early termination)");
  if (a.arithmetic != LP_ARITH_DEFAULT && a.arithmetic != LP_ARITH_FP32)
    return set_error(LP_EINVAL, "arithmetic %d is neither LP_ARITH_DEFAULT nor LP_ARITH_FP32", a.arithmetic);
  if (a.march_order != LP_MARCH_RAYS_PER_WAVE && a.march_order != LP_MARCH_SAMPLES_PER_WAVE)
    return set_error(LP_EINVAL, "march_order %d is neither LP_MARCH_RAYS_PER_WAVE nor LP_MARCH_SAMPLES_PER_WAVE", a.march_order);
  if (backward && a.stop_neg_log_t > 0.0f && !a.neg_log_t_ckpt)
    return set_error(LP_ENULL, "early termination needs neg_log_t_ckpt in the backward (it records where the forward stopped)");
  if ((rc = check_mlp("trunk", a.trunk, true))) return rc;
  if ((rc = check_mlp("opacity", a.opacity, false))) return rc;
  if ((rc = check_mlp("color", a.color, false))) return rc;
  const int C = a.grid.channels;
  int head_in = C;
  if (a.trunk.n_layers > 0) {
    if (a.trunk.dims[0] != C)
      return set_error(LP_EINVAL, "trunk MLP input width %d != grid channels %d", a.trunk.dims[0], C);
    head_in = a.trunk.dims[a.trunk.n_layers];
  }
  if (a.opacity.dims[0] != head_in || a.color.dims[0] != head_in)
    return set_error(LP_EINVAL, "head input widths (%d, %d) != %d", a.opacity.dims[0], a.color.dims[0], head_in);
  if (a.opacity.dims[a.opacity.n_layers] != 1)
    return set_error(LP_EINVAL, "opacity MLP must end in 1 output, got %d", a.opacity.dims[a.opacity.n_layers]);
  if (a.color_chn < 1 || a.color_chn > a.color.dims[a.color.n_layers])
    return set_error(LP_EINVAL, "color_chn %d outside [1, %d]", a.color_chn, a.color.dims[a.color.n_layers]);
  if (a.rays.encoding_dim != head_in)
    return set_error(LP_EINVAL, "ray_encoding should have the same dimension as dim_in_color (%d != %d)",
                     a.rays.encoding_dim, head_in);
  const int64_t expect = mlp_numel(a.trunk) + mlp_numel(a.opacity) + mlp_numel(a.color);
  if (expect != a.n_mlp_params)
    return set_error(LP_EINVAL, "The number of elements in mlp param should be %lld. Got %lld instead.",
                     (long long)expect, (long long)a.n_mlp_params);
  if (a.trunk.offset != 0 || a.opacity.offset != mlp_numel(a.trunk) ||
      a.color.offset != mlp_numel(a.trunk) + mlp_numel(a.opacity))
    return set_error(LP_EINVAL, "MLP offsets do not follow the trunk|opacity|color flat layout");
  if (!a.mlp_params) return set_error(LP_ENULL, "mlp_params is NULL");
  if (a.scaffold) {
    const LpGrid& s = a.scaffold_shape;
    if (s.B != a.grid.grids[0].B || s.D < 1 || s.H < 1 || s.W < 1)
      return set_error(LP_EINVAL, "scaffold shape [%d,%d,%d,%d] incompatible with grid batch %d", s.B, s.D, s.H,
                       s.W, a.grid.grids[0].B);
  }
  if (a.rays.n_rays > 0) {
    if (!backward && (!a.ray_length || !a.neg_log_t || !a.feature))
      return set_error(LP_ENULL, "forward outputs (ray_length, neg_log_t, feature) must be non-NULL");
    if (backward && !a.neg_log_t)
      return set_error(LP_ENULL, "backward needs neg_log_t saved by the forward pass");
  }
  return LP_OK;
}

static int check_splatter(const LpSplatterArgs& a, bool backward) {
  int rc;
  if (a.march_order != LP_MARCH_RAYS_PER_WAVE && a.march_order != LP_MARCH_SAMPLES_PER_WAVE)
    return set_error(LP_EINVAL, "march_order %d is neither LP_MARCH_RAYS_PER_WAVE nor LP_MARCH_SAMPLES_PER_WAVE", a.march_order);
  if ((rc = check_rays(a.rays, true))) return rc;
  if ((rc = check_march(a.march))) return rc;
  if ((rc = check_grid_list("out", a.out, true))) return rc;
  const bool use_mlp = a.mlp.n_layers > 0;
  if (use_mlp) {
    // MLP-Splatter: MLP(sample(input_grid) + encoding) is splatted (reference lightplane_splatter.py:167-338)
    if ((rc = check_mlp("splatter", a.mlp, false))) return rc;
    if ((rc = check_grid_list("input_grid", a.input_grid, true))) return rc;
    if (a.input_grid.grids[0].B != a.out.grids[0].B)
      return set_error(LP_EINVAL, "input_grid batch %d != output grid batch %d", a.input_grid.grids[0].B,
                       a.out.grids[0].B);
    if (a.mlp.dims[0] != a.input_grid.channels || a.mlp.dims[0] != a.rays.encoding_dim)
      return set_error(LP_EINVAL, "MLP input width %d must equal input_grid channels %d and encoding width %d",
                       a.mlp.dims[0], a.input_grid.channels, a.rays.encoding_dim);
    if (a.mlp.dims[a.mlp.n_layers] != a.out.channels)
      return set_error(LP_EINVAL, "MLP output width %d != output grid channels %d", a.mlp.dims[a.mlp.n_layers],
                       a.out.channels);
    if (a.mlp.offset != 0 || mlp_numel(a.mlp) != a.n_mlp_params)
      return set_error(LP_EINVAL, "The number of elements in mlp param should be %lld. Got %lld instead.",
                       (long long)mlp_numel(a.mlp), (long long)a.n_mlp_params);
    if (!a.mlp_params) return set_error(LP_ENULL, "mlp_params is NULL");
  } else if (a.rays.encoding_dim != a.out.channels) {
    return set_error(LP_EINVAL, "splatting feature width %d != output grid channels %d", a.rays.encoding_dim,
                     a.out.channels);
  }
  if (a.rays.n_rays > 0) {
    if (!backward && (!a.out_feature || !a.out_weight))
      return set_error(LP_ENULL, "out_feature / out_weight must be non-NULL");
    if (backward && (!a.grad_out || !a.weight))
      return set_error(LP_ENULL, "grad_out / weight must be non-NULL");
    if (backward && !use_mlp && !a.grad_encoding) return set_error(LP_ENULL, "grad_encoding must be non-NULL");
  }
  return LP_OK;
}

}  // namespace lp

using namespace lp;

extern "C" {

#ifdef LP_EXPERIMENTS  // a build with experiment switches (LP_X_*) identifies itself: negative version, refused by the binding
int lp_version(void) { return -LP_VERSION; }
#else
int lp_version(void) { return LP_VERSION; }
#endif

const char* lp_last_error(void) { return g_err; }

const char* lp_build_info(void) {
  static char info[4096];
  static const bool once = [] {
    snprintf(info, sizeof(info),
             "{\"version\": %d, \"src_hash\": \"%s\", \"test_hooks\": %s, \"tuned_bwd\": %s, \"loop_bwd_deep\": %s, "
             "\"loop_bwd_shallow\": %s, \"mlp_splatter_bwd\": %s, \"forward\": \"bf16x3 (three exact bf16 limbs per fp32 operand, six limb "
             "products, fp32 accumulation) on v_mfma_f32_32x32x16_bf16; generic kernels: fp32 FMA\", \"flags\": %s}",
             lp_version(), LP_BUILD_SRC_HASH, build_info_tuned_bwd_aux(), build_info_tuned_bwd(), build_info_loop_deep(),
             build_info_loop_shallow(), build_info_splatter_mlp(), LP_BUILD_FLAGS_JSON);
    return true;
  }();
  (void)once;
  return info;
}

int lp_abi_sizeof(int which) {
  switch (which) {
    case 0: return (int)sizeof(LpGrid);
    case 1: return (int)sizeof(LpGridList);
    case 2: return (int)sizeof(LpRays);
    case 3: return (int)sizeof(LpMarch);
    case 4: return (int)sizeof(LpMlp);
    case 5: return (int)sizeof(LpRendererArgs);
    case 6: return (int)sizeof(LpSplatterArgs);
    case 7: return (int)sizeof(LpRayEmbedArgs);
    default: return -1;
  }
}

// kernel selection: 1 = the tuned bf16x3 kernels of the default shape (2/2/2 x 32), 3 = layer-looped bf16x3 MFMA family (1-4 layers
// per MLP, widths 16 / 32 / 64), 0 = the shape-generic kernel.  (2 was the fp32-MFMA family of 2/2/2 x 64, retired in 0.2.4: the
// two-block looped kernels with the eight-wave forward measure 1-2 % faster forward + backward and 25-29 % faster forward on it --
// profiles/r04_h64_looped_vs_wide.txt.)  LP_LOOP=1 (developer knob, read once): the layer-looped family also for the shape family 1
// covers (A/B, test coverage).
// LP_ARITH_FP32 (LpRendererArgs.arithmetic): the tuned family where it has such instantiations, the generic fp32 kernels otherwise.
static int select_renderer(const LpRendererArgs& a, const char** why) {
  const char* w32 = "";
  const char* wl = "";
  if (a.arithmetic == LP_ARITH_FP32) {
    if (renderer_mfma_f32_supported(a)) return 1;
    *why = "LP_ARITH_FP32 outside the tuned family's four-wave instantiations: shape-generic fp32 kernels";
    return 0;
  }
  static const bool force_loop = getenv("LP_LOOP") != nullptr && atoi(getenv("LP_LOOP")) != 0;
  const bool loop_ok = renderer_loop_supported(a, &wl) && renderer_loop_fits(a);
  if (force_loop && loop_ok) return 3;
  if (renderer_mfma_supported(a, &w32)) return 1;
  if (loop_ok) return 3;
  *why = wl[0] ? wl : "weight images of this decoder exceed the 160 KB LDS";
  return 0;
}

int lp_renderer_kernel_family(const LpRendererArgs* args) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  const char* why = "";
  return select_renderer(*args, &why);
}

int lp_renderer_backward_segments(const LpRendererArgs* args) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  const char* why = "";
  if (args->kernel == LP_KERNEL_GENERIC) return 1;
  const int fam = select_renderer(*args, &why);
  return fam == 1 ? renderer_mfma_segments(*args) : fam == 3 ? renderer_loop_segments(*args) : 1;
}

// MLP-Splatter: 3 = layer-looped bf16x3 family (2-4 layers, widths 16 / 32 / 64), 0 = generic.  (2 was the two-layer fp32-MFMA
// family [E,32,Cout] of rounds 1-3, retired in round 4: the looped family's two-waves-per-SIMD backward measures within 1 % of
// it or faster on every shape it covered -- profiles/r04_loop_shallow_ab.txt.)
static int splatter_mlp_family(const LpSplatterArgs& a) { return splatter_mlp_loop_supported(a) ? 3 : 0; }

int lp_splatter_kernel_family(const LpSplatterArgs* args) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  if (args->mlp.n_layers > 0) return splatter_mlp_family(*args);
  const int C = args->out.channels;
  return ((C == 16 || C == 32 || C == 64) && args->out.n_rows < ((int64_t)1 << 31)) ? 1 : 0;
}

// copy of the caller's arguments with every per-grid pointer made explicit (what the kernels read)
static int normalized_renderer_args(const LpRendererArgs* args, bool backward, LpRendererArgs& a) {
  int rc = check_renderer(*args, backward);
  if (rc) return rc;
  a = *args;
  const bool have_grid = normalize_grid_list(a.grid);
  const bool have_cgrid = normalize_grid_list(a.color_grid);
  if (a.rays.n_rays > 0 && !have_grid) return set_error(LP_ENULL, "grid.data is NULL (and a grid has no pointer of its own)");
  if (a.rays.n_rays > 0 && !have_cgrid) return set_error(LP_ENULL, "color_grid.data is NULL (and a grid has no pointer of its own)");
  if ((rc = normalize_grad_list("grad_grid", a.grad_grid_list, backward ? a.grad_grid : nullptr, a.grid.n_grids))) return rc;
  if ((rc = normalize_grad_list("grad_color_grid", a.grad_color_grid_list, backward ? a.grad_color_grid : nullptr,
                                a.color_grid.n_grids)))
    return rc;
  if (a.alpha_mode < 0 || a.alpha_mode > 2) return set_error(LP_EINVAL, "alpha_mode %d outside 0..2", a.alpha_mode);
  if ((a.alpha || a.grad_alpha) && a.alpha_mode == 0)
    return set_error(LP_EINVAL, "alpha / grad_alpha given but alpha_mode is 0");
  // the segment sums are written / used only where lp_renderer_backward_segments() says so (same rule both ways)
  if (a.seg_prefix && lp_renderer_backward_segments(&a) <= 1) a.seg_prefix = nullptr;
  return LP_OK;
}

int lp_renderer_forward(const LpRendererArgs* args, void* stream) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  LpRendererArgs a;
  int rc = normalized_renderer_args(args, false, a);
  if (rc) return rc;
  const char* why = "";
  const int fam = select_renderer(a, &why);
  if (a.kernel == LP_KERNEL_MFMA && fam == 0)
    return set_error(LP_EUNSUPPORTED, "MFMA renderer kernel unavailable for this shape: %s", why);
  if (fam == 1 && a.kernel != LP_KERNEL_GENERIC) return renderer_forward_mfma(a, (hipStream_t)stream);
  if (fam == 3 && a.kernel != LP_KERNEL_GENERIC) return renderer_forward_loop(a, (hipStream_t)stream);
  return renderer_forward_generic(a, (hipStream_t)stream);
}

int lp_renderer_backward(const LpRendererArgs* args, void* stream) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  LpRendererArgs a;
  int rc = normalized_renderer_args(args, true, a);
  if (rc) return rc;
  const char* why = "";
  const int fam = select_renderer(a, &why);
  if (a.kernel == LP_KERNEL_MFMA && fam == 0)
    return set_error(LP_EUNSUPPORTED, "MFMA renderer kernel unavailable for this shape: %s", why);
  if (fam == 1 && a.kernel != LP_KERNEL_GENERIC) return renderer_backward_mfma(a, (hipStream_t)stream);
  g_last_backward = (fam == 3 && a.kernel != LP_KERNEL_GENERIC) ? "layer-looped family" : "shape-generic kernels";
  if (fam == 3 && a.kernel != LP_KERNEL_GENERIC) return renderer_backward_loop(a, (hipStream_t)stream);
  return renderer_backward_generic(a, (hipStream_t)stream);
}

int lp_renderer_corner_rows(const LpRendererArgs* args, int64_t* rows, void* stream) {
  if (!args || !rows) return set_error(LP_ENULL, "args / rows is NULL");
  int rc;
  if ((rc = check_rays(args->rays, false))) return rc;
  if ((rc = check_march(args->march))) return rc;
  if ((rc = check_grid_list("grid", args->grid, true))) return rc;
  return renderer_corner_rows_launch(*args, rows, (hipStream_t)stream);
}

int lp_renderer_relu_dump_words(const LpRendererArgs* args) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  const char* why = "";
  const int fam = args->kernel == LP_KERNEL_GENERIC ? 0 : select_renderer(*args, &why);
  if (args->arithmetic != LP_ARITH_DEFAULT) return set_error(LP_EUNSUPPORTED, "relu dump: the LP_ARITH_FP32 instantiations have no dump twin");
  if (fam == 1) {
    if (args->march.num_samples_inf > 64) return set_error(LP_EUNSUPPORTED, "relu dump: the tuned family's eight-wave workgroups (> 64 beyond-far samples) have no dump twin");
    return 5;
  }
  if (fam == 3) return renderer_loop_dump_words(*args);
  return renderer_generic_dump_words(*args);  // (the shape-generic backward: ceil(widest site / 32) words per site)
}

int lp_renderer_backward_relu_dump(const LpRendererArgs* args, uint32_t* dump, int64_t dump_words, void* stream) {
  if (!args || !dump) return set_error(LP_ENULL, "args / dump is NULL");
  LpRendererArgs a;
  int rc = normalized_renderer_args(args, true, a);
  if (rc) return rc;
  const int w = lp_renderer_relu_dump_words(&a);
  if (w < 0) return w;
  const int64_t want = args->rays.n_rays * (int64_t)(args->march.num_samples + args->march.num_samples_inf) * w;
  if (dump_words != want) return set_error(LP_EINVAL, "relu dump: %lld words given, [n_rays][S_tot][%d] = %lld needed", (long long)dump_words, w, (long long)want);
  g_relu_dump = dump;
  rc = lp_renderer_backward(args, stream);
  g_relu_dump = nullptr;
  return rc;
}

static int normalized_splatter_args(const LpSplatterArgs* args, bool backward, LpSplatterArgs& a) {
  int rc = check_splatter(*args, backward);
  if (rc) return rc;
  a = *args;
  if (a.mlp.n_layers > 0) {
    if (!normalize_grid_list(a.input_grid) && a.rays.n_rays > 0)
      return set_error(LP_ENULL, "input_grid.data is NULL (and a grid has no pointer of its own)");
    if ((rc = normalize_grad_list("grad_input_grid", a.grad_input_grid_list, backward ? a.grad_input_grid : nullptr,
                                  a.input_grid.n_grids)))
      return rc;
  }
  return LP_OK;
}

int lp_splatter_forward(const LpSplatterArgs* args_, void* stream) {
  if (!args_) return set_error(LP_ENULL, "args is NULL");
  LpSplatterArgs a_;
  int rc = normalized_splatter_args(args_, false, a_);
  if (rc) return rc;
  const LpSplatterArgs* args = &a_;
  if (args->mlp.n_layers > 0) {
    const int fam = splatter_mlp_family(*args);
    if (args->kernel == LP_KERNEL_MFMA && fam == 0)
      return set_error(LP_EUNSUPPORTED, "MFMA MLP-splatter kernel unavailable for this shape");
    if (fam == 3 && args->kernel != LP_KERNEL_GENERIC) return splatter_mlp_forward_loop(*args, (hipStream_t)stream);
    return splatter_mlp_forward_launch(*args, (hipStream_t)stream);
  }
  return splatter_forward_launch(*args, (hipStream_t)stream);
}

int lp_splatter_normalize(float* feature, const float* weight, int64_t n_rows, int32_t channels, void* stream) {
  if (n_rows < 0 || channels < 1) return set_error(LP_EINVAL, "normalize: bad shape [%lld, %d]", (long long)n_rows, channels);
  if (n_rows > 0 && (!feature || !weight)) return set_error(LP_ENULL, "normalize: NULL buffer");
  return splatter_normalize_launch(feature, weight, n_rows, channels, (hipStream_t)stream);
}

int lp_splatter_backward(const LpSplatterArgs* args_, void* stream) {
  if (!args_) return set_error(LP_ENULL, "args is NULL");
  LpSplatterArgs a_;
  int rc = normalized_splatter_args(args_, true, a_);
  if (rc) return rc;
  const LpSplatterArgs* args = &a_;
  if (args->mlp.n_layers > 0) {
    const int fam = splatter_mlp_family(*args);
    if (args->kernel == LP_KERNEL_MFMA && fam == 0)
      return set_error(LP_EUNSUPPORTED, "MFMA MLP-splatter kernel unavailable for this shape");
    if (fam == 3 && args->kernel != LP_KERNEL_GENERIC) return splatter_mlp_backward_loop(*args, (hipStream_t)stream);
    return splatter_mlp_backward_launch(*args, (hipStream_t)stream);
  }
  return splatter_backward_launch(*args, (hipStream_t)stream);
}

/* developer / test hook (not part of include/lightplane_hip.h): which Renderer backward the process launched last --
 * "tuned family, rays per wavefront", "tuned family, samples per wavefront (transposed march)", "layer-looped family",
 * "shape-generic kernels" (static strings) */
const char* lp_debug_last_renderer_backward(void) { return g_last_backward; }

/* developer hook (not part of include/lightplane_hip.h): per-phase cycle totals of the MFMA backward,
 * only in builds with -DLP_PHASE_TIMING (returns -1 otherwise) */
int lp_debug_phase_cycles(unsigned long long* out16) { return debug_phase_cycles(out16); }

static int check_ray_embed(const LpRayEmbedArgs& a, bool backward) {
  if (a.n_rays < 0) return set_error(LP_EINVAL, "n_rays %lld < 0", (long long)a.n_rays);
  if (a.n_harmonics < 0 || a.n_harmonics > 10) return set_error(LP_EUNSUPPORTED, "n_harmonics %d outside [0, 10]", a.n_harmonics);
  if (a.out_dim < 1 || a.out_dim > LP_MAX_WIDTH) return set_error(LP_EUNSUPPORTED, "out_dim %d outside [1, %d]", a.out_dim, LP_MAX_WIDTH);
  if (a.n_rays == 0) return LP_OK;
  if (!a.directions) return set_error(LP_ENULL, "directions is NULL");
  if (!backward && (!a.weight || !a.bias || !a.out)) return set_error(LP_ENULL, "weight / bias / out must be non-NULL");
  if (backward && (!a.grad_out || (!a.grad_weight && !a.grad_bias)))
    return set_error(LP_ENULL, "grad_out and at least one of grad_weight / grad_bias must be non-NULL");
  return LP_OK;
}

int lp_ray_embedding_forward(const LpRayEmbedArgs* args, void* stream) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  const int rc = check_ray_embed(*args, false);
  if (rc) return rc;
  return ray_embedding_forward_launch(*args, (hipStream_t)stream);
}

int lp_ray_embedding_backward(const LpRayEmbedArgs* args, void* stream) {
  if (!args) return set_error(LP_ENULL, "args is NULL");
  const int rc = check_ray_embed(*args, true);
  if (rc) return rc;
  return ray_embedding_backward_launch(*args, (hipStream_t)stream);
}

int lp_hash_randn(const int32_t* x1, const int32_t* x2, float* out, int64_t n, int32_t seed, void* stream) {
  if (n < 0) return set_error(LP_EINVAL, "n < 0");
  if (n > 0 && (!x1 || !x2 || !out)) return set_error(LP_ENULL, "hash_randn: NULL buffer");
  return hash_randn_launch(x1, x2, out, n, seed, (hipStream_t)stream);
}

}  // extern "C"
