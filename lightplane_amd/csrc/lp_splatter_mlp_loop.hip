// lp_splatter_mlp_loop.hip -- host side of the layer-looped bf16x3 MFMA family of the MLP-Splatter + its forward kernels and the
// backward instantiations for up to four layers / widths of 64 (one wave per SIMD).  Device code: lp_splatter_mlp_loop.h; the
// two-layer backward at two waves per SIMD: lp_splatter_mlp_loop_shallow.hip.
#include "lp_splatter_mlp_loop.h"

namespace lp {

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
bool splatter_mlp_loop_supported(const LpSplatterArgs& a) {
  const LpMlp& m = a.mlp;
  if (m.n_layers < 2 || m.n_layers > SLOOP_MAX) return false;
  if (!width_ok(m.dims[0])) return false;
  for (int l = 1; l < m.n_layers; ++l)
    if (!width_ok(m.dims[l]) || m.dims[l] != m.dims[1]) return false;
  const int CO = m.dims[m.n_layers];
  if (CO != 16 && CO != 32) return false;
  if (!grid_list_rows_ok(a.input_grid)) return false;  // 32-bit row indices
  if (a.out.n_rows >= (int64_t)1 << 31) return false;
  if (a.march.num_samples_inf > LOOP_N_INF) return false;
  const SplatLoopParams p = sloop_params(a, sloop_nb(a));
  return (size_t)p.img_end + (size_t)WAVES * SplatLoopTile::PER_WAVE * 4 <= 160 * 1024;
}

#define LP_DISPATCH_SL(KERNEL, BWD)                                                                        \
  do {                                                                                                     \
    const int E = a.mlp.dims[0], CO = a.mlp.dims[a.mlp.n_layers], NB = sloop_nb(a);                        \
    if (E == 16) {                                                                                         \
      if (CO == 16) rc = NB == 1 ? sloop_launch(KERNEL<16, 16, 1>, a, stream, BWD) : sloop_launch(KERNEL<16, 16, 2>, a, stream, BWD); \
      else rc = NB == 1 ? sloop_launch(KERNEL<16, 32, 1>, a, stream, BWD) : sloop_launch(KERNEL<16, 32, 2>, a, stream, BWD);          \
    } else if (E == 32) {                                                                                  \
      if (CO == 16) rc = NB == 1 ? sloop_launch(KERNEL<32, 16, 1>, a, stream, BWD) : sloop_launch(KERNEL<32, 16, 2>, a, stream, BWD); \
      else rc = NB == 1 ? sloop_launch(KERNEL<32, 32, 1>, a, stream, BWD) : sloop_launch(KERNEL<32, 32, 2>, a, stream, BWD);          \
    } else {                                                                                               \
      if (CO == 16) rc = sloop_launch(KERNEL<64, 16, 2>, a, stream, BWD);                                  \
      else rc = sloop_launch(KERNEL<64, 32, 2>, a, stream, BWD);                                           \
    }                                                                                                      \
  } while (0)

int splatter_mlp_forward_loop(const LpSplatterArgs& a, hipStream_t stream) {
  if (a.rays.n_rays == 0) return LP_OK;
  int rc;
  const int E = a.mlp.dims[0], CO = a.mlp.dims[a.mlp.n_layers], NB = sloop_nb(a);
#define LP_SL_FWD(EV, COV) rc = (NB == 1 && EV < 64) ? sloop_launch_fwd<EV, COV, (EV < 64 ? 1 : 2)>(a, stream) : sloop_launch_fwd<EV, COV, 2>(a, stream)
  if (E == 16) {
    if (CO == 16) LP_SL_FWD(16, 16); else LP_SL_FWD(16, 32);
  } else if (E == 32) {
    if (CO == 16) LP_SL_FWD(32, 16); else LP_SL_FWD(32, 32);
  } else {
    if (CO == 16) LP_SL_FWD(64, 16); else LP_SL_FWD(64, 32);
  }
#undef LP_SL_FWD
  if (rc) return rc;
  return check_launch("splat_mlp_fwd_loop");
}

int splatter_mlp_backward_loop(const LpSplatterArgs& a, hipStream_t stream) {
  if (a.rays.n_rays == 0) return LP_OK;
  int rc;
  static const bool no_shallow = getenv("LP_LOOP_NO_SHALLOW") != nullptr;  // A/B: the deep instantiation for every shape
  if (a.mlp.n_layers <= 2 && sloop_nb(a) == 1 && !no_shallow) {
    if ((rc = splatter_mlp_backward_loop_shallow(a, stream))) return rc;
    return check_launch("splat_mlp_bwd_loop (two layers)");
  }
  LP_DISPATCH_SL(splat_mlp_bwd_loop, true);
  if (rc) return rc;
  return check_launch("splat_mlp_bwd_loop");
}

// what the MLP-Splatter's looped backward computes in (lp_build_info): the dX chains keep three limbs; dW per translation unit
const char* build_info_splatter_mlp() {
  return "{\"dx_limbs\": 3, \"dw\": "
#if LP_LOOP_DW_BF16
         "\"deeper than two layers / width 64: two-limb bf16 operands, v_mfma_f32_16x16x32_bf16; two-layer MLPs (two waves per SIMD): "
         "fp32 operands, v_mfma_f32_16x16x4_f32\"}";
#else
         "\"fp32 operands, v_mfma_f32_16x16x4_f32\"}";
#endif
}

}  // namespace lp
