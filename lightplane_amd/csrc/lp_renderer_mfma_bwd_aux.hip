// lp_renderer_mfma_bwd_aux.hip -- the LP_ARITH_FP32 instantiations of the tuned Renderer backward (lp_renderer_mfma_bwd.h;
// LpRendererArgs.arithmetic): three limbs in the dX chains, fp32 weight-gradient quadrants -- the reference's arithmetic
// (triton_src/shared/const.py:9), selectable per call.  Four-wave workgroups, one sweep per ray.
#include "lp_renderer_mfma_bwd.h"

namespace lp {

#ifndef LP_DEV_ONE
template <int C, int GM>
static int launch_f32(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
  if (a.march.num_samples_inf > LdsBf3Rm<C>::N_INF || a.seg_prefix)
    return set_error(LP_EUNSUPPORTED, "LP_ARITH_FP32: four-wave workgroups only (<= 64 beyond-far samples, one sweep per ray)");
  const bool plain = bwd_is_plain(a);
  if (a.color_chn <= 3)
    return plain ? launch_bwd3w<C, GM, true, 3, 4, false, false, true>(a, mp, stream) : launch_bwd3w<C, GM, false, 3, 4, false, false, true>(a, mp, stream);
  return plain ? launch_bwd3w<C, GM, true, 4, 4, false, false, true>(a, mp, stream) : launch_bwd3w<C, GM, false, 4, 4, false, false, true>(a, mp, stream);
}
template <int C>
static int launch_f32_gm(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  switch (gm) {
    case GM_TRIPLANE: return launch_f32<C, GM_TRIPLANE>(a, mp, stream);
    case GM_VOXEL: return launch_f32<C, GM_VOXEL>(a, mp, stream);
    default: return launch_f32<C, GM_GENERIC>(a, mp, stream);
  }
}
#endif
int renderer_bwd_bf3_f32(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
#ifdef LP_DEV_ONE
  return set_error(LP_EUNSUPPORTED, "LP_DEV_ONE build");
#else
  return a.grid.channels == 16 ? launch_f32_gm<16>(a, mp, gm, stream) : launch_f32_gm<32>(a, mp, gm, stream);
#endif
}

}  // namespace lp
