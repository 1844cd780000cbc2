// lp_renderer_mfma_bwd_dump.hip -- with -DLP_TEST_HOOKS: the DUMP twins of the tuned Renderer backward (lp_renderer_mfma_bwd.h)
// behind lp_renderer_backward_relu_dump(): four-wave workgroups, full and segmented sweeps, three / four colour channels; compiled
// with the flags of the production translation units (build.py FILE_FLAGS).  A library built without the flag has no twin: the hook
// returns LP_EUNSUPPORTED.
#include "lp_renderer_mfma_bwd.h"

namespace lp {

#if defined(LP_TEST_HOOKS) && !defined(LP_DEV_ONE)
template <int C, int GM>
static int launch_dump(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
  if (a.march.num_samples_inf > LdsBf3Rm<C>::N_INF || a.arithmetic != LP_ARITH_DEFAULT)
    return set_error(LP_EUNSUPPORTED, "relu dump: only the four-wave, default-arithmetic instantiations have a dump twin");
  const bool plain = bwd_is_plain(a);
  if (a.color_chn <= 3) {
    if (a.seg_prefix)
      return plain ? launch_bwd3w<C, GM, true, 3, 4, true, true>(a, mp, stream) : launch_bwd3w<C, GM, false, 3, 4, true, true>(a, mp, stream);
    return plain ? launch_bwd3w<C, GM, true, 3, 4, false, true>(a, mp, stream) : launch_bwd3w<C, GM, false, 3, 4, false, true>(a, mp, stream);
  }
  if (a.seg_prefix)
    return plain ? launch_bwd3w<C, GM, true, 4, 4, true, true>(a, mp, stream) : launch_bwd3w<C, GM, false, 4, 4, true, true>(a, mp, stream);
  return plain ? launch_bwd3w<C, GM, true, 4, 4, false, true>(a, mp, stream) : launch_bwd3w<C, GM, false, 4, 4, false, true>(a, mp, stream);
}
template <int C>
static int launch_dump_gm(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  switch (gm) {
    case GM_TRIPLANE: return launch_dump<C, GM_TRIPLANE>(a, mp, stream);
    case GM_VOXEL: return launch_dump<C, GM_VOXEL>(a, mp, stream);
    default: return launch_dump<C, GM_GENERIC>(a, mp, stream);
  }
}
int renderer_bwd_bf3_dump(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  return a.grid.channels == 16 ? launch_dump_gm<16>(a, mp, gm, stream) : launch_dump_gm<32>(a, mp, gm, stream);
}
const char* build_info_tuned_bwd_aux() { return "1"; }
#else
int renderer_bwd_bf3_dump(const LpRendererArgs&, const MfmaParams&, int, hipStream_t) {
  return set_error(LP_EUNSUPPORTED, "relu dump: this library was built without -DLP_TEST_HOOKS (no DUMP twins)");
}
const char* build_info_tuned_bwd_aux() { return "0"; }
#endif

}  // namespace lp
