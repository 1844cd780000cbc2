// lp_renderer_mfma_bwd_c32.hip -- the tuned Renderer backward (lp_renderer_mfma_bwd.h) for 32 grid channels.
#include "lp_renderer_mfma_bwd.h"

namespace lp {

int renderer_bwd_bf3_c32(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
#ifdef LP_DEV_ONE
  return set_error(LP_EUNSUPPORTED, "LP_DEV_ONE build: 16 channels only");
#else
  return launch_bwd_gm<32>(a, mp, gm, stream);
#endif
}

}  // namespace lp
