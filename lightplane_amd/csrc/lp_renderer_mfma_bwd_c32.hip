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

#ifdef LP_PHASE_TIMING
int debug_phase_cycles_c32(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {0};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#else
int debug_phase_cycles_c32(unsigned long long*) { return -1; }
#endif

}  // namespace lp
