// lp_renderer_mfma.hip -- MFMA Renderer kernels (placeholder until the kernels land).
#include "lp_device.h"
#include "lp_host.h"

namespace lp {
bool renderer_mfma_supported(const LpRendererArgs& a, const char** why) {
  (void)a;
  *why = "MFMA kernels not built yet";
  return false;
}
int renderer_forward_mfma(const LpRendererArgs&, hipStream_t) { return set_error(LP_EUNSUPPORTED, "no MFMA kernel"); }
int renderer_backward_mfma(const LpRendererArgs&, hipStream_t) { return set_error(LP_EUNSUPPORTED, "no MFMA kernel"); }
}  // namespace lp
