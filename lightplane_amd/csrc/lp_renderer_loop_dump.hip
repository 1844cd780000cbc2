// lp_renderer_loop_dump.hip -- DUMP twins (lp_renderer_backward_relu_dump) of lp_renderer_loop.hip's backward instantiations (deep,
// wide-colour and two-block decoders, one wave per SIMD), compiled with the same flags as their production twins.
#include "lp_renderer_loop.h"

namespace lp {

int renderer_backward_loop_deep_dump(const LpRendererArgs& a, const LoopParams& p, int NB, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
#ifdef LP_TEST_HOOKS
  return loop_bwd_table_deep<true>(a, p, NB, nb, lds, tri, stream);
#else
  return set_error(LP_EUNSUPPORTED, "relu dump: this library was built without -DLP_TEST_HOOKS (no DUMP twins)");
#endif
}

}  // namespace lp
