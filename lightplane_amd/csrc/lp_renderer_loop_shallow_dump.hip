// lp_renderer_loop_shallow_dump.hip -- DUMP twins (lp_renderer_backward_relu_dump) of lp_renderer_loop_shallow.hip's backward
// instantiations, compiled with the same per-file flags (build.py FILE_FLAGS) so that a twin is its production kernel + stores.
#include "lp_renderer_loop.h"

namespace lp {

int renderer_backward_loop_shallow_dump(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
#ifdef LP_TEST_HOOKS
  return loop_bwd_table_shallow<true>(a, p, nb, lds, tri, stream);
#else
  return set_error(LP_EUNSUPPORTED, "relu dump: this library was built without -DLP_TEST_HOOKS (no DUMP twins)");
#endif
}

}  // namespace lp
