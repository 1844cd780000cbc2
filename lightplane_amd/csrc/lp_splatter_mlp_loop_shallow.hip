// lp_splatter_mlp_loop_shallow.hip -- the layer-looped MLP-Splatter backward for TWO-layer MLPs [E, H, Cout] (E, H in {16, 32};
// LightplaneMLPSplatter's default depth) at two waves per SIMD: splat_mlp_bwd_loop<E, CO, 1, ML = 2> keeps one hidden activation
// instead of three and is compiled with the spill-avoiding switches of build.py (FILE_FLAGS), like the Renderer's
// lp_renderer_loop_shallow.hip.
#include "lp_splatter_mlp_loop.h"

namespace lp {

int splatter_mlp_backward_loop_shallow(const LpSplatterArgs& a, hipStream_t stream) {
  const int E = a.mlp.dims[0], CO = a.mlp.dims[a.mlp.n_layers];
  if (E == 16) return CO == 16 ? sloop_launch(splat_mlp_bwd_loop<16, 16, 1, 2>, a, stream, true) : sloop_launch(splat_mlp_bwd_loop<16, 32, 1, 2>, a, stream, true);
  return CO == 16 ? sloop_launch(splat_mlp_bwd_loop<32, 16, 1, 2>, a, stream, true) : sloop_launch(splat_mlp_bwd_loop<32, 32, 1, 2>, a, stream, true);
}

}  // namespace lp
