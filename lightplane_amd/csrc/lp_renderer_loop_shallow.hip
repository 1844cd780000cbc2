// lp_renderer_loop_shallow.hip -- the layer-looped Renderer backward for SHALLOW decoders at two waves per SIMD.
//
// renderer_bwd_loop<C, 1, TG, MT, MH> (lp_renderer_loop.h) keeps every hidden activation of the recompute in registers: with the
// family's maxima (4 trunk + 3 + 3 head layers) that is 160 registers next to the dW quadrants of ten layers, so those
// instantiations run one wave per SIMD (lp_renderer_loop.hip).  A decoder of at most 2 trunk layers and one hidden layer per head
// -- the reference's notebook shapes 1/1/1 x 16 and 2/1/1 x 32, its example's 1/1/2, the default 2/2/2, the two-grid decoder's
// 0/2/2 -- needs 64, and the instantiations of this file (MT = 2 or 1, MH = 1, __launch_bounds__(256, 2)) fit 256 registers
// WITHOUT spills once MachineLICM is off and the allocator may re-materialise instead of spilling (build.py FILE_FLAGS: 252 / 246
// VGPRs, no spilled register; with the default switches 49 / 27 spilled).  LDS: four block images + tiles = 71 KB, two
// workgroups per CU.  The second resident wave is worth 1.6x: 256^2 rays x 128 samples, 2/2/2 x 32 through this family
// (LP_LOOP=1) backward 3.85 -> 2.42 ms, 1/1/1 x 16 forward + backward 2.75 -> 1.87 ms (the fp32-MFMA flex kernels it replaces:
// 2.05 ms) -- profiles/r04_loop_shallow_ab.txt.  The same switches cost the deep one-wave instantiations 1-2 %, hence two
// translation units.
#include "lp_renderer_loop.h"

namespace lp {

int renderer_backward_loop_shallow(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
  return loop_bwd_table_shallow<false>(a, p, nb, lds, tri, stream);
}

// what this translation unit's backward computes in (lp_build_info)
const char* build_info_loop_shallow() {
#define LP_STR2(x) #x
#define LP_STR(x) LP_STR2(x)
  return "{\"dx_limbs\": " LP_STR(LP_DX_LIMBS) ", \"dw\": "
#if LP_LOOP_DW_BF16
         "\"two-limb bf16 operands, v_mfma_f32_16x16x32_bf16\"}";
#else
         "\"fp32 operands, v_mfma_f32_16x16x4_f32\"}";
#endif
}

}  // namespace lp
