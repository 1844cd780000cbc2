// lp_renderer_mfma_bwd.hip -- host dispatch of the tuned Renderer backward + its instantiations for 16 grid channels.
// Kernel template: lp_renderer_mfma_bwd.h; 32 channels: lp_renderer_mfma_bwd_c32.hip; LP_ARITH_FP32: lp_renderer_mfma_bwd_aux.hip;
// DUMP twins: lp_renderer_mfma_bwd_dump.hip.
#include "lp_renderer_mfma_bwd.h"

namespace lp {

int renderer_bwd_bf3_c16(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  return launch_bwd_gm<16>(a, mp, gm, stream);
}

int renderer_backward_mfma2(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  int rc;
#ifdef LP_DEV_ONE
  rc = renderer_bwd_bf3_c16(a, mp, gm, stream);
#else
  g_last_backward = "tuned family, rays per wavefront";
  if (renderer_bwd_tm_supported(a)) {
    g_last_backward = "tuned family, samples per wavefront (transposed march)";
    rc = renderer_bwd_bf3_tm_launch(a, mp, gm, stream);
  }
  else if (mp.relu_dump) rc = renderer_bwd_bf3_dump(a, mp, gm, stream);                    // test hook
  else if (a.arithmetic == LP_ARITH_FP32) rc = renderer_bwd_bf3_f32(a, mp, gm, stream);     // the reference's arithmetic, per call
  else if (a.grid.channels == 16) rc = renderer_bwd_bf3_c16(a, mp, gm, stream);
  else rc = renderer_bwd_bf3_c32(a, mp, gm, stream);
#endif
  if (rc) return rc;
  return check_launch("renderer_bwd_mfma2");
}

// what the tuned backward of this binary computes in (lp_build_info)
const char* build_info_tuned_bwd() {
#define LP_STR2(x) #x
#define LP_STR(x) LP_STR2(x)
  return "{\"dx_limbs\": " LP_STR(LP_DX_LIMBS) ", \"dw\": "
#if LP_DW_BF16
         "\"two-limb bf16 operands, v_mfma_f32_16x16x32_bf16, three limb products\""
#else
         "\"fp32 operands, v_mfma_f32_16x16x4_f32\""
#endif
         ", \"recompute\": \"bf16x3, v_mfma_f32_32x32x16_bf16, six limb products\", \"arith_fp32\": \"dx_limbs 3, dw fp32 "
         "v_mfma_f32_16x16x4_f32 (four-wave workgroups, <= 64 beyond-far samples)\"}";
}

#ifdef LP_PHASE_TIMING
int debug_phase_cycles(unsigned long long* out) {
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) return set_error((int)e, "sync: %s", hipGetErrorString(e));
  e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * 16);
  if (e != hipSuccess) return set_error((int)e, "from symbol: %s", hipGetErrorString(e));
  unsigned long long z[16] = {0};
  e = hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z));
  if (e != hipSuccess) return set_error((int)e, "to symbol: %s", hipGetErrorString(e));
  unsigned long long c32[16];  // (g_phase exists once per translation unit: add the 32-channel unit's totals)
  if (debug_phase_cycles_c32(c32) == 0)
    for (int i = 0; i < 16; ++i) out[i] += c32[i];
  return 0;
}
#else
int debug_phase_cycles(unsigned long long*) { return -1; }
#endif

}  // namespace lp
