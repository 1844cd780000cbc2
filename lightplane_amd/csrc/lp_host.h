// lp_host.h -- host-side glue shared by the translation units of liblightplane_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>

#include "../../include/lightplane_hip.h"

namespace lp {

// Records a formatted message for lp_last_error() (thread-local) and returns `code`.
int set_error(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
// hipGetLastError() -> LP_OK or the positive hipError_t (message recorded).
int check_launch(const char* what);

// test hook (lp_renderer_backward_relu_dump, lp_api.hip): while non-NULL, the MFMA backwards launch their DUMP twins, which also
// write the ReLU decisions of the recompute here.  Thread-local; NULL in every product call.
extern thread_local uint32_t* g_relu_dump;
// developer / test hook (lp_debug_last_renderer_backward, lp_api.hip): which Renderer backward the PROCESS launched last (autograd runs
// the backward on its own thread: a thread-local would be invisible to the caller; a plain pointer to a static string -- a race between
// two launching threads only makes the answer one of the two)
extern const char* volatile g_last_backward;

// The MFMA kernels index grid rows with 32-bit integers and address them as a 64-bit base + row * C * 4 (+ a 32-bit byte offset for
// the corner inside the cell): any grid-list whose every grid ends below row 2^31 of the tensor that holds it, and whose z-slice
// (H x W rows: the farthest corner of a cell) stays below 2 GB.  No limit on the byte size -- a 256^3 x 32 grid per batch element
// is 2.1 GB, and batches of them are what 288 GB of HBM are for.  (Rounds 2-6 excluded grid-lists of 4 GB or more: the per-slot
// scatter walk formed a 32-bit byte offset from the tensor's base.)
inline bool grid_list_rows_ok(const LpGridList& gl) {
  if (gl.n_rows >= ((int64_t)1 << 31)) return false;
  for (int g = 0; g < gl.n_grids; ++g) {
    const LpGrid& d = gl.grids[g];
    if (d.row_offset + (int64_t)d.B * d.D * d.H * d.W >= ((int64_t)1 << 31)) return false;
    if (((int64_t)d.H * d.W + d.W + 1) * gl.channels * 4 >= ((int64_t)1 << 31)) return false;
  }
  return true;
}

// LP_SEG_LEN-sample blocks per segment of a segment-parallel launch (small batches: forward and backward of both MFMA families).
// Every workgroup pays the weight staging and, in the backward, the dW flush once, and workgroups that share a CU share its SIMDs:
//  * as many 16-sample segments (two blocks) as keep the launch within ONE round of resident workgroups -- measured
//    (scripts/bench_small_batch.py, S = 128, tuned backward): 16-sample segments 4 096 rays 0.39 ms / 16 384 rays 1.13 ms, 32-sample
//    segments 0.51 / 0.88 ms;
//  * 8-sample segments (one block, ABI 0.2.7; LP_SEG_LEN was 16 before) only while even they leave at most one workgroup per CU: the
//    reference example's decoder (2/2/2 x 64) on 1 024 random rays 0.89 -> 0.56 ms per training step, 2/2/2 x 32 0.50 -> 0.35 ms; on
//    4 096 rays, where 16-sample segments already give every CU a workgroup, 8-sample ones were 18 % SLOWER (0.73 -> 0.86 ms:
//    profiles/r06_train_step.txt).
// LP_SEG_BLOCKS (developer knob) forces the number of blocks.
inline int seg_blocks_for(unsigned ray_blocks, int n_rec, unsigned resident) {
  static const int forced = getenv("LP_SEG_BLOCKS") ? atoi(getenv("LP_SEG_BLOCKS")) : 0;
  if (forced > 0) return forced < n_rec ? forced : n_rec;
  if (n_rec <= 1 || (uint64_t)ray_blocks * (uint64_t)n_rec <= 256u) return 1;
  int m = 2;
  while (m < n_rec && (uint64_t)ray_blocks * (uint64_t)((n_rec + m - 1) / m) > resident) m += 2;
  return m < n_rec ? m : n_rec;
}

// generic (shape-agnostic) kernels: lp_renderer_generic.hip
int renderer_forward_generic(const LpRendererArgs& a, hipStream_t stream);
int renderer_backward_generic(const LpRendererArgs& a, hipStream_t stream);
int renderer_corner_rows_launch(const LpRendererArgs& a, int64_t* rows, hipStream_t stream);

// MFMA kernels: lp_renderer_mfma.hip
bool renderer_mfma_supported(const LpRendererArgs& a, const char** why);
int renderer_forward_mfma(const LpRendererArgs& a, hipStream_t stream);
int renderer_backward_mfma(const LpRendererArgs& a, hipStream_t stream);
int renderer_mfma_segments(const LpRendererArgs& a);  // segments of the segment-parallel backward (1 = none)
bool renderer_tm_eligible(const LpRendererArgs& a);                        // these arguments may march samples per wavefront (forward and backward)
int renderer_tm_rays_per_wave(const LpRendererArgs& a, int resident_workgroups);  // rays per wave of that march (1 .. 32)
bool renderer_mfma_f32_supported(const LpRendererArgs& a);  // the tuned family has an LP_ARITH_FP32 instantiation for these arguments
int renderer_forward_combine_launch(const LpRendererArgs& a, int seg_blocks, hipStream_t stream);  // chains the segments of a segmented forward

// layer-looped bf16x3 MFMA family (1-4 layers per MLP, hidden 16 / 32 / 64): lp_renderer_loop.hip
bool renderer_loop_supported(const LpRendererArgs& a, const char** why);
bool renderer_loop_fits(const LpRendererArgs& a);  // its weight images + tiles fit the 160 KB LDS
int renderer_loop_segments(const LpRendererArgs& a);  // segments of the segment-parallel march (1 = none)
int renderer_forward_loop(const LpRendererArgs& a, hipStream_t stream);
int renderer_backward_loop(const LpRendererArgs& a, hipStream_t stream);
int renderer_loop_dump_words(const LpRendererArgs& a);  // words per (ray, sample) of the ReLU dump
int renderer_generic_dump_words(const LpRendererArgs& a);  // ... of the shape-generic backward (sites x ceil(widest site / 32) + 1)
// per-translation-unit arithmetic reports for lp_build_info() (JSON fragments, static storage)
const char* build_info_tuned_bwd();
const char* build_info_tuned_bwd_aux();
const char* build_info_loop_deep();
const char* build_info_loop_shallow();
const char* build_info_splatter_mlp();

// splatter: lp_splatter.hip
int splatter_forward_launch(const LpSplatterArgs& a, hipStream_t stream);
int splatter_backward_launch(const LpSplatterArgs& a, hipStream_t stream);
// MLP-Splatter: lp_splatter_mlp.hip
int splatter_mlp_forward_launch(const LpSplatterArgs& a, hipStream_t stream);
int splatter_mlp_backward_launch(const LpSplatterArgs& a, hipStream_t stream);
// MLP-Splatter, layer-looped bf16x3 family (2-4 layers, widths 16 / 32 / 64): lp_splatter_mlp_loop.hip
bool splatter_mlp_loop_supported(const LpSplatterArgs& a);
int splatter_mlp_forward_loop(const LpSplatterArgs& a, hipStream_t stream);
int splatter_mlp_backward_loop(const LpSplatterArgs& a, hipStream_t stream);
int splatter_normalize_launch(float* feature, const float* weight, int64_t n_rows, int channels,
                              hipStream_t stream);
// ray-direction embedding of the module front-end: lp_ray_embedding.hip
int ray_embedding_forward_launch(const LpRayEmbedArgs& a, hipStream_t stream);
int ray_embedding_backward_launch(const LpRayEmbedArgs& a, hipStream_t stream);
int hash_randn_launch(const int32_t* x1, const int32_t* x2, float* out, int64_t n, int32_t seed,
                      hipStream_t stream);

}  // namespace lp
