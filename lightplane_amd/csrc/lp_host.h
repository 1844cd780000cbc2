// lp_host.h -- host-side glue shared by the translation units of liblightplane_hip.so.
#pragma once
#include <hip/hip_runtime.h>

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
