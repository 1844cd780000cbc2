// lp_ray_embedding.hip -- the ray-direction embedding of the module front-end as ONE kernel per direction.
//
// Replaces the PyTorch op chain of the reference's LightplaneRenderer._get_ray_embedding
// (lightplane/renderer_module.py:578-601): F.normalize -> calc_harmonic_embedding (ray_utils.py:181-212: mul, add,
// sin, flatten, cat) -> torch.nn.Linear, eight small launches forward and about as many in autograd's backward --
// which doubles the step time of a small-batch training step.
//
//   d   = directions / max(|directions|_2, 1e-12)
//   emb = [ sin(d_c 2^k + p pi/2) ]_{p<2, c<3, k<n}  ++  d          index (p*3 + c)*n + k ; 6 n + 3 values
//   out = emb @ weight^T + bias                                      weight [E, 6 n + 3] (torch.nn.Linear)
//
// Forward: one lane = one ray; the weights sit in LDS (read as wave-uniform broadcasts), a ray's E outputs leave as
// float4 rows.  Backward: a workgroup stages the embeddings and the upstream gradients of its 256 rays in LDS,
// thread t owns the entries t, t + 256, ... of grad_weight and sums them over the rays, one atomic per entry and
// workgroup (bias: the first E threads).  The ray directions get no gradient (the reference treats ray geometry as
// non-differentiable everywhere else, lightplane_renderer.py:724-756).
#include "lp_device.h"
#include "lp_host.h"

namespace lp {

constexpr int RE_THREADS = 256;
constexpr int RE_MAX_IN = 3 + 6 * 10;  // n_harmonics <= 10

LP_DEV int embed_ray(const float* directions, int64_t ray, int n, float* emb) {
  float d[3] = {directions[3 * ray + 0], directions[3 * ray + 1], directions[3 * ray + 2]};
  const float nrm = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  const float inv = fmaxf(nrm, 1e-12f);
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = d[c] / inv;
  const float half_pi = 1.57079632679489661923f;
  for (int p = 0; p < 2; ++p)
    for (int c = 0; c < 3; ++c) {
      float f = 1.0f;
      for (int k = 0; k < n; ++k) {
        emb[(p * 3 + c) * n + k] = sinf(d[c] * f + (p ? half_pi : 0.0f));
        f *= 2.0f;
      }
    }
  emb[6 * n + 0] = d[0];
  emb[6 * n + 1] = d[1];
  emb[6 * n + 2] = d[2];
  return 6 * n + 3;
}

__global__ void __launch_bounds__(RE_THREADS) ray_embedding_fwd(const LpRayEmbedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = 6 * a.n_harmonics + 3, E = a.out_dim;
  float* w = lds;          // [E][D]
  float* b = lds + E * D;  // [E]
  for (int i = threadIdx.x; i < E * D; i += RE_THREADS) w[i] = a.weight[i];
  for (int i = threadIdx.x; i < E; i += RE_THREADS) b[i] = a.bias[i];
  __syncthreads();
  const int64_t ray = (int64_t)blockIdx.x * RE_THREADS + threadIdx.x;
  if (ray >= a.n_rays) return;
  float emb[RE_MAX_IN];
  embed_ray(a.directions, ray, a.n_harmonics, emb);
  float* out = a.out + ray * E;
  for (int e0 = 0; e0 < E; e0 += 4) {
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j;
      float s = (e < E) ? b[e] : 0.0f;
      if (e < E) {
        const float* we = w + e * D;
        for (int i = 0; i < D; ++i) s = fmaf(emb[i], we[i], s);
      }
      acc[j] = s;
    }
    if (e0 + 4 <= E && (E & 3) == 0) {
      *reinterpret_cast<float4*>(out + e0) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
      for (int j = 0; j < 4 && e0 + j < E; ++j) out[e0 + j] = acc[j];
    }
  }
}

__global__ void __launch_bounds__(RE_THREADS) ray_embedding_bwd(const LpRayEmbedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int D = 6 * a.n_harmonics + 3, E = a.out_dim;
  const int DL = D + 1, EL = E + 1;  // padded rows: the column walks below are conflict-free
  float* emb_s = lds;                    // [256][DL]
  float* g_s = lds + RE_THREADS * DL;    // [256][EL]
  const int64_t ray0 = (int64_t)blockIdx.x * RE_THREADS;
  const int64_t ray = ray0 + threadIdx.x;
  const bool valid = ray < a.n_rays;
  {
    float emb[RE_MAX_IN];
    if (valid) embed_ray(a.directions, ray, a.n_harmonics, emb);
    for (int i = 0; i < D; ++i) emb_s[threadIdx.x * DL + i] = valid ? emb[i] : 0.0f;
    for (int e = 0; e < E; ++e) g_s[threadIdx.x * EL + e] = valid ? a.grad_out[ray * E + e] : 0.0f;
  }
  __syncthreads();
  if (a.grad_weight) {
    for (int idx = threadIdx.x; idx < E * D; idx += RE_THREADS) {
      const int e = idx / D, i = idx - e * D;
      float s = 0.0f;
      for (int r = 0; r < RE_THREADS; ++r) s = fmaf(g_s[r * EL + e], emb_s[r * DL + i], s);
      atomic_add_f32(a.grad_weight + idx, s);
    }
  }
  if (a.grad_bias) {
    for (int e = threadIdx.x; e < E; e += RE_THREADS) {
      float s = 0.0f;
      for (int r = 0; r < RE_THREADS; ++r) s += g_s[r * EL + e];
      atomic_add_f32(a.grad_bias + e, s);
    }
  }
}

static unsigned re_blocks(const LpRayEmbedArgs& a) { return (unsigned)((a.n_rays + RE_THREADS - 1) / RE_THREADS); }

int ray_embedding_forward_launch(const LpRayEmbedArgs& a, hipStream_t stream) {
  if (a.n_rays == 0) return LP_OK;
  const int D = 6 * a.n_harmonics + 3;
  const size_t lds = (size_t)(a.out_dim * D + a.out_dim) * sizeof(float);
  hipLaunchKernelGGL(ray_embedding_fwd, dim3(re_blocks(a)), dim3(RE_THREADS), lds, stream, a);
  return check_launch("ray_embedding_fwd");
}

int ray_embedding_backward_launch(const LpRayEmbedArgs& a, hipStream_t stream) {
  if (a.n_rays == 0) return LP_OK;
  const int D = 6 * a.n_harmonics + 3;
  const size_t lds = (size_t)RE_THREADS * (D + 1 + a.out_dim + 1) * sizeof(float);
  if (lds > 150 * 1024)
    return set_error(LP_EUNSUPPORTED, "ray embedding backward: %d harmonics x %d outputs need %zu bytes of LDS", a.n_harmonics,
                     a.out_dim, lds);
  const hipError_t e = hipFuncSetAttribute((const void*)ray_embedding_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  hipLaunchKernelGGL(ray_embedding_bwd, dim3(re_blocks(a)), dim3(RE_THREADS), lds, stream, a);
  return check_launch("ray_embedding_bwd");
}

}  // namespace lp
