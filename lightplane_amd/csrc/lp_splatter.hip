// lp_splatter.hip -- Splatter kernels (rays -> 3D grid-list scatter-add and its backward).
//
// Replaces the reference's Triton splatter kernels (templates/splatter_fw.py:71-165,
// splatter_bw.py:75-180) and the host-side normalisation (lightplane_splatter.py:541,584).
//
// Lane mapping: a group of LPR (power of two, <= 64) adjacent lanes owns one ray; lane `sub`
// of the group owns channels sub, sub+LPR, ...  All lanes of a group walk the same samples
// (geometry is recomputed per lane -- a handful of VALU ops) so that every atomic / gather
// instruction of the group touches ONE contiguous C*4-byte row: fully coalesced 128-byte
// segments for C=32 instead of 64 scattered dwords.  Features and unit weights are splatted
// in the same march (the reference launches the kernel twice).
#include "lp_device.h"
#include "lp_host.h"

namespace lp {

template <int LPR, int CPL>
__global__ void __launch_bounds__(256) splat_fwd_kernel(const LpSplatterArgs a) {
  const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ray_id = gtid / LPR;
  const int sub = (int)(gtid % LPR);
  if (ray_id >= a.rays.n_rays) return;
  const int C = a.out.channels;
  const Ray ray = load_ray(a.rays, ray_id);
  float e[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = sub + j * LPR;
    e[j] = (c < C) ? a.rays.encoding[ray_id * C + c] : 0.0f;
  }
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  for (int s = 0; s < s_tot; ++s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    if (mask && !point_in_bounds(x, y, z)) continue;
    for (int g = 0; g < a.out.n_grids; ++g) {
      const Corners cs = grid_corners<true>(a.out.grids[g], ray.b, x, y, z);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < cs.n && cs.row[k] >= 0) {
          const float w = cs.w[k];
          float* dst = a.out_feature + cs.row[k] * C;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            const int c = sub + j * LPR;
            if (c < C) atomic_add_f32(dst + c, w * e[j]);
          }
          if (sub == 0) atomic_add_f32(a.out_weight + cs.row[k], w);
        }
      }
    }
  }
}

template <int LPR, int CPL>
__global__ void __launch_bounds__(256) splat_bwd_kernel(const LpSplatterArgs a) {
  const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t ray_id = gtid / LPR;
  const int sub = (int)(gtid % LPR);
  if (ray_id >= a.rays.n_rays) return;
  const int C = a.out.channels;
  const Ray ray = load_ray(a.rays, ray_id);
  float acc[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) acc[j] = 0.0f;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  for (int s = 0; s < s_tot; ++s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    if (mask && !point_in_bounds(x, y, z)) continue;
    for (int g = 0; g < a.out.n_grids; ++g) {
      const Corners cs = grid_corners<true>(a.out.grids[g], ray.b, x, y, z);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < cs.n && cs.row[k] >= 0) {
          // gradient of out = feat / max(weight, 1e-5) w.r.t. feat (weights carry no gradient)
          const float wn = cs.w[k] / fmaxf(a.weight[cs.row[k]], 1e-5f);
          const float* src = a.grad_out + cs.row[k] * C;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            const int c = sub + j * LPR;
            if (c < C) acc[j] = fmaf(wn, src[c], acc[j]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
    const int c = sub + j * LPR;
    if (c < C) a.grad_encoding[ray_id * C + c] = acc[j];
  }
}

__global__ void __launch_bounds__(256) splat_normalize_kernel(float* feature, const float* weight,
                                                              int64_t n_rows, int C) {
  const int64_t n = n_rows * C;
  const int64_t stride = (int64_t)gridDim.x * 256;
  if ((C & 3) == 0) {
    const int64_t n4 = n / 4;
    const int c4 = C / 4;
    float4* f4 = reinterpret_cast<float4*>(feature);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
      float4 v = f4[i];
      // true division (matches torch's feat / clamp(weight)) -- the pass is HBM bound anyway
      const float d = fmaxf(weight[i / c4], 1e-5f);
      v.x = v.x / d; v.y = v.y / d; v.z = v.z / d; v.w = v.w / d;
      f4[i] = v;
    }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
      feature[i] = feature[i] / fmaxf(weight[i / C], 1e-5f);
  }
}

__global__ void __launch_bounds__(256) hash_randn_kernel(const int32_t* x1, const int32_t* x2, float* out,
                                                         int64_t n, int32_t seed) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = hash_randn(x1[i], x2[i], seed);
}

// ---------------------------------------------------------------------------------------

#define LP_SPLAT_DISPATCH(KERNEL)                                                              \
  do {                                                                                         \
    const int C = a.out.channels;                                                              \
    int lpr = 1;                                                                               \
    while (lpr < C && lpr < 64) lpr <<= 1;                                                     \
    const int cpl = (C + lpr - 1) / lpr;                                                       \
    const int64_t threads = a.rays.n_rays * lpr;                                               \
    const unsigned blocks = (unsigned)((threads + 255) / 256);                                 \
    if (blocks == 0) return LP_OK;                                                             \
    if (cpl > 2) return set_error(LP_EUNSUPPORTED, "splatter: %d channels > 128", C);          \
    switch (lpr) {                                                                             \
      case 1: hipLaunchKernelGGL((KERNEL<1, 1>), dim3(blocks), dim3(256), 0, stream, a); break;   \
      case 2: hipLaunchKernelGGL((KERNEL<2, 1>), dim3(blocks), dim3(256), 0, stream, a); break;   \
      case 4: hipLaunchKernelGGL((KERNEL<4, 1>), dim3(blocks), dim3(256), 0, stream, a); break;   \
      case 8: hipLaunchKernelGGL((KERNEL<8, 1>), dim3(blocks), dim3(256), 0, stream, a); break;   \
      case 16: hipLaunchKernelGGL((KERNEL<16, 1>), dim3(blocks), dim3(256), 0, stream, a); break; \
      case 32: hipLaunchKernelGGL((KERNEL<32, 1>), dim3(blocks), dim3(256), 0, stream, a); break; \
      default:                                                                                 \
        if (cpl == 1) hipLaunchKernelGGL((KERNEL<64, 1>), dim3(blocks), dim3(256), 0, stream, a); \
        else hipLaunchKernelGGL((KERNEL<64, 2>), dim3(blocks), dim3(256), 0, stream, a);          \
    }                                                                                          \
  } while (0)

int splatter_forward_launch(const LpSplatterArgs& a, hipStream_t stream) {
  LP_SPLAT_DISPATCH(splat_fwd_kernel);
  return check_launch("splat_fwd_kernel");
}

int splatter_backward_launch(const LpSplatterArgs& a, hipStream_t stream) {
  LP_SPLAT_DISPATCH(splat_bwd_kernel);
  return check_launch("splat_bwd_kernel");
}

int splatter_normalize_launch(float* feature, const float* weight, int64_t n_rows, int channels,
                              hipStream_t stream) {
  const int64_t n = n_rows * channels;
  if (n == 0) return LP_OK;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(splat_normalize_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, feature, weight,
                     n_rows, channels);
  return check_launch("splat_normalize_kernel");
}

int hash_randn_launch(const int32_t* x1, const int32_t* x2, float* out, int64_t n, int32_t seed,
                      hipStream_t stream) {
  if (n == 0) return LP_OK;
  hipLaunchKernelGGL(hash_randn_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x1, x2, out, n,
                     seed);
  return check_launch("hash_randn_kernel");
}

}  // namespace lp
