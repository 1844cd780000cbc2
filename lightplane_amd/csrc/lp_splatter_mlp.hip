// lp_splatter_mlp.hip -- MLP-Splatter kernels: sample(input grid-list) + ray encoding -> MLP ->
// scatter-add into the output grid-list, and the backward of that chain.
//
// Replaces the reference's Triton kernels fw_kernel_wMLP / bw_kernel_wMLP
// (templates/splatter_fw.py:168-309, splatter_bw.py:183-394; launch sites
// lightplane_splatter.py:503-539, 664).  Semantics follow the naive reference
// (naive_splatter.py:185-289): the input grid-list is sampled with the Renderer's interpolation
// (F.grid_sample un-normalisation), the output is splatted with the Splatter's
// ((x+1)/2*size - 0.5), ReLU between the MLP layers and none after the last one; samples
// outside the cube contribute nothing when mask_out_of_bounds is set.  Features and unit weights
// are splatted in the same march (the reference launches twice).
//
// Shape-generic like lp_renderer_generic.hip: one lane = one ray, one wave = one workgroup,
// activations in a private array, weights through wave-uniform loads, weight gradients reduced
// over the 64 rays of the wave with the LDS-staged outer product of lp_generic_mlp.h.
#include "lp_generic_mlp.h"
#include "lp_host.h"

namespace lp {

struct SplatMlpPlan {
  int in;                    // [E] sampled input feature + encoding
  int out[LP_MAX_LAYERS];    // layer outputs (hidden: post ReLU, last: raw)
  int total;
  int stage_ld;              // LDS staging row stride (floats), bwd only
};

struct SplatMlpArgs {
  LpSplatterArgs a;
  SplatMlpPlan p;
};

// MLP(sample(input_grid, p) + enc): fills act[], returns the offset of the output vector
// Xs: the wave's LDS tile [64][stage_ld] -- wide layers then run for the whole wave on the fp32 matrix cores (dense_wave; the backward's
// recompute, wave-uniform control flow) -- or nullptr (the forward kernel, whose lanes leave the march individually).
LP_DEV int splat_mlp_forward(const SplatMlpArgs& sa, const Ray& ray, float x, float y, float z, const float* enc,
                             float* act, float* Xs = nullptr, int lane = 0) {
  const LpSplatterArgs& a = sa.a;
  const SplatMlpPlan& p = sa.p;
  const LpMlp& m = a.mlp;
  const int E = m.dims[0];
  sample_list(a.input_grid, ray.b, x, y, z, false, act + p.in);
  for (int c = 0; c < E; ++c) act[p.in + c] += enc[c];
  const float* cur = act + p.in;
  for (int l = 0; l < m.n_layers; ++l) {
    const bool last = (l == m.n_layers - 1);
    if (Xs && dense_on_mfma(m.dims[l], m.dims[l + 1]))
      dense_wave(mlp_w(a.mlp_params, m, l), mlp_b(a.mlp_params, m, l), m.dims[l], m.dims[l + 1], m.dims[l + 1], cur,
                 act + p.out[l], !last, Xs, p.stage_ld, lane);
    else
      dense(mlp_w(a.mlp_params, m, l), mlp_b(a.mlp_params, m, l), m.dims[l], m.dims[l + 1], m.dims[l + 1], cur,
            act + p.out[l], !last);
    cur = act + p.out[l];
  }
  return p.out[m.n_layers - 1];
}

template <int ACT_CAP>
__global__ void __launch_bounds__(64) splat_mlp_fwd_kernel(const SplatMlpArgs sa) {
  const LpSplatterArgs& a = sa.a;
  const int64_t ray_id = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (ray_id >= a.rays.n_rays) return;
  const Ray ray = load_ray(a.rays, ray_id);
  const int E = a.rays.encoding_dim;
  const int C = a.out.channels;
  float act[ACT_CAP];
  float enc[LP_MAX_WIDTH];
  for (int c = 0; c < E; ++c) enc[c] = a.rays.encoding[ray_id * E + c];
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  for (int s = 0; s < s_tot; ++s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    if (mask && !point_in_bounds(x, y, z)) continue;
    const float* v = act + splat_mlp_forward(sa, ray, x, y, z, enc, act);
    for (int g = 0; g < a.out.n_grids; ++g) {
      const Corners cs = grid_corners<true>(a.out.grids[g], ray.b, x, y, z);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (k < cs.n && cs.row[k] >= 0) {
          const float w = cs.w[k];
          float* dst = a.out_feature + cs.row[k] * C;
          for (int c = 0; c < C; ++c) atomic_add_f32(dst + c, w * v[c]);
          atomic_add_f32(a.out_weight + cs.row[k], w);
        }
      }
    }
  }
}

template <int ACT_CAP, bool LDS_ACC>
__global__ void __launch_bounds__(64) splat_mlp_bwd_kernel(const SplatMlpArgs sa) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const LpSplatterArgs& a = sa.a;
  const SplatMlpPlan& p = sa.p;
  const LpMlp& m = a.mlp;
  const int lane = threadIdx.x;
  const int64_t ray_id = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;

  float* Xs = lds;
  float* Ys = lds + 64 * p.stage_ld;
  float* gparams_lds = lds + 128 * p.stage_ld;
  float* gparams = nullptr;
  if (a.grad_mlp_params) {
    if (LDS_ACC) {
      for (int64_t i = lane; i < a.n_mlp_params; i += 64) gparams_lds[i] = 0.0f;
      gparams = gparams_lds;
    } else {
      gparams = a.grad_mlp_params;
    }
  }
  __syncthreads();

  float act[ACT_CAP];
  float enc[LP_MAX_WIDTH], denc[LP_MAX_WIDTH];
  float dy[LP_MAX_WIDTH], dx[LP_MAX_WIDTH];
  const Ray ray = load_ray(a.rays, rid);
  const int E = a.rays.encoding_dim;
  const int C = a.out.channels;
  for (int c = 0; c < E; ++c) {
    enc[c] = a.rays.encoding[rid * E + c];
    denc[c] = 0.0f;
  }
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  for (int s = 0; s < s_tot; ++s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    // every lane walks every sample (the weight-gradient reduction is a wave operation);
    // lanes without a contribution stage zeros
    const bool live = valid && !(mask && !point_in_bounds(x, y, z));
    splat_mlp_forward(sa, ray, x, y, z, enc, act, Xs, lane);
    // gradient w.r.t. the splatted vector: gather of grad_out / max(weight, 1e-5)
    for (int c = 0; c < C; ++c) dy[c] = 0.0f;
    if (live) {
      for (int g = 0; g < a.out.n_grids; ++g) {
        const Corners cs = grid_corners<true>(a.out.grids[g], ray.b, x, y, z);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          if (k < cs.n && cs.row[k] >= 0) {
            const float wn = cs.w[k] / fmaxf(a.weight[cs.row[k]], 1e-5f);
            const float* src = a.grad_out + cs.row[k] * C;
            for (int c = 0; c < C; ++c) dy[c] = fmaf(wn, src[c], dy[c]);
          }
        }
      }
    }
    if (LDS_ACC)
      mlp_backward<true>(a.mlp_params, p.stage_ld, m, C, p.in, p.out, act, dy, dx, gparams, Xs, Ys, lane, live);
    else
      mlp_backward<false>(a.mlp_params, p.stage_ld, m, C, p.in, p.out, act, dy, dx, gparams, Xs, Ys, lane, live);
    if (live)
      for (int c = 0; c < E; ++c) denc[c] += dx[c];
    if (a.grad_input_grid_list[0]) {  // whole rows per atomic instruction through the staging tiles (lp_generic_mlp.h)
      if (splat_wave_ok(p.stage_ld))
        splat_list_wave(a.input_grid, a.grad_input_grid_list, ray.b, x, y, z, false, dx, live, Xs, Ys, p.stage_ld, lane);
      else if (live)
        splat_list(a.input_grid, a.grad_input_grid_list, ray.b, x, y, z, false, dx);
    }
  }
  if (valid && a.grad_encoding)
    for (int c = 0; c < E; ++c) a.grad_encoding[ray_id * E + c] = denc[c];
  if (LDS_ACC && a.grad_mlp_params) {
    __syncthreads();
    for (int64_t i = lane; i < a.n_mlp_params; i += 64) {
      const float v = gparams_lds[i];
      if (v != 0.0f) atomic_add_f32(a.grad_mlp_params + i, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

static int make_plan(const LpSplatterArgs& a, SplatMlpPlan& p) {
  int pos = 0;
  p.in = pos; pos += a.mlp.dims[0];
  int maxw = a.mlp.dims[0];
  for (int l = 0; l < a.mlp.n_layers; ++l) {
    p.out[l] = pos; pos += a.mlp.dims[l + 1];
    maxw = a.mlp.dims[l + 1] > maxw ? a.mlp.dims[l + 1] : maxw;
  }
  p.total = pos;
  p.stage_ld = maxw + 1;
  return pos;
}

int splatter_mlp_forward_launch(const LpSplatterArgs& a, hipStream_t stream) {
  SplatMlpArgs sa;
  sa.a = a;
  const int total = make_plan(a, sa.p);
  const unsigned blocks = (unsigned)((a.rays.n_rays + 63) / 64);
  if (blocks == 0) return LP_OK;
  if (total <= 256)
    hipLaunchKernelGGL(splat_mlp_fwd_kernel<256>, dim3(blocks), dim3(64), 0, stream, sa);
  else if (total <= 1024)
    hipLaunchKernelGGL(splat_mlp_fwd_kernel<1024>, dim3(blocks), dim3(64), 0, stream, sa);
  else
    return set_error(LP_EUNSUPPORTED, "MLP splatter: sum of layer widths %d exceeds 1024", total);
  return check_launch("splat_mlp_fwd_kernel");
}

int splatter_mlp_backward_launch(const LpSplatterArgs& a, hipStream_t stream) {
  SplatMlpArgs sa;
  sa.a = a;
  const int total = make_plan(a, sa.p);
  const unsigned blocks = (unsigned)((a.rays.n_rays + 63) / 64);
  if (blocks == 0) return LP_OK;
  if (total > 1024) return set_error(LP_EUNSUPPORTED, "MLP splatter: sum of layer widths %d exceeds 1024", total);
  const size_t stage_bytes = (size_t)128 * sa.p.stage_ld * sizeof(float);
  const size_t param_bytes = (size_t)a.n_mlp_params * sizeof(float);
  const bool lds_acc = a.grad_mlp_params && (stage_bytes + param_bytes <= 96 * 1024);
  const size_t lds = stage_bytes + (lds_acc ? param_bytes : 0);
#define LP_LAUNCH_SBWD(CAP, ACC)                                                                    \
  do {                                                                                              \
    hipError_t e = hipFuncSetAttribute((const void*)splat_mlp_bwd_kernel<CAP, ACC>,                 \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);       \
    if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); \
    hipLaunchKernelGGL((splat_mlp_bwd_kernel<CAP, ACC>), dim3(blocks), dim3(64), lds, stream, sa);  \
  } while (0)
  if (total <= 256) {
    if (lds_acc) LP_LAUNCH_SBWD(256, true); else LP_LAUNCH_SBWD(256, false);
  } else {
    if (lds_acc) LP_LAUNCH_SBWD(1024, true); else LP_LAUNCH_SBWD(1024, false);
  }
#undef LP_LAUNCH_SBWD
  return check_launch("splat_mlp_bwd_kernel");
}

}  // namespace lp
