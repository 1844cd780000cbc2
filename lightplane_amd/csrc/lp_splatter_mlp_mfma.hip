// lp_splatter_mlp_mfma.hip -- MLP-Splatter on the matrix cores (gfx950).
//
// Shape family: two-layer MLP [E, 32, Cout] with E = input-grid channels = encoding width in
// {16, 32} and Cout in {16, 32} -- LightplaneMLPSplatter's default shape.  Everything else runs on
// the shape-generic kernels of lp_splatter_mlp.hip.
//
// Forward, per sample of 32 rays (one wave, the lane <-> (ray, feature) mapping of the Renderer):
// gather the input grid-list (Renderer interpolation) + ray encoding -> two MFMA layers, register
// to register -> the output vector is transposed through LDS to [channel][ray] and splatted with the
// run-merged walk of the Splatter (one row-contiguous atomic per run and slot, unit weights batched
// eight slots per instruction).
// Backward: recompute the two layers, gather d v from grad_out / clamp(weight) at the output taps,
// dX chains on MFMA, weight gradients shared by the workgroup as 16x16 quadrants
// (v_mfma_f32_16x16x4_f32 over the X / dY tiles of all four waves, see lp_renderer_mfma_bwd.hip),
// input-grid gradient through the Renderer's run-merged scatter.
#include "lp_mfma_common.h"
#include "lp_splat_walk.h"

namespace lp {

typedef float f32x4m __attribute__((ext_vector_type(4)));
#define LP_MFMA16M(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TM_LD = 36;  // row stride of the feature-major tiles

struct MlpSplatParams {
  int64_t w1, w2, b1, b2;  // float offsets inside mlp_params
  int dbg;
  int n_seg;  // small batches: segments the march is cut into (blockIdx = (128 rays, segment); samples are independent)
};

struct LdsS {
  static constexpr int W1 = 0;                 // [32][33], rows >= E zero
  static constexpr int W2 = W1 + 32 * W_LD;    // [32][33], columns >= Cout zero
  static constexpr int B1 = W2 + 32 * W_LD;    // [32]
  static constexpr int B2 = B1 + 32;           // [32], entries >= Cout zero
  static constexpr int INF = B2 + 32;          // [MAX_INF]
  static constexpr int END = INF + MAX_INF;
  // per wave: X tile (fwd: transposed output vector; bwd: X tile / dx tile), dY tile (+ walk weight table)
  static constexpr int XT = 0;
  static constexpr int YT = 32 * TM_LD;
  static constexpr int PER_WAVE = 2 * 32 * TM_LD;
  static constexpr int TOTAL = END + WAVES * PER_WAVE;
};

template <int E, int CO>
LP_DEV void stage_mlp(const LpSplatterArgs& a, const MlpSplatParams& mp, float* lds) {
  const float* P = a.mlp_params;
  const int tid = threadIdx.x;
  for (int i = tid; i < 32 * 32; i += 256) {
    const int row = i >> 5, col = i & 31;
    lds[LdsS::W1 + row * W_LD + col] = (row < E) ? P[mp.w1 + row * 32 + col] : 0.0f;
    lds[LdsS::W2 + row * W_LD + col] = (col < CO) ? P[mp.w2 + row * CO + col] : 0.0f;
  }
  for (int i = tid; i < 32; i += 256) {
    lds[LdsS::B1 + i] = P[mp.b1 + i];
    lds[LdsS::B2 + i] = (i < CO) ? P[mp.b2 + i] : 0.0f;
  }
  for (int i = tid; i < MAX_INF; i += 256)
    lds[LdsS::INF + i] = (i < a.march.num_samples_inf) ? inf_scale(i, a.march) : 0.0f;
}

LP_DEV f32x16 bias16(const float* b /* vector + 4h */) {
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(b + 8 * j);
    acc[4 * j + 0] = v.x; acc[4 * j + 1] = v.y; acc[4 * j + 2] = v.z; acc[4 * j + 3] = v.w;
  }
  return acc;
}

// Splatter-side walk of one output grid: the vector to splat sits in LDS as [channel][ray] (tile `vT`);
// features per run and slot, then the unit weights eight slots at a time.
template <int C>
LP_DEV void splat_walk_lds(float* feat, float* wgt, const LpGrid& g, int b, float x, float y, float z, bool live,
                           int lane, const float* vT, float* wT) {
  constexpr int CPL = C / 16;  // channels per lane: 16 lanes per tap slot, four slots per pass
  const int h = lane >> 5, r = lane & 31, sub = lane & 15, grp = lane >> 4;
  TapSet tp;
  grid_tapset<true>(g, b, x, y, z, tp);
  if (!live) {
    tp.ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tp.w[k] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) wT[(4 * h + i) * 32 + r] = h ? tp.w[4 + i] : tp.w[i];
  const int row0 = tp.row0;
  const int ok = (int)tp.ok;
  const int prow_ = lane_prev(row0), pok_ = lane_prev(ok);  // all lanes enabled: see run_head()
  const bool head = run_head(r, row0, prow_, ok, pok_);
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(head));
  const bool voxel = g.D > 1 && g.H > 1 && g.W > 1;
  const int n_pass = voxel ? 2 : 1;
  for (int p = 0; p < n_pass; ++p) {
    const int k = p * 4 + grp;
    const int koff = (k & 1) * tp.su + ((k >> 1) & 1) * tp.sv + (k >> 2) * tp.st;
    const unsigned kbit = 1u << k;
    const float4* wsrc = reinterpret_cast<const float4*>(wT + k * 32);
    float run[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      float dx[CPL][8];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float4* dsrc = reinterpret_cast<const float4*>(vT + (sub + 16 * j) * TM_LD);
        const float4 d0 = dsrc[2 * c8], d1 = dsrc[2 * c8 + 1];
        dx[j][0] = d0.x; dx[j][1] = d0.y; dx[j][2] = d0.z; dx[j][3] = d0.w;
        dx[j][4] = d1.x; dx[j][5] = d1.y; dx[j][6] = d1.z; dx[j][7] = d1.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask >> rr) & 1u)) {
          if (s_ok & kbit) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) atomic_add_f32(feat + (int64_t)(s_row + koff) * C + sub + 16 * j, run[j]);
          }
#pragma unroll
          for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
          s_row = __builtin_amdgcn_readlane(row0, rr);
          s_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) run[j] = fmaf(w[i], dx[j][i], run[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s_ok & kbit) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) atomic_add_f32(feat + (int64_t)(s_row + koff) * C + sub + 16 * j, run[j]);
    }
  }
  {
    const int k = lane & 7;
    const int koff = (k & 1) * tp.su + ((k >> 1) & 1) * tp.sv + (k >> 2) * tp.st;
    const unsigned kbit = (lane < 8 && k < (voxel ? 8 : 4)) ? (1u << k) : 0u;
    const float4* wsrc = reinterpret_cast<const float4*>(wT + k * 32);
    float runw = 0.0f;
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask >> rr) & 1u)) {
          if (s_ok & kbit) atomic_add_f32(wgt + (int64_t)(s_row + koff), runw);
          runw = 0.0f;
          s_row = __builtin_amdgcn_readlane(row0, rr);
          s_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
        }
        runw += w[i];
      }
    }
    if (s_ok & kbit) atomic_add_f32(wgt + (int64_t)(s_row + koff), runw);
  }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// `rv` is a Renderer-shaped view of the arguments (rv.grid = the INPUT grid-list, rv.march) so that the
// Renderer's gather can be used as is.
template <int E, int CO, int GMI>
__global__ void __launch_bounds__(256, 2) splat_mlp_fwd_mfma(const LpSplatterArgs a, const LpRendererArgs rv,
                                                             const MlpSplatParams mp) {
  using M = LdsS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_mlp<E, CO>(a, mp, lds);
  __syncthreads();
  const float* lds_inf = lds + (M::INF - Lds::INF);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  float* const wv = lds + M::END + wave * M::PER_WAVE;
  float* const vt = wv + M::XT;
  float* const wT = wv + M::YT;
  const int blk = (int)blockIdx.x / mp.n_seg, seg = (int)blockIdx.x - blk * mp.n_seg;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[E / 2];
#pragma unroll
  for (int j = 0; j < E / 8; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(a.rays.encoding + rid * E + 8 * j + 4 * h);
    enc[4 * j + 0] = v.x; enc[4 * j + 1] = v.y; enc[4 * j + 2] = v.z; enc[4 * j + 3] = v.w;
  }
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool mask = a.march.mask_out_of_bounds != 0;
  const int per_seg = (s_tot + mp.n_seg - 1) / mp.n_seg;
  const int s_lo = seg * per_seg, s_hi = (s_lo + per_seg < s_tot) ? s_lo + per_seg : s_tot;
  for (int s = s_lo; s < s_hi; ++s) {
    Sample<E> sm;
    fetch_sample<E, GMI, true>(rv, lds_inf, ray, s, h, sm);
    const bool live = valid && !(mask && !point_in_bounds(sm.x, sm.y, sm.z));
    const int zo = opaque_zero();
    const float* wl = lds + (4 * h) * W_LD + r + zo;
    float xin[E / 2], h1[16];
#pragma unroll
    for (int q = 0; q < E / 2; ++q) xin[q] = sm.x0[q] + enc[q];
    f32x16 acc = layer<E / 2>(wl + M::W1, xin, bias16(lds + M::B1 + 4 * h + zo));
#pragma unroll
    for (int q = 0; q < 16; ++q) h1[q] = relu_f(acc[q]);
    acc = layer<16>(wl + M::W2, h1, bias16(lds + M::B2 + 4 * h + zo));
    LP_SCHED_FENCE();
    // output vector -> [channel][ray]
#pragma unroll
    for (int q = 0; q < CO / 2; ++q) vt[featq(q, h) * TM_LD + r] = acc[q];
#pragma unroll 1
    for (int g = 0; g < a.out.n_grids; ++g) {
      const LpGrid& og = a.out.grids[g];
      if (og.D > 1 && og.H > 1 && og.W > 1)
        splat_walk_vox<CO, 32>(a.out_feature, a.out_weight, og, ray.b, sm.x, sm.y, sm.z, live, lane,
                               SplatSrcLds{vt, TM_LD, lane & 15}, wT, 0);
      else
        splat_walk_lds<CO>(a.out_feature, a.out_weight, og, ray.b, sm.x, sm.y, sm.z, live, lane, vt, wT);
    }
  }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
LP_DEV constexpr int pi16m(int m) { return m < 4 ? 2 * m : (m < 12 ? 2 * (m - 4) + 1 : 2 * (m - 8)); }
LP_DEV void lds_barrier_m() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

LP_DEV void tile_store_m(float* tile, int r, int h, const float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[featq(q, h) * TM_LD + r] = v[q];
}

// one 16x16 dW quadrant over the rays of the source waves [v0, v1)
LP_DEV f32x4m dw_quadrant_m(const float* wave0, int a_off, int b_off, int v0, int v1, f32x4m acc, float& db) {
  float s = 0.0f;
  for (int v = v0; v < v1; ++v) {
    const float* base = wave0 + v * LdsS::PER_WAVE;
    const float4 a0 = *reinterpret_cast<const float4*>(base + a_off);
    const float4 a1 = *reinterpret_cast<const float4*>(base + a_off + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(base + b_off);
    const float4 b1 = *reinterpret_cast<const float4*>(base + b_off + 4);
    acc = LP_MFMA16M(a0.x, b0.x, acc);
    acc = LP_MFMA16M(a0.y, b0.y, acc);
    acc = LP_MFMA16M(a0.z, b0.z, acc);
    acc = LP_MFMA16M(a0.w, b0.w, acc);
    acc = LP_MFMA16M(a1.x, b1.x, acc);
    acc = LP_MFMA16M(a1.y, b1.y, acc);
    acc = LP_MFMA16M(a1.z, b1.z, acc);
    acc = LP_MFMA16M(a1.w, b1.w, acc);
    s += ((b0.x + b0.y) + (b0.z + b0.w)) + ((b1.x + b1.y) + (b1.z + b1.w));
  }
  db += s;
  return acc;
}

template <int E, int CO, int GMI>
__global__ void __launch_bounds__(256, 2) splat_mlp_bwd_mfma(const LpSplatterArgs a, const LpRendererArgs rv,
                                                             const MlpSplatParams mp) {
  using M = LdsS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_mlp<E, CO>(a, mp, lds);
  __syncthreads();
  const float* lds_inf = lds + (M::INF - Lds::INF);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  float* const wave0 = lds + M::END;
  float* const wv = wave0 + wave * M::PER_WAVE;
  float* const xt = wv + M::XT;
  float* const yt = wv + M::YT;
  const int blk = (int)blockIdx.x / mp.n_seg, seg = (int)blockIdx.x - blk * mp.n_seg;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[E / 2], denc[E / 2];
#pragma unroll
  for (int j = 0; j < E / 8; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(a.rays.encoding + rid * E + 8 * j + 4 * h);
    enc[4 * j + 0] = v.x; enc[4 * j + 1] = v.y; enc[4 * j + 2] = v.z; enc[4 * j + 3] = v.w;
  }
#pragma unroll
  for (int q = 0; q < E / 2; ++q) denc[q] = 0.0f;
  const bool want_params = a.grad_mlp_params != nullptr;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool mask = a.march.mask_out_of_bounds != 0;

  // dW quadrants: layer 2 is [32 x CO] (2 x CO/16 quadrants), layer 1 is [E x 32] (E/16 x 2 quadrants).
  // With nq quadrants the four waves form 4/nq groups; every group covers all quadrants over nq source waves.
  const int m16 = lane & 15, ka = lane >> 4;
  constexpr int NQ2 = 2 * (CO / 16), NQ1 = (E / 16) * 2;
  const int q2 = wave % NQ2, g2 = wave / NQ2;
  const int mi2 = q2 / (CO / 16), ni2 = q2 % (CO / 16);
  const int q1 = wave % NQ1, g1 = wave / NQ1;
  const int mi1 = q1 / 2, ni1 = q1 % 2;
  const int a_off2 = M::XT + (16 * mi2 + pi16m(m16)) * TM_LD + 8 * ka;
  const int b_off2 = M::YT + (16 * ni2 + pi16m(m16)) * TM_LD + 8 * ka;
  const int a_off1 = M::XT + (16 * mi1 + pi16m(m16)) * TM_LD + 8 * ka;
  const int b_off1 = M::YT + (16 * ni1 + pi16m(m16)) * TM_LD + 8 * ka;
  f32x4m dq1 = {0, 0, 0, 0}, dq2 = {0, 0, 0, 0};
  float db1 = 0.0f, db2 = 0.0f;

  const int per_seg = (s_tot + mp.n_seg - 1) / mp.n_seg;
  const int s_lo = seg * per_seg, s_hi = (s_lo + per_seg < s_tot) ? s_lo + per_seg : s_tot;
  for (int s = s_lo; s < s_hi; ++s) {
    Sample<E> sm;
    fetch_sample<E, GMI, true>(rv, lds_inf, ray, s, h, sm);
    const float x = sm.x, y = sm.y, z = sm.z;
    const bool live = valid && !(mask && !point_in_bounds(x, y, z));
    const int zo = opaque_zero();
    const float* wl = lds + (4 * h) * W_LD + r + zo;
    const float* wt = lds + r * W_LD + 4 * h + zo;
    // ---- forward recompute (the output vector itself is not needed) ----
    float xin[16], h1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) xin[q] = (q < E / 2) ? sm.x0[q < E / 2 ? q : 0] + enc[q < E / 2 ? q : 0] : 0.0f;
    f32x16 acc = layer<E / 2>(wl + M::W1, xin, bias16(lds + M::B1 + 4 * h + zo));
#pragma unroll
    for (int q = 0; q < 16; ++q) h1[q] = relu_f(acc[q]);
    LP_SCHED_FENCE();
    // ---- d v: gather of grad_out / clamp(weight) at the output taps (Splatter interpolation) ----
    float dv[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) dv[q] = 0.0f;
#pragma unroll 1
    for (int g = 0; g < a.out.n_grids; ++g) {
      Taps t;
      grid_taps<true>(a.out.grids[g], ray.b, x, y, z, t);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int row = t.row[k];
        const bool ok = live && row >= 0 && k < t.n;
        const int rowc = ok ? row : 0;
        const float wn = ok ? t.w[k] / fmaxf(a.weight[rowc], 1e-5f) : 0.0f;
        const float4* src = reinterpret_cast<const float4*>(a.grad_out + (int64_t)rowc * CO + 4 * h);
#pragma unroll
        for (int j = 0; j < CO / 8; ++j) {
          const float4 v = src[2 * j];
          dv[4 * j + 0] = fmaf(wn, v.x, dv[4 * j + 0]);
          dv[4 * j + 1] = fmaf(wn, v.y, dv[4 * j + 1]);
          dv[4 * j + 2] = fmaf(wn, v.z, dv[4 * j + 2]);
          dv[4 * j + 3] = fmaf(wn, v.w, dv[4 * j + 3]);
        }
        if (k == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    LP_SCHED_FENCE();
    // ---- layer 2: dW2 += h1^T dv ; dh1 = relu'(h1) (W2 dv) ----
    __builtin_amdgcn_s_setprio(1);  // barrier-coupled layer phases at raised priority (see lp_renderer_mfma_bwd.hip)
    if (want_params) {
      tile_store_m(xt, r, h, h1);
      tile_store_m(yt, r, h, dv);
    }
    acc = (f32x16){0};
    acc = layer_t(wt + M::W2, dv, acc);
    if (want_params) {
      lds_barrier_m();
      dq2 = dw_quadrant_m(wave0, a_off2, b_off2, g2 * NQ2, g2 * NQ2 + NQ2, dq2, db2);
      lds_barrier_m();
    }
    float dh1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) dh1[q] = (h1[q] > 0.0f) ? acc[q] : 0.0f;
    LP_SCHED_FENCE();
    // ---- layer 1: dW1 += xin^T dh1 ; dxin = W1 dh1 ----
    if (want_params) {
      tile_store_m(xt, r, h, xin);
      tile_store_m(yt, r, h, dh1);
    }
    acc = (f32x16){0};
    acc = layer_t(wt + M::W1, dh1, acc);  // rows >= E are zero weights
    if (want_params) {
      lds_barrier_m();
      dq1 = dw_quadrant_m(wave0, a_off1, b_off1, g1 * NQ1, g1 * NQ1 + NQ1, dq1, db1);
      lds_barrier_m();
    }
    if (live) {
#pragma unroll
      for (int q = 0; q < E / 2; ++q) denc[q] += acc[q];
    }
    LP_SCHED_FENCE();
    // ---- input-grid gradient: the Renderer's run-merged scatter of dxin ----
    __builtin_amdgcn_s_setprio(0);
    if (a.grad_input_grid_list[0]) {
#pragma unroll
      for (int q = 0; q < E / 2; ++q) xt[featq(q, h) * DX_LD + r] = acc[q];
#pragma unroll 1
      for (int g = 0; g < a.input_grid.n_grids; ++g)
        scatter_grid<E, GMI, false>(a.grad_input_grid_list[g], a.input_grid.grids[g], ray.b, x, y, z, live, lane, xt, yt, mp.dbg);
    }
  }

  if (valid && a.grad_encoding && mp.n_seg == 1) {
#pragma unroll
    for (int j = 0; j < E / 8; ++j)
      *reinterpret_cast<float4*>(a.grad_encoding + ray_id * E + 8 * j + 4 * h) =
          make_float4(denc[4 * j], denc[4 * j + 1], denc[4 * j + 2], denc[4 * j + 3]);
  } else if (valid && a.grad_encoding) {  // the segments of a ray add up (the launcher zero-fills)
#pragma unroll
    for (int j = 0; j < E / 8; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) atomic_add_f32(a.grad_encoding + ray_id * E + 8 * j + 4 * h + i, denc[4 * j + i]);
  }
  if (want_params) {
    float* G = a.grad_mlp_params;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int prow = pi16m(4 * ka + i);
      const int col2 = 16 * ni2 + pi16m(m16), col1 = 16 * ni1 + pi16m(m16);
      atomic_add_f32(G + mp.w2 + (16 * mi2 + prow) * CO + col2, dq2[i]);
      atomic_add_f32(G + mp.w1 + (16 * mi1 + prow) * 32 + col1, dq1[i]);
    }
    db1 += __shfl_xor(db1, 16); db1 += __shfl_xor(db1, 32);
    db2 += __shfl_xor(db2, 16); db2 += __shfl_xor(db2, 32);
    if (ka == 0) {
      if (mi2 == 0) atomic_add_f32(G + mp.b2 + 16 * ni2 + pi16m(m16), db2);
      if (mi1 == 0) atomic_add_f32(G + mp.b1 + 16 * ni1 + pi16m(m16), db1);
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

bool splatter_mlp_mfma_supported(const LpSplatterArgs& a) {
  const LpMlp& m = a.mlp;
  if (m.n_layers != 2 || m.dims[1] != 32) return false;
  if ((m.dims[0] != 16 && m.dims[0] != 32) || (m.dims[2] != 16 && m.dims[2] != 32)) return false;
  if (a.input_grid.n_rows * m.dims[0] * 4 >= (int64_t)1 << 32) return false;  // 32-bit scatter offsets
  if (a.out.n_rows >= (int64_t)1 << 31) return false;
  if (a.march.num_samples_inf > MAX_INF) return false;
  return true;
}

static int grid_mode_of(const LpGridList& gl) {
  auto is_voxel = [](const LpGrid& g) { return g.D > 1 && g.H > 1 && g.W > 1; };
  if (gl.n_grids == 1 && is_voxel(gl.grids[0])) return GM_VOXEL;
  if (is_canonical_triplane(gl)) return GM_TRIPLANE;
  return GM_GENERIC;
}

template <typename K>
static int launch_m(K kernel, const LpSplatterArgs& a, hipStream_t stream, bool backward) {
  const size_t lds = LdsS::TOTAL * sizeof(float);
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  LpRendererArgs rv = {};
  rv.grid = a.input_grid;
  rv.march = a.march;
  rv.rays = a.rays;
  MlpSplatParams mp;
  const int E = a.mlp.dims[0], CO = a.mlp.dims[2];
  mp.w1 = 0;
  mp.w2 = mp.w1 + (int64_t)E * 32;
  mp.b1 = mp.w2 + (int64_t)32 * CO;
  mp.b2 = mp.b1 + 32;
  static const int dbg = getenv("LP_MFMA_DEBUG") ? atoi(getenv("LP_MFMA_DEBUG")) : 0;
  mp.dbg = dbg;
  const unsigned nb = (unsigned)((a.rays.n_rays + WAVES * RAYS_PER_WAVE - 1) / (WAVES * RAYS_PER_WAVE));
  // small batches (see splat_segments in lp_splatter.hip): segments of >= 16 samples, within one round of workgroups
  static const int forced = getenv("LP_SPLAT_SEGMENTS") ? atoi(getenv("LP_SPLAT_SEGMENTS")) : 0;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  int n_seg = forced > 0 ? forced : (int)(512u / (nb ? nb : 1u));
  if (n_seg > s_tot / 16) n_seg = s_tot / 16;
  if (n_seg < 1) n_seg = 1;
  mp.n_seg = n_seg;
  if (backward && n_seg > 1 && a.grad_encoding) {
    const hipError_t e2 = hipMemsetAsync(a.grad_encoding, 0, (size_t)a.rays.n_rays * E * sizeof(float), stream);
    if (e2 != hipSuccess) return set_error((int)e2, "hipMemsetAsync(grad_encoding): %s", hipGetErrorString(e2));
  }
  hipLaunchKernelGGL(kernel, dim3(nb * (unsigned)n_seg), dim3(256), lds, stream, a, rv, mp);
  return LP_OK;
}

#define LP_DISPATCH_M(KERNEL, BWD)                                                                \
  do {                                                                                         \
    const int E = a.mlp.dims[0], CO = a.mlp.dims[2], gm = grid_mode_of(a.input_grid);          \
    if (E == 16 && CO == 16) {                                                                 \
      rc = gm == GM_TRIPLANE ? launch_m(KERNEL<16, 16, GM_TRIPLANE>, a, stream, BWD)                \
           : gm == GM_VOXEL  ? launch_m(KERNEL<16, 16, GM_VOXEL>, a, stream, BWD)                   \
                             : launch_m(KERNEL<16, 16, GM_GENERIC>, a, stream, BWD);                \
    } else if (E == 16) {                                                                      \
      rc = gm == GM_TRIPLANE ? launch_m(KERNEL<16, 32, GM_TRIPLANE>, a, stream, BWD)                \
           : gm == GM_VOXEL  ? launch_m(KERNEL<16, 32, GM_VOXEL>, a, stream, BWD)                   \
                             : launch_m(KERNEL<16, 32, GM_GENERIC>, a, stream, BWD);                \
    } else if (CO == 16) {                                                                     \
      rc = gm == GM_TRIPLANE ? launch_m(KERNEL<32, 16, GM_TRIPLANE>, a, stream, BWD)                \
           : gm == GM_VOXEL  ? launch_m(KERNEL<32, 16, GM_VOXEL>, a, stream, BWD)                   \
                             : launch_m(KERNEL<32, 16, GM_GENERIC>, a, stream, BWD);                \
    } else {                                                                                   \
      rc = gm == GM_TRIPLANE ? launch_m(KERNEL<32, 32, GM_TRIPLANE>, a, stream, BWD)                \
           : gm == GM_VOXEL  ? launch_m(KERNEL<32, 32, GM_VOXEL>, a, stream, BWD)                   \
                             : launch_m(KERNEL<32, 32, GM_GENERIC>, a, stream, BWD);                \
    }                                                                                          \
  } while (0)

int splatter_mlp_forward_mfma(const LpSplatterArgs& a, hipStream_t stream) {
  if (a.rays.n_rays == 0) return LP_OK;
  int rc;
  LP_DISPATCH_M(splat_mlp_fwd_mfma, false);
  if (rc) return rc;
  return check_launch("splat_mlp_fwd_mfma");
}

int splatter_mlp_backward_mfma(const LpSplatterArgs& a, hipStream_t stream) {
  if (a.rays.n_rays == 0) return LP_OK;
  int rc;
  LP_DISPATCH_M(splat_mlp_bwd_mfma, true);
  if (rc) return rc;
  return check_launch("splat_mlp_bwd_mfma");
}

}  // namespace lp
