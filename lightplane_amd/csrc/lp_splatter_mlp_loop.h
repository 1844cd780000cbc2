// lp_splatter_mlp_loop.h -- device code of the LAYER-LOOPED bf16x3 MFMA family of the MLP-Splatter.
//
// LightplaneMLPSplatter is free-form (splatter_module.py:164-331: mlp_n_layers, mlp_hidden_chn; default: two layers) and
// the reference's own sweep uses hidden width 64, 3-4 layers and 32 / 64 input features
// (tests/test_splatter_with_autograd.py:49-51).  This family -- the only MFMA family of the MLP-Splatter since round 4 -- covers
//   2..4 layers [E, H, .., H, Cout], E (input-grid channels = encoding width) and H in {16, 32, 64}, Cout in {16, 32}
// with the building blocks of lp_loop.h (32 x 32 blocks of row-major bf16 limb images, six limb products per chunk on
// v_mfma_f32_32x32x16_bf16, workgroup-shared fp32 dW quadrants), the gather / scatter of the Renderer for the input
// grid-list and the run-merged walks of the Splatter for the output grid-list.
// The backward keeps the hidden activations of the recompute in registers; NB = 2 (any width above 32) runs at one wave
// per SIMD.
#pragma once
#include "lp_host.h"
#include "lp_loop.h"
#include "lp_splat_walk.h"

namespace lp {

constexpr int SLOOP_MAX = 4;  // layers

struct SplatLoopParams {
  int n;                       // layers
  LoopLayer l[SLOOP_MAX];
  int inf;                     // float offset of the beyond-far table in the small block
  int img_end;                 // bytes before the per-wave tiles
  int n_seg;                   // segments the march is cut into (blockIdx = (ray block, segment))
  int fwd_group;               // forward: > 0 = INTERLEAVED samples per segment (g, g + n_seg, ...), issued in groups of this many ray
                               // blocks segment after segment (lp_splatter.hip, splat_forward_segments / _group); 0 = contiguous ranges
  int dbg;
};

// per-wave LDS area (floats): two [32][36] tiles, contiguous (together: the [64][36] dx tile of a 64-channel input grid),
// and the 8 x 32 weight table of the walks
struct SplatLoopTile {
  static constexpr int XT = 0;
  static constexpr int YT = 32 * LT_LD;
  static constexpr int WT = 2 * 32 * LT_LD;
  static constexpr int PER_WAVE = WT + 8 * 32;
};

// the forward keeps one tile (the vector to splat, [channel][ray]) and the weight table per wave: 5.6 KB instead of 10.2 KB, so
// that EIGHT waves fit beside 96 KB of limb images
struct SplatLoopTileFwd {
  static constexpr int XT = 0;
  static constexpr int WT = 32 * LT_LD;
  static constexpr int PER_WAVE = WT + 8 * 32;
};

template <int NB>
LP_DEV void sloop_stage(const LpSplatterArgs& a, const SplatLoopParams& sp, float* lds) {
  const int tid = threadIdx.x;
  for (int l = 0; l < sp.n; ++l) loop_stage_layer<NB>(reinterpret_cast<char*>(lds), lds, a.mlp_params, sp.l[l], tid);
  for (int i = tid; i < LOOP_N_INF; i += (int)blockDim.x) lds[sp.inf + i] = (i < a.march.num_samples_inf) ? inf_scale(i, a.march) : 0.0f;
}

// sampled feature + ray encoding -> NB blocks: x0 register q holds channel 8 (q >> 2) + 4 h + (q & 3), i.e. register
// 16 blk + q' is feature feat(q', h) of block blk
template <int E, int NB>
LP_DEV void sloop_input(const float (&x0)[E / 2], const float (&enc)[E / 2], float (&out)[NB][16]) {
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool in = 16 * blk + q < E / 2;
      out[blk][q] = in ? x0[in ? 16 * blk + q : 0] + enc[in ? 16 * blk + q : 0] : 0.0f;
    }
  }
}

template <int E>
LP_DEV void sloop_load_encoding(const LpSplatterArgs& a, int64_t rid, int h, float (&enc)[E / 2]) {
#pragma unroll
  for (int j = 0; j < E / 8; ++j) {
    const float4 v = *reinterpret_cast<const float4*>(a.rays.encoding + rid * E + 8 * j + 4 * h);
    enc[4 * j + 0] = v.x; enc[4 * j + 1] = v.y; enc[4 * j + 2] = v.z; enc[4 * j + 3] = v.w;
  }
}

// Splatter-side walk of one plane output grid: the vector to splat sits in LDS as [channel][ray] (tile `vT`); features per
// run and slot, then the unit weights eight slots at a time 
template <int C>
LP_DEV void sloop_walk_lds(float* feat, float* wgt, const LpGrid& g, int b, float x, float y, float z, bool live, int lane,
                           const float* vT, float* wT) {
  constexpr int CPL = C / 16;
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
  const int prow_ = lane_prev(row0), pok_ = lane_prev(ok);
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
        const float4* dsrc = reinterpret_cast<const float4*>(vT + (sub + 16 * j) * LT_LD);
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

// ---------------------------------------------------------------------------------------------------------------
// forward: `rv` is a Renderer-shaped view of the arguments (rv.grid = the INPUT grid-list) for the Renderer's gather
// ---------------------------------------------------------------------------------------------------------------
// NW: waves per workgroup -- 4, or 8 where the limb images exclude a second four-wave workgroup per CU (three or four 64-wide
// layers: 72 / 96 KB): eight waves share one copy of the images, two waves per SIMD (sloop_launch_fwd; the staging loops stride
// by 256 threads, the upper four waves re-write what the lower four write)
template <int E, int CO, int NB, int NW = WAVES>
__global__ void __launch_bounds__(64 * NW, 2) splat_mlp_fwd_loop(const LpSplatterArgs a, const LpRendererArgs rv,
                                                                            const SplatLoopParams sp) {
  using T = SplatLoopTileFwd;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  sloop_stage<NB>(a, sp, lds);
  __syncthreads();
  const float* const geo = lds + sp.inf - Lds::INF;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  float* const wv = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + sp.img_end) + wave * T::PER_WAVE;
  float* const vt = wv + T::XT;
  float* const wT = wv + T::WT;
  // (ray block, segment): the plain Splatter's forward launch shape (splat_fwd_walk_kernel) -- interleaved samples, grouped issue order
  const int grp = sp.fwd_group > 0 ? sp.fwd_group : 1;
  const int n_blk = (int)gridDim.x / sp.n_seg;
  const int per_group = grp * sp.n_seg;
  const int gi = (int)blockIdx.x / per_group, gl = (int)blockIdx.x - gi * per_group;
  const int g_size = (n_blk - gi * grp < grp) ? n_blk - gi * grp : grp;
  const int seg = gl / g_size;
  const int blk = gi * grp + (gl - seg * g_size);
  const int64_t ray_id = ((int64_t)blk * NW + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[E / 2];
  sloop_load_encoding<E>(a, rid, h, enc);
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool mask = a.march.mask_out_of_bounds != 0;
  const bool interleaved = sp.fwd_group > 0;
  const int per_seg = (s_tot + sp.n_seg - 1) / sp.n_seg;
  const int s_lo = interleaved ? seg : seg * per_seg;
  const int s_hi = interleaved ? s_tot : ((s_lo + per_seg < s_tot) ? s_lo + per_seg : s_tot);
  const int s_step = interleaved ? sp.n_seg : 1;
  for (int s = s_lo; s < s_hi; s += s_step) {
    Sample<E> sm;
    fetch_sample<E, GM_GENERIC, true>(rv, geo, ray, s, h, sm);
    const bool live = valid && !(mask && !point_in_bounds(sm.x, sm.y, sm.z));
    const int zo = opaque_zero();
    const char* lbase = reinterpret_cast<const char*>(lds) + zo;
    const float* smf = lds + zo;
    float cur[NB][16];
    sloop_input<E, NB>(sm.x0, enc, cur);
#pragma unroll
    for (int l = 0; l < SLOOP_MAX; ++l) {
      if (l < sp.n - 1) {
        float nxt[NB][16];
        loop_layer_fwd<NB>(lbase, smf, sp.l[l], lane, cur, nxt);
        loop_copy<NB>(nxt, cur);
      }
    }
    float v[NB][16];
    // the output layer (no activation): one of l[1] .. l[3]
#pragma unroll
    for (int l = 1; l < SLOOP_MAX; ++l) {
      if (l == sp.n - 1) loop_layer_fwd<NB, false>(lbase, smf, sp.l[l], lane, cur, v);
    }
    LP_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < CO / 2; ++q) vt[featq(q, h) * LT_LD + r] = v[0][q];
#pragma unroll 1
    for (int g = 0; g < a.out.n_grids; ++g) {
      const LpGrid& og = a.out.grids[g];
      if (og.D > 1 && og.H > 1 && og.W > 1)
        splat_walk_vox<CO, 32>(a.out_feature, a.out_weight, og, ray.b, sm.x, sm.y, sm.z, live, lane, SplatSrcLds{vt, LT_LD, lane & 15}, wT, 0);
      else
        sloop_walk_lds<CO>(a.out_feature, a.out_weight, og, ray.b, sm.x, sm.y, sm.z, live, lane, vt, wT);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
// ML = layers the instantiation is unrolled for: SLOOP_MAX, or 2 -- a two-layer MLP keeps ONE hidden activation and fits 256
// registers without spills (lp_splatter_mlp_loop_shallow.hip).  Every width-<= 32 instantiation is compiled for two waves per SIMD
// (its images + tiles are <= 63 KB: two workgroups per CU): the four-layer ones were at 244 + 32 ... 256 + 82 registers, i.e. one
// wave per SIMD for a handful of registers; under the bound <32,16> fits 254 without a spill and <32,32> spills 70 and is STILL 17 %
// faster ([32,32,32,32]: 10.03 -> 8.30 ms fwd+bwd, profiles/r04_loop_shallow_ab.txt)
template <int E, int CO, int NB, int ML = SLOOP_MAX>
__global__ void __launch_bounds__(256, NB == 1 ? 2 : 1) splat_mlp_bwd_loop(const LpSplatterArgs a, const LpRendererArgs rv, const SplatLoopParams sp) {
  using T = SplatLoopTile;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  sloop_stage<NB>(a, sp, lds);
  __syncthreads();
  const float* const geo = lds + sp.inf - Lds::INF;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, r = lane & 31;
  float* const wave0 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + sp.img_end);
  float* const wv = wave0 + wave * T::PER_WAVE;
  float* const xt = wv + T::XT;
  float* const yt = wv + T::YT;
  float* const wT = wv + T::WT;
  const int blk = (int)blockIdx.x / sp.n_seg, seg = (int)blockIdx.x - blk * sp.n_seg;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[E / 2], denc[E / 2];
  sloop_load_encoding<E>(a, rid, h, enc);
#pragma unroll
  for (int q = 0; q < E / 2; ++q) denc[q] = 0.0f;
  const bool want_params = a.grad_mlp_params != nullptr;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool mask = a.march.mask_out_of_bounds != 0;
  const int mi = wave >> 1, ni = wave & 1;
  const int m16 = lane & 15, ka = lane >> 4;
#if LP_LOOP_DW_BF16
  // byte offsets of this MFMA lane's supplier address in a source wave's X / dY limb tiles (lp_loop.h, loop_dw_quadrant_bf)
  const int a_off = T::XT * 4 + rm_off(loop_rho(8 * ka + (m16 >> 2)), 4 * (m16 & 3)) + 32 * mi;
  const int b_off = T::YT * 4 + rm_off(loop_rho(8 * ka + (m16 >> 2)), 4 * (m16 & 3)) + 32 * ni;
#else
  const int a_off = T::XT + (16 * mi + pi16l(m16)) * LT_LD + 8 * ka;
  const int b_off = T::YT + (16 * ni + pi16l(m16)) * LT_LD + 8 * ka;
#endif
  LoopDw<NB> dw[ML];
#pragma unroll
  for (int l = 0; l < ML; ++l) loop_dw_zero<NB>(dw[l]);

  const int per_seg = (s_tot + sp.n_seg - 1) / sp.n_seg;
  const int s_lo = seg * per_seg, s_hi = (s_lo + per_seg < s_tot) ? s_lo + per_seg : s_tot;
  for (int s = s_lo; s < s_hi; ++s) {
    Sample<E> sm;
    fetch_sample<E, GM_GENERIC, true>(rv, geo, ray, s, h, sm);
    const float x = sm.x, y = sm.y, z = sm.z;
    const bool live = valid && !(mask && !point_in_bounds(x, y, z));
    const int zo = opaque_zero();
    const char* lbase = reinterpret_cast<const char*>(lds) + zo;
    const float* smf = lds + zo;
    // ---- forward recompute: the hidden activations are kept (the output vector itself is not needed) ----
    float xin[NB][16];
    sloop_input<E, NB>(sm.x0, enc, xin);
    float act[ML - 1][NB][16];  // act[l] = output of layer l (post-ReLU), l < n - 1
#pragma unroll
    for (int l = 0; l < ML - 1; ++l) {
      if (l < sp.n - 1) {
        if (l == 0) loop_layer_fwd<NB>(lbase, smf, sp.l[0], lane, xin, act[0]);
        else loop_layer_fwd<NB>(lbase, smf, sp.l[l], lane, act[l > 0 ? l - 1 : 0], act[l]);
      }
    }
    LP_SCHED_FENCE();
    // ---- d v: gather of grad_out / clamp(weight) at the output taps (Splatter interpolation) ----
    float g[NB][16];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) g[b][q] = 0.0f;
    }
#pragma unroll 1
    for (int gi = 0; gi < a.out.n_grids; ++gi) {
      Taps t;
      grid_taps<true>(a.out.grids[gi], ray.b, x, y, z, t);
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
          g[0][4 * j + 0] = fmaf(wn, v.x, g[0][4 * j + 0]);
          g[0][4 * j + 1] = fmaf(wn, v.y, g[0][4 * j + 1]);
          g[0][4 * j + 2] = fmaf(wn, v.z, g[0][4 * j + 2]);
          g[0][4 * j + 3] = fmaf(wn, v.w, g[0][4 * j + 3]);
        }
        if (k == 3) __builtin_amdgcn_sched_barrier(0);
      }
    }
    LP_SCHED_FENCE();
    // ---- layers, last -> first: dW (workgroup-shared quadrants), dX through the ReLU of the layer's input ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int l = ML - 1; l >= 0; --l) {
      if (l < sp.n) {
        f32x16 dx[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) dx[b] = (f32x16){0};
        if (l > 0) {
          loop_layer_bwd<NB>(lbase, sp.l[l], lane, xt, yt, wave0, T::PER_WAVE, a_off, b_off, want_params, true, act[l > 0 ? l - 1 : 0], g, dw[l], dx);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = (act[l > 0 ? l - 1 : 0][b][q] > 0.0f) ? dx[b][q] : 0.0f;
          }
        } else {
          loop_layer_bwd<NB>(lbase, sp.l[0], lane, xt, yt, wave0, T::PER_WAVE, a_off, b_off, want_params, true, xin, g, dw[0], dx);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = dx[b][q];
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
    // g = d (sampled feature + encoding)
    if (live) {
#pragma unroll
      for (int q = 0; q < E / 2; ++q) denc[q] += g[q >> 4][q & 15];
    }
    LP_SCHED_FENCE();
    // ---- input-grid gradient: the Renderer's run-merged scatter; the dx tile [E][36] spans both tiles when E = 64 ----
    if (a.grad_input_grid_list[0]) {
#pragma unroll
      for (int q = 0; q < E / 2; ++q) xt[(32 * (q >> 4) + featq(q & 15, h)) * DX_LD + r] = g[q >> 4][q & 15];
#pragma unroll 1
      for (int gi = 0; gi < a.input_grid.n_grids; ++gi)
        scatter_grid<E, GM_GENERIC, false>(a.grad_input_grid_list[gi], a.input_grid.grids[gi], ray.b, x, y, z, live, lane, xt, wT, sp.dbg);
    }
  }

  if (valid && a.grad_encoding && sp.n_seg == 1) {
#pragma unroll
    for (int j = 0; j < E / 8; ++j)
      *reinterpret_cast<float4*>(a.grad_encoding + ray_id * E + 8 * j + 4 * h) =
          make_float4(denc[4 * j], denc[4 * j + 1], denc[4 * j + 2], denc[4 * j + 3]);
  } else if (valid && a.grad_encoding) {  // the segments of a ray add up (the launcher zero-fills)
#pragma unroll
    for (int j = 0; j < E / 8; ++j) {
#pragma unroll
      for (int i = 0; i < 4; ++i) atomic_add_f32(a.grad_encoding + ray_id * E + 8 * j + 4 * h + i, denc[4 * j + i]);
    }
  }
  if (want_params) {
#pragma unroll
    for (int l = 0; l < ML; ++l) {
      if (l < sp.n) loop_dw_flush<NB>(a.grad_mlp_params, sp.l[l], dw[l], wave, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// parameters / launch helper shared by the family's two translation units
// ---------------------------------------------------------------------------------------------------------------
static bool width_ok(int w) { return w == 16 || w == 32 || w == 64; }

static SplatLoopParams sloop_params(const LpSplatterArgs& a, int NB) {
  SplatLoopParams p = {};
  const LpMlp& m = a.mlp;
  p.n = m.n_layers;
  int64_t off = m.offset, boff = m.offset;
  for (int l = 0; l < m.n_layers; ++l) boff += (int64_t)m.dims[l] * m.dims[l + 1];
  int f = 0, img = 0;
  for (int l = 0; l < m.n_layers && l < SLOOP_MAX; ++l) {
    p.l[l].w = off;
    p.l[l].b = boff;
    p.l[l].rows_in = m.dims[l];
    p.l[l].cols = m.dims[l + 1];
    p.l[l].ld = m.dims[l + 1];
    p.l[l].ob = (m.dims[l + 1] + 31) / 32;
    p.l[l].bias = f;
    f += 32 * NB;
    p.l[l].img = img;
    img += loop_layer_bytes(m.dims[l], m.dims[l + 1]);
    off += (int64_t)m.dims[l] * m.dims[l + 1];
    boff += m.dims[l + 1];
  }
  p.inf = f;
  f += LOOP_N_INF;
  const int small_bytes = f * 4;
  for (int l = 0; l < p.n; ++l) p.l[l].img += small_bytes;
  p.img_end = small_bytes + img;
  static const int dbg = getenv("LP_MFMA_DEBUG") ? atoi(getenv("LP_MFMA_DEBUG")) : 0;
  p.dbg = dbg;
  p.n_seg = 1;
  p.fwd_group = 0;
  return p;
}

static int sloop_nb(const LpSplatterArgs& a) {
  int w = a.mlp.dims[0];
  for (int l = 1; l < a.mlp.n_layers; ++l) w = a.mlp.dims[l] > w ? a.mlp.dims[l] : w;
  return w <= 32 ? 1 : 2;
}

template <typename K>
static int sloop_launch(K kernel, const LpSplatterArgs& a, hipStream_t stream, bool backward, int nw = WAVES,
                        int per_wave = SplatLoopTile::PER_WAVE);

// forward: four-wave workgroups, two per CU, where images + tiles fit twice in the 160 KB; otherwise ONE eight-wave workgroup
// per CU over one copy of the images (two-block MLPs of three / four layers) -- two waves per SIMD either way
template <int E, int CO, int NB>
static int sloop_launch_fwd(const LpSplatterArgs& a, hipStream_t stream) {
  if constexpr (NB == 2) {
    static const bool no_nw8 = getenv("LP_LOOP_FWD_NW4") != nullptr;  // A/B
    static const bool force8 = getenv("LP_LOOP_FWD_NW8") != nullptr;  // tests: also for batches below one round of workgroups
    const SplatLoopParams p0 = sloop_params(a, NB);
    const size_t lds4 = (size_t)p0.img_end + (size_t)WAVES * SplatLoopTileFwd::PER_WAVE * 4;
    const size_t lds8 = (size_t)p0.img_end + (size_t)8 * SplatLoopTileFwd::PER_WAVE * 4;
    const unsigned nb8 = (unsigned)((a.rays.n_rays + 8 * RAYS_PER_WAVE - 1) / (8 * RAYS_PER_WAVE));
    if (2 * lds4 > 160 * 1024 && lds8 <= 160 * 1024 && (nb8 >= 256u || force8) && !no_nw8)
      return sloop_launch(splat_mlp_fwd_loop<E, CO, NB, 8>, a, stream, false, 8, SplatLoopTileFwd::PER_WAVE);
  }
  return sloop_launch(splat_mlp_fwd_loop<E, CO, NB, WAVES>, a, stream, false, WAVES, SplatLoopTileFwd::PER_WAVE);
}

template <typename K>
static int sloop_launch(K kernel, const LpSplatterArgs& a, hipStream_t stream, bool backward, int nw, int per_wave) {
  SplatLoopParams p = sloop_params(a, sloop_nb(a));
  const size_t lds = (size_t)p.img_end + (size_t)nw * per_wave * 4;
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  LpRendererArgs rv = {};
  rv.grid = a.input_grid;
  rv.march = a.march;
  rv.rays = a.rays;
  const unsigned nb = (unsigned)((a.rays.n_rays + nw * RAYS_PER_WAVE - 1) / (nw * RAYS_PER_WAVE));
  // small batches (see splat_segments in lp_splatter.hip): segments of >= 16 samples, within one round of workgroups
  static const int forced = getenv("LP_SPLAT_SEGMENTS") ? atoi(getenv("LP_SPLAT_SEGMENTS")) : 0;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  int n_seg = forced > 0 ? forced : (int)(256u / (nb ? nb : 1u));
  if (n_seg > s_tot / 16) n_seg = s_tot / 16;
  if (n_seg < 1) n_seg = 1;
  p.fwd_group = 0;
  if (!backward) {
    // forward: as the plain Splatter's (lp_splatter.hip: interleaved segments of >= 16 samples here -- every workgroup stages the MLP's
    // limb images --, segment-major while the output grid is cache-sized; LP_SPLAT_FWD_SEGMENTS / LP_SPLAT_FWD_GROUP: A/B, 0 group = rounds 2-5)
    static const int forced_f = getenv("LP_SPLAT_FWD_SEGMENTS") ? atoi(getenv("LP_SPLAT_FWD_SEGMENTS")) : 0;
    static const int forced_g = getenv("LP_SPLAT_FWD_GROUP") ? atoi(getenv("LP_SPLAT_FWD_GROUP")) : -1;
    if (forced_g != 0) {
      n_seg = forced_f > 0 ? forced_f : s_tot / 16;
      if (forced_f <= 0 && n_seg > 16) n_seg = 16;
      if (n_seg > s_tot) n_seg = s_tot;
      if (n_seg < 1) n_seg = 1;
      const double grid_bytes = (double)a.out.n_rows * (double)a.out.channels * 4.0;
      p.fwd_group = forced_g > 0 ? forced_g : ((grid_bytes <= 1.0e9 && nb <= 4096u) ? (int)nb : 1);
    }
  }
  p.n_seg = n_seg;
  if (backward && n_seg > 1 && a.grad_encoding) {
    const hipError_t e2 = hipMemsetAsync(a.grad_encoding, 0, (size_t)a.rays.n_rays * a.mlp.dims[0] * sizeof(float), stream);
    if (e2 != hipSuccess) return set_error((int)e2, "hipMemsetAsync(grad_encoding): %s", hipGetErrorString(e2));
  }
  hipLaunchKernelGGL(kernel, dim3(nb * (unsigned)n_seg), dim3(64 * nw), lds, stream, a, rv, p);
  return LP_OK;
}

// backward of a TWO-layer MLP [E, H, Cout], E, H in {16, 32}, at two waves per SIMD: lp_splatter_mlp_loop_shallow.hip
int splatter_mlp_backward_loop_shallow(const LpSplatterArgs& a, hipStream_t stream);

}  // namespace lp
