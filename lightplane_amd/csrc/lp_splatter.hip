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
#include <stdlib.h>

#include "lp_device.h"
#include "lp_host.h"
#include "lp_splat_walk.h"

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

// ---------------------------------------------------------------------------------------
// Forward, run-merged ("walk") variant for C in {16, 32}: one wave = 32 rays.
// The encodings of the wave's rays are transposed once (LDS) so that every lane holds ONE channel
// of all 32 rays in registers; lane group `grp` (C lanes) works on tap slot k of the current sample.
// Rays are walked in order and consecutive rays that fall into the same cell (neighbouring pixels
// usually do) are summed in a register: one row-contiguous atomic (+ one weight atomic) per RUN
// instead of per ray.  Run boundaries are the same for all slots of a grid, i.e. one wave-uniform
// bit mask (ballot in the lane = ray layout): the walk is scalar-branched, tap rows come from
// v_readlane.  Same scheme as the Renderer's gradient scatter (lp_mfma_common.h).
// ---------------------------------------------------------------------------------------
// RPW rays per wave (16: twice as many waves in flight for the same ray count -- the walk is a
// latency-bound scalar loop, and 256x256 rays are only 2048 waves of 32).
// (Enc: enc[j][item] -- a register array [C / 16][RPW] of the wave's rays, or SplatEncConst for the transposed march)
template <int CPL>
struct SplatEncConst {
  float v[CPL];
  struct Row { float x; LP_DEV float operator[](int) const { return x; } };
  LP_DEV Row operator[](int j) const { return Row{v[j]}; }
};
template <int C, int RPW, class Enc>
LP_DEV void splat_walk(float* feat, float* wgt, const LpGrid& g, int b, float x, float y, float z, bool live,
                       int lane, const Enc& enc, float* wT, int dbg) {
  constexpr int CPL = C / 16;        // channels per lane: 16 lanes per tap slot, four slots per pass
  constexpr int NQ = 64 / RPW;       // lanes per ray in the lane = ray layout
  constexpr int SPQ = 8 / NQ;        // tap-weight slots each of them writes
  const int q = lane / RPW, r = lane % RPW, sub = lane & 15, grp = lane >> 4;
  TapSet tp;
  grid_tapset<true>(g, b, x, y, z, tp);
  if (!live) {
    tp.ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tp.w[k] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < SPQ; ++i) {
    float v = tp.w[i];
#pragma unroll
    for (int qq = 1; qq < NQ; ++qq) v = (q == qq) ? tp.w[qq * SPQ + i] : v;
    wT[(q * SPQ + i) * RPW + r] = v;
  }
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
    const float4* wsrc = reinterpret_cast<const float4*>(wT + k * RPW);
    float run[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
#pragma unroll
    for (int c8 = 0; c8 < RPW / 8; ++c8) {
      const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask >> rr) & 1u)) {
          if ((s_ok & kbit) && !(dbg & 1)) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) atomic_add_f32(feat + (int64_t)(s_row + koff) * C + sub + 16 * j, run[j]);
          }
#pragma unroll
          for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
          s_row = __builtin_amdgcn_readlane(row0, rr);
          s_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) run[j] = fmaf(w[i], enc[j][rr], run[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if ((s_ok & kbit) && !(dbg & 1)) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) atomic_add_f32(feat + (int64_t)(s_row + koff) * C + sub + 16 * j, run[j]);
    }
  }
  // unit weights: ONE more walk in which lane j < 8 owns tap slot j, so that a run costs a single
  // atomic instruction for all its slots (x-neighbours share a 64-byte segment of the weight grid)
  {
    const int k = lane & 7;
    const int koff = (k & 1) * tp.su + ((k >> 1) & 1) * tp.sv + (k >> 2) * tp.st;
    const unsigned kbit = (lane < 8 && k < (voxel ? 8 : 4)) ? (1u << k) : 0u;
    const float4* wsrc = reinterpret_cast<const float4*>(wT + k * RPW);
    float runw = 0.0f;
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
#pragma unroll
    for (int c8 = 0; c8 < RPW / 8; ++c8) {
      const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask >> rr) & 1u)) {
          if ((s_ok & kbit) && !(dbg & 2)) atomic_add_f32(wgt + (int64_t)(s_row + koff), runw);
          runw = 0.0f;
          s_row = __builtin_amdgcn_readlane(row0, rr);
          s_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
        }
        runw += w[i];
      }
    }
    if ((s_ok & kbit) && !(dbg & 2)) atomic_add_f32(wgt + (int64_t)(s_row + koff), runw);
  }
}

// W2 (all output grids voxel grids, C a multiple of 32): the voxel walk with two carry axes (splat_walk_vox2, lp_splat_walk.h) --
// lane = channel of 32, lane group = corner along the third axis
template <int C, int RPW, bool W2 = false>
__global__ void __launch_bounds__(256) splat_fwd_walk_kernel(const LpSplatterArgs a, int dbg, int n_seg, int grp) {
  constexpr int LD = RPW + 4;  // row stride of the transposed encoding tile [channel][ray]
  constexpr int NQ = 64 / RPW;
  constexpr int CPL = C / 16;
  constexpr int WLD = RPW + 8;  // padded stride of the weight table of the voxel walk (bank-conflict free, lp_splat_walk.h)
  __shared__ __attribute__((aligned(16))) float lds[4][C * LD > 8 * WLD ? C * LD : 8 * WLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane / RPW, r = lane % RPW;
  float* tile = lds[wave];
  // blockIdx = (ray block, segment of the march) -- samples are independent, so a segment is just a subset of the sample loop
  // launch order: groups of `grp` ray blocks; inside a group segment after segment, every ray block of the group for each
  const int n_blk = (int)gridDim.x / n_seg;
  const int per_group = grp * n_seg;
  const int gi = (int)blockIdx.x / per_group, gl = (int)blockIdx.x - gi * per_group;
  const int g_size = (n_blk - gi * grp < grp) ? n_blk - gi * grp : grp;  // (the last group may be short)
  const int seg = gl / g_size;
  const int blk = gi * grp + (gl - seg * g_size);
  const int64_t ray_id = ((int64_t)blk * 4 + wave) * RPW + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  // encoding -> [channel][ray]; lane (q, r) moves channels q*C/NQ .. of its ray
  constexpr int CPQ = C / NQ;
#pragma unroll
  for (int j = 0; j < CPQ / 4; ++j) {
    const int c0 = q * CPQ + 4 * j;
    const float4 v = *reinterpret_cast<const float4*>(a.rays.encoding + rid * C + c0);
    tile[(c0 + 0) * LD + r] = valid ? v.x : 0.0f;
    tile[(c0 + 1) * LD + r] = valid ? v.y : 0.0f;
    tile[(c0 + 2) * LD + r] = valid ? v.z : 0.0f;
    tile[(c0 + 3) * LD + r] = valid ? v.w : 0.0f;
  }
  constexpr int LPG = W2 ? 32 : 16;   // lanes (= channels) per lane group
  constexpr int NJ = C / LPG;
  float enc[NJ][RPW];  // channels (lane & (LPG - 1)) + LPG j of all rays
#pragma unroll
  for (int jc = 0; jc < NJ; ++jc) {
    const float4* src = reinterpret_cast<const float4*>(tile + ((lane & (LPG - 1)) + LPG * jc) * LD);
#pragma unroll
    for (int j = 0; j < RPW / 4; ++j) {
      const float4 v = src[j];
      enc[jc][4 * j + 0] = v.x; enc[jc][4 * j + 1] = v.y; enc[jc][4 * j + 2] = v.z; enc[jc][4 * j + 3] = v.w;
    }
  }
  float* wT = tile;  // the tile is free now: [8][RPW] tap weights of the current sample
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  // segment `seg` takes the samples seg, seg + n_seg, ...: interleaved, not a contiguous range (splat_forward_segments() on the host);
  // (dbg & 8: contiguous ranges, the A/B)
  const bool interleaved = !(dbg & 8);
  const int per_seg = (s_tot + n_seg - 1) / n_seg;
  const int s_lo = interleaved ? seg : seg * per_seg;
  const int s_hi = interleaved ? s_tot : ((s_lo + per_seg < s_tot) ? s_lo + per_seg : s_tot);
  const int s_step = interleaved ? n_seg : 1;
  for (int s = s_lo; s < s_hi; s += s_step) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    const bool live = valid && !(mask && !point_in_bounds(x, y, z));
    for (int g = 0; g < a.out.n_grids; ++g) {
      const LpGrid& og = a.out.grids[g];
      if constexpr (W2) {
        splat_walk_vox2<C, RPW, WLD, float[NJ][RPW]>(a.out_feature, a.out_weight, og, ray.b, x, y, z, live, lane, enc, wT, dbg);
      } else {
        if (og.D > 1 && og.H > 1 && og.W > 1 && !(dbg & 4))
          splat_walk_vox<C, RPW, SplatSrcRegs<CPL, RPW>, true, WLD>(a.out_feature, a.out_weight, og, ray.b, x, y, z, live, lane, SplatSrcRegs<CPL, RPW>{enc}, wT, dbg);
        else
          splat_walk<C, RPW>(a.out_feature, a.out_weight, og, ray.b, x, y, z, live, lane, enc, wT, dbg);
      }
    }
  }
}

// Forward walk with a TRANSPOSED MARCH (LpSplatterArgs.march_order == LP_MARCH_SAMPLES_PER_WAVE): a wave = ONE ray x 32 consecutive
// samples, its rays one after the other.  For batches of unrelated rays (the reference's splatter_speed_benchmark.py draws random
// rays): with lane = ray every ray is its own run -- 8 corner rows x C / 16 segments + 8 weight segments per ray-sample --, with lane =
// sample the same walk (run heads by ballot, carried columns, weight windows) merges the samples a ray spends in one cell and carries
// the shared face when it steps to a neighbour.  The splatted vector is constant along a ray: no transposition through LDS, no
// [channel][ray] register array; `rpw` rays per wave (a small batch is dealt over the chip by it).
// W2 (all output grids voxel grids, 32 / 64 channels): the voxel walk with two carry axes -- a ray steps along all three grid axes in
// the proportions of its direction, one carry axis catches the dominant one only
template <int C, bool W2 = false>
__global__ void __launch_bounds__(256) splat_fwd_ray_kernel(const LpSplatterArgs a, int dbg, int rpw) {
  constexpr int LPG = W2 ? 32 : 16;
  constexpr int CPL = C / LPG;
  constexpr int WLD = 32 + 8;
  __shared__ __attribute__((aligned(16))) float lds[4][8 * WLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31;
  float* wT = lds[wave];
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_blk = (s_tot + 31) >> 5;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  const int64_t ray0 = ((int64_t)blockIdx.x * 4 + wave) * rpw;
  for (int k = 0; k < rpw; ++k) {
    const int64_t rid = ray0 + k;
    if (rid >= a.rays.n_rays) break;  // wave-uniform
    const Ray ray = load_ray(a.rays, rid);
    SplatSrcConst<CPL> src;
    SplatEncConst<CPL> enc;
#pragma unroll
    for (int j = 0; j < CPL; ++j) src.v[j] = enc.v[j] = a.rays.encoding[rid * C + (lane & (LPG - 1)) + LPG * j];
    for (int bs = 0; bs < n_blk; ++bs) {
      const int s = bs * 32 + r;
      const int sc = s < s_tot ? s : s_tot - 1;
      const float depth = sample_depth(sc, a.march, ray.near_t, ray.far_t);
      float x, y, z;
      sample_point(ray, depth, contract, x, y, z);
      const bool live = s < s_tot && !(mask && !point_in_bounds(x, y, z));
      for (int g = 0; g < a.out.n_grids; ++g) {
        const LpGrid& og = a.out.grids[g];
        if constexpr (W2) {
          splat_walk_vox2<C, 32, WLD, SplatEncConst<CPL>, true>(a.out_feature, a.out_weight, og, ray.b, x, y, z, live, lane, enc, wT, dbg);
        } else {
          if (og.D > 1 && og.H > 1 && og.W > 1 && !(dbg & 4))
            splat_walk_vox<C, 32, SplatSrcConst<CPL>, true, WLD>(a.out_feature, a.out_weight, og, ray.b, x, y, z, live, lane, src, wT, dbg);
          else
            splat_walk<C, 32>(a.out_feature, a.out_weight, og, ray.b, x, y, z, live, lane, enc, wT, dbg);
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

// Backward on voxel grids as the mirror of splat_walk_vox: a wave = 16 image-adjacent rays, lane group = (y, z)
// corner pair, lane = channel.  The rays are walked in order; a run of rays in one cell reads the two x-columns
// of that cell ONCE (rows of grad_out scaled by 1 / clamp(weight): 64 contiguous bytes per lane group), every
// ray of the run accumulates w_near * column_near + w_far * column_far into its own register.  All loads of
// eight rays are issued before the first one is used.  The four corner pairs of a ray are summed across the
// lane groups at the end.  (The per-ray kernel below re-derives the geometry in every lane and reads eight rows
// per ray and sample.)
// RPW = rays per wave: 16 (default), or 8 -- half the per-lane accumulators and batch buffers, so twice the waves fit a SIMD
// and twice the gathers are in flight, for the price of the per-sample geometry being amortised over 8 rays instead of 16
// (LP_SPLAT_BWD_RPW, measured in DESIGN.md 4.4).
// The 64 / RPW lanes of a ray each take ONE SAMPLE of a block of 64 / RPW consecutive samples (round 6): the per-sample geometry
// (depth, point, tap set: ~60 % of the instructions of a VALU-bound kernel when every lane of a ray repeated it for the same sample)
// is computed once per (ray, sample); the walk then goes through the block's samples, reading the tap rows of sample j from the lanes
// j RPW .. and its weights from table j.
// waves per SIMD the 8-ray backward walk is compiled for, 32 / 16 channels (32: four since round 6 -- 128 registers, 19 spilled: cfg 3
// backward 0.707 -> 0.687 ms, cfg 5 49.8 -> 47.5 ms against three waves without spills)
#ifndef LP_SBW_OCC32
#define LP_SBW_OCC32 4
#endif
#ifndef LP_SBW_OCC16
#define LP_SBW_OCC16 4
#endif
template <int C, int B, int RPW = 16>
__global__ void __launch_bounds__(256, RPW == 8 ? (C < 32 ? LP_SBW_OCC16 : (C < 64 ? LP_SBW_OCC32 : 2)) : ((B == 4 && C < 64) ? 3 : 2))
splat_bwd_walk_kernel(const LpSplatterArgs a, int n_seg) {
  static_assert(RPW == 16 || (RPW == 8 && B == 8), "rays per wave: 16, or 8 with one batch of 8");
  constexpr int CPL = C / 16, NQ = 64 / RPW;
  __shared__ __attribute__((aligned(16))) float lds[4][NQ * 8 * RPW];  // per wave: [sample of the block][tap][ray]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane / RPW, r = lane % RPW, sub = lane & 15, grp = lane >> 4;
  float* wT = lds[wave];
  const int blk = (int)blockIdx.x / n_seg, seg = (int)blockIdx.x - blk * n_seg;  // (ray block, segment of the march)
  const int64_t ray0 = ((int64_t)blk * 4 + wave) * RPW;
  const bool valid = ray0 + r < a.rays.n_rays;
  const int64_t ray_id = patch_ray_index(ray0 + r, a.rays.n_rays, a.rays.row_length);  // (row_length: 2 x 4 / 4 x 4 pixel patches per wave)
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float acc[CPL][RPW];
#pragma unroll
  for (int j = 0; j < CPL; ++j)
#pragma unroll
    for (int i = 0; i < RPW; ++i) acc[j][i] = 0.0f;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask_oob = a.march.mask_out_of_bounds != 0;
  // segment `seg` takes the sample blocks seg, seg + n_seg, ... (interleaved like the forward's samples: contiguous ranges give the
  // segment in the middle of the march all the samples inside the grid -- cfg 3 backward 0.88 -> 0.71 ms with the same three segments;
  // the segments of a ray block stay side by side in the launch: issued segment-major the gather loses its L2 locality, 0.89 ms)
  const int s_hi = s_tot;
  for (int sb = seg * NQ; sb < s_hi; sb += NQ * n_seg) {
    const int s = sb + q;  // this lane's sample of the block
    const float depth = sample_depth(s < s_tot ? s : s_tot - 1, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    const bool live = valid && s < s_hi && !(mask_oob && !point_in_bounds(x, y, z));
    for (int g = 0; g < a.out.n_grids; ++g) {
      TapSet tp;
      grid_tapset<true>(a.out.grids[g], ray.b, x, y, z, tp);
      if (!live) {
        tp.ok = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) tp.w[k] = 0.0f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) wT[(q * 8 + k) * RPW + r] = tp.w[k];
      const int row0 = tp.row0;
      const int ok = (int)tp.ok;
      const int prow_ = lane_prev(row0), pok_ = lane_prev(ok);  // all lanes enabled: see run_head()
      const bool head = run_head(r, row0, prow_, ok, pok_);
      const unsigned long long heads = (unsigned long long)__ballot(head);
      const unsigned long long lives = (unsigned long long)__ballot(ok != 0);
      const int koff = (grp & 1) * tp.sv + (grp >> 1) * tp.st;
      const unsigned bit_lo = 1u << (2 * grp), bit_hi = 2u << (2 * grp);
      const int n_here = (s_hi - sb < NQ) ? s_hi - sb : NQ;
#pragma unroll 1
      for (int js = 0; js < n_here; ++js) {
        const int l0 = js * RPW;  // first lane of sample js of the block
        if (((lives >> l0) & ((1ull << RPW) - 1ull)) == 0ull) continue;  // no ray of the wave touches the grid at this sample
        const unsigned mask = ((unsigned)(heads >> l0) | (B == 8 ? 0x101u : 0x1111u)) & ((1u << RPW) - 1u);
        const float4* wlo = reinterpret_cast<const float4*>(wT + (js * 8 + 2 * grp) * RPW);
        const float4* whi = reinterpret_cast<const float4*>(wT + (js * 8 + 2 * grp + 1) * RPW);
#pragma unroll
        for (int cb = 0; cb < RPW / B; ++cb) {
          // columns of the runs that start in this batch of B rays (the first ray of a batch is forced to be a
          // run head: a run crossing a batch boundary is simply read again)
          float glo[B][CPL], ghi[B][CPL], ilo[B], ihi[B];
#pragma unroll
          for (int i = 0; i < B; ++i) {
            const int rr = B * cb + i;
            if ((mask >> rr) & 1u) {
              const int s_row = __builtin_amdgcn_readlane(row0, l0 + rr);
              const unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, l0 + rr);
              const int64_t row = (int64_t)(s_row + koff);
              const bool l_ok = (s_ok & bit_lo) != 0, h_ok = (s_ok & bit_hi) != 0;
              ilo[i] = l_ok ? a.weight[row] : 1.0f;
              ihi[i] = h_ok ? a.weight[row + 1] : 1.0f;
#pragma unroll
              for (int j = 0; j < CPL; ++j) {
                glo[i][j] = l_ok ? a.grad_out[row * C + sub + 16 * j] : 0.0f;
                ghi[i][j] = h_ok ? a.grad_out[(row + 1) * C + sub + 16 * j] : 0.0f;
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          float w0[B], w1[B];
#pragma unroll
          for (int i4 = 0; i4 < B / 4; ++i4) {
            const float4 a4 = wlo[(B / 4) * cb + i4], b4 = whi[(B / 4) * cb + i4];
            w0[4 * i4 + 0] = a4.x; w0[4 * i4 + 1] = a4.y; w0[4 * i4 + 2] = a4.z; w0[4 * i4 + 3] = a4.w;
            w1[4 * i4 + 0] = b4.x; w1[4 * i4 + 1] = b4.y; w1[4 * i4 + 2] = b4.z; w1[4 * i4 + 3] = b4.w;
          }
          float cl[CPL], ch[CPL];
#pragma unroll
          for (int j = 0; j < CPL; ++j) cl[j] = ch[j] = 0.0f;
#pragma unroll
          for (int i = 0; i < B; ++i) {
            const int rr = B * cb + i;
            if ((mask >> rr) & 1u) {
              // gradient of out = feat / max(weight, 1e-5) w.r.t. feat (weights carry no gradient)
              const float il = 1.0f / fmaxf(ilo[i], 1e-5f), ih = 1.0f / fmaxf(ihi[i], 1e-5f);
#pragma unroll
              for (int j = 0; j < CPL; ++j) {
                cl[j] = glo[i][j] * il;
                ch[j] = ghi[i][j] * ih;
              }
            }
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[j][rr] = fmaf(w1[i], ch[j], fmaf(w0[i], cl[j], acc[j][rr]));
          }
        }
      }
    }
  }
  // sum the four corner pairs (lane groups), then lane group 0 writes [ray][channel]
#pragma unroll
  for (int j = 0; j < CPL; ++j) {
#pragma unroll
    for (int i = 0; i < RPW; ++i) {
      float v = acc[j][i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      if (grp == 0 && ray0 + i < a.rays.n_rays) {
        float* dst = a.grad_encoding + patch_ray_index(ray0 + i, a.rays.n_rays, a.rays.row_length) * C + sub + 16 * j;
        if (n_seg > 1) atomic_add_f32(dst, v);  // the segments of a ray add up (the launcher zero-fills)
        else *dst = v;
      }
    }
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

// Small batches (BACKWARD): a wave marches its rays one sample after the other, so fewer than ~16 k rays leave most of the chip idle.
// The samples of a ray are independent in the Splatter, so the march is cut into segments of at least 16 samples, as many
// as bring the launch to ~3 workgroups per CU (LP_SPLAT_SEGMENTS=1 switches it off).  Walk kernels on MI355X, 256 samples
// into a 128^3 x 32 grid (scripts/bench_small_batch.py --splatter, profiles/r02_small_batch.txt): 4 096 rays backward
// 1.19 -> 0.23 ms; 16 384 rays 1.16 -> 0.81 ms; at 32 768 rays (512 ray blocks) two segments no longer pay (1.17 -> 1.29 ms),
// hence the 768.
static int splat_segments(const LpSplatterArgs& a, unsigned ray_blocks) {
  static const int forced = getenv("LP_SPLAT_SEGMENTS") ? atoi(getenv("LP_SPLAT_SEGMENTS")) : 0;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  int n = forced > 0 ? forced : (int)(768u / (ray_blocks ? ray_blocks : 1u));
  if (n > s_tot / 16) n = s_tot / 16;
  return n < 1 ? 1 : n;
}

// FORWARD walk (round 6, profiles/r06_splat_launch_order.txt): the march of every ray block is dealt to up to 32 workgroups that take
// INTERLEAVED samples (segment g: samples g, g + n, ...; at least 8 each).  Rounds 2-5 cut it into <= 3 CONTIGUOUS ranges: the samples
// in front of and behind the grid cost a fraction of those inside, the middle range's waves ran alone for most of the launch (average
// wave lifetime 31 % of the kernel's) -- cfg 3 forward 2.18 ms.  Interleaved and fine-grained every workgroup gets the same mix; and the
// ORDER the (ray block, segment) workgroups are issued in decides how far apart in time the atomics to one grid row are:
//   * segment-major (all ray blocks at segment 0, then at segment 1, ...): the chip works on ONE depth phase of the whole image at a time
//     -- cfg 3 forward 1.63 ms (20.4 Mrays/s fwd+bwd from 17.3); right while the grid is small enough to stay cached between
//     phases and a phase of the whole batch is short;
//   * ray-block-major (the segments of a ray block side by side): right for grids far beyond the caches (cfg 5, 256^3 x 32 = 2.15 GB:
//     forward 149.8 -> 139.4 ms; segment-major 157) and for very large batches (1024^2 rays: every phase re-reads all encodings).
// Measured on nine image / grid shapes; the rule below picks the faster order on each.  LP_SPLAT_FWD_SEGMENTS / LP_SPLAT_FWD_GROUP: A/B.
static int splat_forward_segments(const LpSplatterArgs& a) {
  static const int forced = getenv("LP_SPLAT_FWD_SEGMENTS") ? atoi(getenv("LP_SPLAT_FWD_SEGMENTS")) : 0;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  int n = forced > 0 ? forced : s_tot / 8;
  if (forced <= 0 && n > 32) n = 32;
  if (n > s_tot) n = s_tot;
  return n < 1 ? 1 : n;
}
// ray blocks per launch group: inside a group the workgroups are issued segment after segment (group = every ray block: segment-major;
// group = 1: ray-block-major)
static int splat_forward_group(const LpSplatterArgs& a, unsigned ray_blocks) {
  static const int forced = getenv("LP_SPLAT_FWD_GROUP") ? atoi(getenv("LP_SPLAT_FWD_GROUP")) : 0;
  if (forced > 0) return forced;
  const double grid_bytes = (double)a.out.n_rows * (double)a.out.channels * 4.0;
  return (grid_bytes <= 1.0e9 && ray_blocks <= 4096u) ? (int)ray_blocks : 1;
}

int splatter_forward_launch(const LpSplatterArgs& a, hipStream_t stream) {
  const int Cw = a.out.channels;
  static const bool no_walk = getenv("LP_SPLAT_NO_WALK") != nullptr;  // A/B timing knob
  if ((Cw == 16 || Cw == 32 || Cw == 64) && a.out.n_rows < ((int64_t)1 << 31) && !no_walk) {
    // 32 rays per wave since round 4 (C = 16 / 32): the per-sample geometry is amortised over twice the rays and a run is cut at every
    // 32nd ray instead of every 16th -- cfg 3 forward 2.26 -> 2.15 ms, cfg 5's splat forward 155.6 -> 149.4 ms
    // (profiles/r04_splat_fwd_rpw32_ab.txt); LP_SPLAT_RPW=16 selects the 16-ray waves (A/B, tests)
    static const int rpw = getenv("LP_SPLAT_RPW") ? atoi(getenv("LP_SPLAT_RPW")) : 32;
    static const int dbg = getenv("LP_SPLAT_DEBUG") ? atoi(getenv("LP_SPLAT_DEBUG")) : 0;  // timing experiments
    static const bool tm_off = getenv("LP_TM_OFF") != nullptr;
    if (a.march_order == LP_MARCH_SAMPLES_PER_WAVE && !tm_off) {  // batches of unrelated rays: one ray x 32 samples per wavefront
      if (a.rays.n_rays == 0) return LP_OK;
      static const int forced = getenv("LP_TM_RPW") ? atoi(getenv("LP_TM_RPW")) : 0;
      int rays_pw = 32;  // (rays per wave: fewer while the batch leaves workgroup slots idle -- 4 workgroups per CU)
      while (rays_pw > 1 && (a.rays.n_rays + 4 * rays_pw - 1) / (4 * rays_pw) < 1024) rays_pw >>= 1;
      if (forced >= 1 && forced <= 32) rays_pw = forced;
      const unsigned blocks = (unsigned)((a.rays.n_rays + 4 * rays_pw - 1) / (4 * rays_pw));
      static const bool walk2r = getenv("LP_SPLAT_WALK2") == nullptr || atoi(getenv("LP_SPLAT_WALK2")) != 0;
      bool all_vox = walk2r && !(dbg & 4);
      for (int g = 0; g < a.out.n_grids; ++g) all_vox = all_vox && a.out.grids[g].D > 1 && a.out.grids[g].H > 1 && a.out.grids[g].W > 1;
      if (all_vox && Cw == 64) hipLaunchKernelGGL((splat_fwd_ray_kernel<64, true>), dim3(blocks), dim3(256), 0, stream, a, dbg, rays_pw);
      else if (all_vox && Cw == 32) hipLaunchKernelGGL((splat_fwd_ray_kernel<32, true>), dim3(blocks), dim3(256), 0, stream, a, dbg, rays_pw);
      else if (Cw == 64) hipLaunchKernelGGL((splat_fwd_ray_kernel<64>), dim3(blocks), dim3(256), 0, stream, a, dbg, rays_pw);
      else if (Cw == 32) hipLaunchKernelGGL((splat_fwd_ray_kernel<32>), dim3(blocks), dim3(256), 0, stream, a, dbg, rays_pw);
      else hipLaunchKernelGGL((splat_fwd_ray_kernel<16>), dim3(blocks), dim3(256), 0, stream, a, dbg, rays_pw);
      return check_launch("splat_fwd_ray_kernel");
    }
    const int rpw_eff = Cw == 64 ? 16 : rpw;
    const unsigned ray_blocks = (unsigned)((a.rays.n_rays + 4 * rpw_eff - 1) / (4 * rpw_eff));
    if (ray_blocks == 0) return LP_OK;
    const int n_seg = splat_forward_segments(a);
    const unsigned blocks = ray_blocks * (unsigned)n_seg;
    const int grp = splat_forward_group(a, ray_blocks);
    // (64 channels -- the reference's own speed benchmark splats into [1,160,160,160,64] -- : four channels per lane)
    // voxel grids only, 32 / 64 channels: the walk with two carry axes  (LP_SPLAT_WALK2=0: the one-axis walk, A/B)
    static const bool walk2 = getenv("LP_SPLAT_WALK2") == nullptr || atoi(getenv("LP_SPLAT_WALK2")) != 0;
    bool all_voxel = walk2 && !(dbg & 4);
    for (int g = 0; g < a.out.n_grids; ++g) all_voxel = all_voxel && a.out.grids[g].D > 1 && a.out.grids[g].H > 1 && a.out.grids[g].W > 1;
    if (all_voxel && Cw == 64) hipLaunchKernelGGL((splat_fwd_walk_kernel<64, 16, true>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    else if (all_voxel && Cw == 32 && rpw == 32) hipLaunchKernelGGL((splat_fwd_walk_kernel<32, 32, true>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    else if (Cw == 64) hipLaunchKernelGGL((splat_fwd_walk_kernel<64, 16>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    else if (Cw == 16 && rpw == 32) hipLaunchKernelGGL((splat_fwd_walk_kernel<16, 32>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    else if (Cw == 16) hipLaunchKernelGGL((splat_fwd_walk_kernel<16, 16>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    else if (rpw == 32) hipLaunchKernelGGL((splat_fwd_walk_kernel<32, 32>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    else hipLaunchKernelGGL((splat_fwd_walk_kernel<32, 16>), dim3(blocks), dim3(256), 0, stream, a, dbg, n_seg, grp);
    return check_launch("splat_fwd_walk_kernel");
  }
  LP_SPLAT_DISPATCH(splat_fwd_kernel);
  return check_launch("splat_fwd_kernel");
}

int splatter_backward_launch(const LpSplatterArgs& a, hipStream_t stream) {
  const int Cw = a.out.channels;
  static const bool no_walk = getenv("LP_SPLAT_NO_WALK") != nullptr;  // A/B timing knob
  // (plane grids take the same walk: their tap sets have four slots, so the two lane groups of the second z layer see
  // validity bits and weights of zero and contribute nothing -- half the lanes idle, but a run of rays still reads its
  // rows once instead of once per ray)
  if ((Cw == 16 || Cw == 32 || Cw == 64) && a.out.n_grids > 0 && a.out.n_rows < ((int64_t)1 << 31) && !no_walk) {
    // 8 rays per wave (round 4): cfg 3 backward 2.32 -> 2.01 ms -- 146 instead of 210 registers (C = 32), three waves per SIMD
    // instead of two, i.e. 1.5x the gathers in flight of a latency-bound walk; LP_SPLAT_BWD_RPW=16 selects the 16-ray waves
    static const int rpw = getenv("LP_SPLAT_BWD_RPW") ? atoi(getenv("LP_SPLAT_BWD_RPW")) : 8;
    const int rpw_eff = rpw == 8 ? 8 : 16;
    const unsigned ray_blocks = (unsigned)((a.rays.n_rays + 4 * rpw_eff - 1) / (4 * rpw_eff));
    if (ray_blocks == 0) return LP_OK;
    int n_seg = splat_segments(a, rpw_eff == 8 ? (ray_blocks + 1) / 2 : ray_blocks);
    // Mid-sized batches: fill whole rounds of resident waves.  cfg 3 = 8 192 waves of 8 rays on 3 072 wave slots (C = 32: three waves
    // per SIMD) = 2.67 rounds, the last one a third empty; three segments make it exactly 8 rounds: backward 2.04 -> 1.94 ms (2 or 4
    // segments, 5.33 / 10.67 rounds: 2.18 / 2.21 ms -- profiles/r04_knob_sweep.txt).  Among 1, 2, 3, 4, 6 segments the fewest that bring
    // the last round to >= 90 % full, for launches of up to 16 rounds (beyond that the tail does not matter).
    static const int forced_seg = getenv("LP_SPLAT_SEGMENTS") ? atoi(getenv("LP_SPLAT_SEGMENTS")) : 0;
    if (n_seg == 1 && forced_seg <= 0) {
      static int n_simd = 0;
      if (n_simd == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0)
          n_simd = 4 * cus;
        else n_simd = 1024;
      }
      // resident workgroups per CU (= waves per SIMD: four-wave workgroups) of the instantiation that will run, asked from
      // the runtime once per instantiation (C = 16 / 32 / 64 at 8 rays per wave: 4 / 3 / 2 with this compiler) -- not a table
      // that has to track the register allocator
      const int ki = (Cw == 16 ? 0 : Cw == 32 ? 1 : 2) + (rpw_eff == 8 ? 0 : 3);
      static int occ_cache[6] = {0, 0, 0, 0, 0, 0};
      if (occ_cache[ki] == 0) {
        const void* kfn = ki == 0 ? (const void*)splat_bwd_walk_kernel<16, 8, 8> : ki == 1 ? (const void*)splat_bwd_walk_kernel<32, 8, 8>
                        : ki == 2 ? (const void*)splat_bwd_walk_kernel<64, 8, 8> : ki == 3 ? (const void*)splat_bwd_walk_kernel<16, 8>
                        : ki == 4 ? (const void*)splat_bwd_walk_kernel<32, 8> : (const void*)splat_bwd_walk_kernel<64, 4>;
        int nb = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kfn, 256, 0) != hipSuccess || nb < 1) nb = 2;
        occ_cache[ki] = nb > 8 ? 8 : nb;
      }
      const int occ = occ_cache[ki];
      const uint64_t cap = (uint64_t)n_simd * occ, waves = (uint64_t)ray_blocks * 4;
      const int s_tot = a.march.num_samples + a.march.num_samples_inf;
      if (waves >= cap && waves <= 16 * cap) {
        // (round 6: the segments take INTERLEAVED sample blocks, so more of them also even out the work of a wave -- a launch of up to
        // four rounds starts at three segments whatever its round arithmetic says: 256^2 rays x 16 ch, exactly two rounds: 0.80 -> 0.65 ms)
        const int cand[5] = {1, 2, 3, 4, 6};
        for (int i = (waves <= 4 * cap) ? 2 : 0; i < 5; ++i) {
          const int n = cand[i];
          if (n > 1 && s_tot / n < 16) break;
          const uint64_t tot = waves * n, rounds = (tot + cap - 1) / cap;
          if ((double)tot / (double)(rounds * cap) >= 0.9) { n_seg = n; break; }
        }
      }
    }
    // (With n_seg > 1 the segments of a ray ADD their partial grad_encoding with fp32 atomics: the sum's order, hence its last
    // bits, vary from run to run -- like every grid / parameter gradient of this library.  LP_SPLAT_SEGMENTS=1 restores the
    // single-writer, bit-reproducible form.)
    if (n_seg > 1) {  // the segments accumulate into grad_encoding
      const hipError_t e = hipMemsetAsync(a.grad_encoding, 0, (size_t)a.rays.n_rays * Cw * sizeof(float), stream);
      if (e != hipSuccess) return set_error((int)e, "hipMemsetAsync(grad_encoding): %s", hipGetErrorString(e));
    }
    const unsigned blocks = ray_blocks * (unsigned)n_seg;
    // batches of 8 rays; batches of 4 at three waves/SIMD measured the same (cfg 3)
    if (Cw == 16 && rpw_eff == 8) hipLaunchKernelGGL((splat_bwd_walk_kernel<16, 8, 8>), dim3(blocks), dim3(256), 0, stream, a, n_seg);
    else if (Cw == 32 && rpw_eff == 8) hipLaunchKernelGGL((splat_bwd_walk_kernel<32, 8, 8>), dim3(blocks), dim3(256), 0, stream, a, n_seg);
    else if (Cw == 64 && rpw_eff == 8) hipLaunchKernelGGL((splat_bwd_walk_kernel<64, 8, 8>), dim3(blocks), dim3(256), 0, stream, a, n_seg);
    else if (Cw == 16) hipLaunchKernelGGL((splat_bwd_walk_kernel<16, 8>), dim3(blocks), dim3(256), 0, stream, a, n_seg);
    else if (Cw == 32) hipLaunchKernelGGL((splat_bwd_walk_kernel<32, 8>), dim3(blocks), dim3(256), 0, stream, a, n_seg);
    else hipLaunchKernelGGL((splat_bwd_walk_kernel<64, 4>), dim3(blocks), dim3(256), 0, stream, a, n_seg);  // (batches of 8: 300 B of scratch)
    return check_launch("splat_bwd_walk_kernel");
  }
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
