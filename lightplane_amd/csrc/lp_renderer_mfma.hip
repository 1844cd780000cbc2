// lp_renderer_mfma.hip -- Renderer forward / backward on the CDNA4 matrix cores (gfx950).
//
// Shape family (anything else falls back to lp_renderer_generic.hip): single grid-list with
// C in {16, 32} channels, trunk [C,32,32], opacity [32,32,1], colour [32,32,>=Cc] with Cc <= 4.
// That covers every BASELINE.json configuration.
//
// Mapping.  One wave = 32 rays.  Lane l = (h = l>>5, r = l&31) works on ray r and on the
// feature subset F_h = { feat(q,h) = (q&3) + 8*(q>>2) + 4*h : q = 0..15 } of every 32-wide
// activation -- exactly the rows a lane receives from v_mfma_f32_32x32x2_f32 when the product
// is formed TRANSPOSED:  Y^T[out, ray] = W^T[out, k] * X^T[k, ray].
//   A operand (lane l)  = W[feat(kk,h)][l&31]            (weights, read from LDS)
//   B operand (lane l)  = X[ray l&31][feat(kk,h)]        (= accumulator register kk of the
//                                                          previous layer: no data movement)
//   D register q        = Y[ray l&31][feat(q,h)]
// so the whole trunk -> heads chain (and the dX chain of the backward) runs register to
// register; bias + ReLU are lane-local.  K-slot kk of an MFMA pairs feature feat(kk,0) (lanes
// 0-31) with feat(kk,1) (lanes 32-63): the contraction order is a permutation of the feature
// index, which a sum does not care about.  fp32 MFMA is bit-for-bit an fmaf chain, so the
// numerics are plain fp32.  (Layout algebra checked in scripts/mfma_layout_check.py.)
//
// The 12 (triplane) / 8 (voxel) corner gathers of a ray are split between its two lanes by
// 16-byte channel chunks (lane h loads channels 8j+4h .. 8j+4h+3): dwordx4 loads, features land
// directly in B-operand order.  The gather of sample s+1 is issued BEFORE the MFMA chain of
// sample s (software pipelining by hand): the loads fly and the interpolation VALU work issues in
// the shadow of the 64-cycle MFMAs, and -- in the backward -- the loads are older than the
// atomics of the current sample in the (in-order) vmcnt queue, so waiting for them does not drain
// the atomics.  For the common grid-list shapes (three planes / one voxel grid) the sample loop
// body is one branch-free basic block so that the scheduler can actually interleave the two.
//
// Backward (far -> near, recompute): weight gradients dW = X^T dY contract over RAYS, i.e. over
// lanes; X and dY are transposed through a per-wave padded LDS tile ([ray][33]) and fed to the
// same MFMA (A = X[ray 2kk+h][l&31], B = dY[ray 2kk+h][l&31]); the 32x32 dW tiles stay in
// accumulator registers for the whole kernel.  Grid gradients are transposed through LDS as well
// so that every global_atomic_add_f32 instruction covers whole contiguous C-float rows (measured
// on MI355X: 336 Gadd/s vs 19.6 Gadd/s lane-per-row), and contributions of neighbouring rays to
// the same cell are merged in a register first (run-length merge).
#include "lp_mfma_common.h"

namespace lp {

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int C, int GM>
__global__ void __launch_bounds__(256, 2) renderer_fwd_mfma(const LpRendererArgs a, const MfmaParams mp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights<C, false>(a, mp, lds);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int64_t ray_id = ((int64_t)blockIdx.x * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[16];
  load_encoding(a, rid, h, enc);
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_ckpt = ckpt_count(a.march);
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;
  float nlt = 0.0f, t_prev = 1.0f, len = 0.0f, depth_prev = 0.0f;
  float facc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  Sample<C> nx;
  fetch_sample<C, GM>(a, lds, ray, 0, h, nx);
  Act<C> t;
  for (int s = 0; s < s_tot; ++s) {
    const float depth = nx.depth, occ = nx.occ;
#pragma unroll
    for (int q = 0; q < C / 2; ++q) t.x0[q] = nx.x0[q];
    // software pipeline: the next sample's gather is interleaved with this sample's MFMA chain
    const int zo = opaque_zero();
    const Heads hd = decode_prefetch<C, GM>(a, lds, ray, lane, enc, t, (s + 1 < s_tot) ? s + 1 : s, nx, zo);
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    depth_prev = depth;
    float raw = hd.raw_o;
    if (a.noise_sigma > 0.0f) raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    nlt = nlt + opacity * delta;
    if (a.neg_log_t_ckpt && valid && h == 0) {
      const int ck = ckpt_index(s, a.march);
      if (ck >= 0) a.neg_log_t_ckpt[ray_id * n_ckpt + ck] = nlt;
    }
    const float tr = __expf(-nlt);
    const float w = t_prev - tr;
    t_prev = tr;
    len = fmaf(w, depth, len);
#pragma unroll
    for (int c = 0; c < 4; ++c) facc[c] = fmaf(w, sigmoid_f(hd.raw_c[c]) * occ, facc[c]);
  }
  if (valid && h == 0) {
    a.ray_length[ray_id] = len;
    a.neg_log_t[ray_id] = nlt;
    for (int c = 0; c < a.color_chn; ++c) a.feature[ray_id * a.color_chn + c] = facc[c];
  }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------

// write a 16-register activation (accumulator order) into a [ray][33] tile
LP_DEV void tile_store16(float* tile, int r, int h, const float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[r * TILE_LD + featq(q, h)] = v[q];
}

// dW tile (16 accumulator registers, kept for the whole kernel) += X^T dY with X, dY read
// transposed from the tiles.  Register q of lane l is dW[feat(q,h)][l&31].
// (ds_add_f32 into a block-shared LDS copy was measured 8x slower than the whole rest of the
// kernel: LDS float atomics retire ~0.3 lane-adds per clock per CU on gfx950.)
LP_DEV f32x16 dw_mfma(const float* tx, const float* ty, int lane, f32x16 acc) {
  const int h = lane >> 5, j = lane & 31;
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = LP_MFMA(tx[(2 * kk + h) * TILE_LD + j], ty[(2 * kk + h) * TILE_LD + j], acc);
  return acc;
}

// column sums of a tile over this half's 16 rays (bias gradient partial)
LP_DEV float tile_colsum(const float* ty, int lane) {
  const int h = lane >> 5, j = lane & 31;
  float s = 0.0f;
#pragma unroll 4
  for (int rr = 0; rr < 16; ++rr) s += ty[(16 * h + rr) * TILE_LD + j];
  return s;
}

template <int C, int GM, int OCC, bool PIPE>
__global__ void __launch_bounds__(256, OCC) renderer_bwd_mfma(const LpRendererArgs a, const MfmaParams mp) {
  using M = Lds;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights<C, true>(a, mp, lds);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  float* tx = lds + M::WAVE0 + wave * M::PER_WAVE + M::TX;
  float* ty = lds + M::WAVE0 + wave * M::PER_WAVE + M::TY;
  float* ts = lds + M::WAVE0 + wave * M::PER_WAVE + M::TS;
  for (int i = lane; i < M::PER_WAVE; i += 64) tx[i] = 0.0f;
  __syncthreads();

  const int64_t ray_id = ((int64_t)blockIdx.x * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[16], denc[16];
  load_encoding(a, rid, h, enc);
#pragma unroll
  for (int q = 0; q < 16; ++q) denc[q] = 0.0f;
  float gfeat[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    gfeat[c] = (valid && a.grad_feature && c < a.color_chn) ? a.grad_feature[rid * a.color_chn + c] : 0.0f;
  const float g_len = (valid && a.grad_ray_length) ? a.grad_ray_length[rid] : 0.0f;
  const float g_nlt = (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f;

  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_ckpt = ckpt_count(a.march);
  const bool want_params = a.grad_mlp_params != nullptr;
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;

  // weight-gradient accumulators (whole kernel): four 32x32 MFMA tiles + small partials
  f32x16 dw_t1 = {0}, dw_t2 = {0}, dw_o1 = {0}, dw_c1 = {0};
  float db_t1 = 0.0f, db_t2 = 0.0f, db_o1 = 0.0f, db_c1 = 0.0f;
  // output layers of the heads: lane (f = l&31, half h) owns feature f, partial over 16 rays
  float dwo2 = 0.0f, dwc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float dbo2 = 0.0f, dbc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  float nlt = a.neg_log_t[rid];
  float suffix = 0.0f, p_next = 0.0f;
  Sample<C> nx;
  if (PIPE) fetch_sample<C, GM>(a, lds, ray, s_tot - 1, h, nx);
  Act<C> t;
  for (int s = s_tot - 1; s >= 0; --s) {
    if (!PIPE) fetch_sample<C, GM>(a, lds, ray, s, h, nx);
    const float depth = nx.depth, occ = nx.occ, x = nx.x, y = nx.y, z = nx.z;
#pragma unroll
    for (int q = 0; q < C / 2; ++q) t.x0[q] = nx.x0[q];
    // software pipeline: the gather of the next (nearer) sample is interleaved with this sample's
    // MFMA chain and, more importantly, is issued before this sample's atomics
    const int zo = opaque_zero();
    const float* ldz = lds + zo;
    const float* wt = lds + r * W_LD + 4 * h + zo;  // dX operand base of this lane
    const Heads hd = decode_prefetch<C, GM, PIPE>(a, lds, ray, lane, enc, t, (s > 0) ? s - 1 : 0, nx, zo);
    const float depth_prev = PIPE ? nx.depth
                                  : sample_depth_tab((s > 0) ? s - 1 : 0, a.march, ray.near_t, ray.far_t, lds + M::INF);
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    float raw = hd.raw_o;
    if (a.noise_sigma > 0.0f) raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    if (a.neg_log_t_ckpt) {
      const int ck = ckpt_index(s, a.march);
      if (ck >= 0) nlt = a.neg_log_t_ckpt[rid * n_ckpt + ck];
    }
    const float t_i = __expf(-nlt);
    nlt = fmaxf(nlt - opacity * delta, 0.0f);
    const float t_im1 = __expf(-nlt);
    const float w = t_im1 - t_i;
    float sg[4];
    float p_i = g_len * depth;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sg[c] = sigmoid_f(hd.raw_c[c]);
      p_i = fmaf(gfeat[c], sg[c] * occ, p_i);
    }
    suffix = fmaf(t_i, p_i - p_next, suffix);
    p_next = p_i;
    const float d_a = suffix + g_nlt;
    const float dro = valid ? d_a * delta * a.gain * occ * d_softplus_f(raw) : 0.0f;
    float drc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) drc[c] = valid ? w * gfeat[c] * occ * sg[c] * (1.0f - sg[c]) : 0.0f;

    LP_SCHED_FENCE();
    // ---- output layers of the heads (VALU): gradient w.r.t. ho / hc ----
    float dho[16], dhc[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 wo = *reinterpret_cast<const float4*>(ldz + M::WO2 + 8 * j + 4 * h);
      const float wov[4] = {wo.x, wo.y, wo.z, wo.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = 4 * j + i;
        const float4 wc = *reinterpret_cast<const float4*>(ldz + M::WC2 + (8 * j + 4 * h + i) * 4);
        dho[q] = (t.ho[q] > 0.0f) ? dro * wov[i] : 0.0f;
        float v = drc[0] * wc.x;
        v = fmaf(drc[1], wc.y, v);
        v = fmaf(drc[2], wc.z, v);
        v = fmaf(drc[3], wc.w, v);
        dhc[q] = (t.hc[q] > 0.0f) ? v : 0.0f;
      }
    }
    if (h == 0) {  // per-ray scalars: count each ray once
      dbo2 += dro;
#pragma unroll
      for (int c = 0; c < 4; ++c) dbc2[c] += drc[c];
    }
    if (want_params) {
      // weight gradients of the two output layers: dW[f] += sum_ray h[ray][f] * d_raw[ray].
      // ho / hc go through the transposition tiles, the per-ray scalars through `ts`.
      tile_store16(tx, r, h, t.ho);
      tile_store16(ty, r, h, t.hc);
      if (h == 0) {
        *reinterpret_cast<float4*>(ts + r * 8) = make_float4(drc[0], drc[1], drc[2], drc[3]);
        ts[r * 8 + 4] = dro;
      }
      const int f = lane & 31;
#pragma unroll 2
      for (int rr = 0; rr < 16; ++rr) {
        const int ry = 16 * h + rr;
        const float hov = tx[ry * TILE_LD + f], hcv = ty[ry * TILE_LD + f];
        const float4 dc = *reinterpret_cast<const float4*>(ts + ry * 8);
        dwo2 = fmaf(hov, ts[ry * 8 + 4], dwo2);
        dwc2[0] = fmaf(hcv, dc.x, dwc2[0]);
        dwc2[1] = fmaf(hcv, dc.y, dwc2[1]);
        dwc2[2] = fmaf(hcv, dc.z, dwc2[2]);
        dwc2[3] = fmaf(hcv, dc.w, dwc2[3]);
      }
    }

    LP_SCHED_FENCE();
    // ---- colour hidden layer: dW_c1 += (e+enc)^T dhc ; d(e+enc) = Wc1 dhc ----
    if (want_params) {
      float ein[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) ein[q] = t.e[q] + enc[q];
      tile_store16(tx, r, h, ein);
      tile_store16(ty, r, h, dhc);
      dw_c1 = dw_mfma(tx, ty, lane, dw_c1);
      db_c1 += tile_colsum(ty, lane);
    }
    f32x16 acc = {0};
    acc = layer_t(wt + M::WC1, dhc, acc);
#pragma unroll
    for (int q = 0; q < 16; ++q) denc[q] += acc[q];
    LP_SCHED_FENCE();
    // ---- opacity hidden layer: dW_o1 += e^T dho ; de += Wo1 dho ----
    if (want_params) {
      tile_store16(tx, r, h, t.e);
      tile_store16(ty, r, h, dho);
      dw_o1 = dw_mfma(tx, ty, lane, dw_o1);
      db_o1 += tile_colsum(ty, lane);
    }
    acc = layer_t(wt + M::WO1, dho, acc);
    float de[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) de[q] = (t.e[q] > 0.0f) ? acc[q] : 0.0f;
    LP_SCHED_FENCE();
    // ---- trunk layer 2 ----
    if (want_params) {
      tile_store16(tx, r, h, t.h1);
      tile_store16(ty, r, h, de);
      dw_t2 = dw_mfma(tx, ty, lane, dw_t2);
      db_t2 += tile_colsum(ty, lane);
    }
    acc = (f32x16){0};
    acc = layer_t(wt + M::WT2, de, acc);
    float dh1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) dh1[q] = (t.h1[q] > 0.0f) ? acc[q] : 0.0f;
    LP_SCHED_FENCE();
    // ---- trunk layer 1 ----
    if (want_params) {
      // x0 tile: only the first C columns carry data, the rest must read as zero
#pragma unroll
      for (int q = 0; q < 16; ++q) tx[r * TILE_LD + featq(q, h)] = (q < C / 2) ? t.x0[q < C / 2 ? q : 0] : 0.0f;
      tile_store16(ty, r, h, dh1);
      dw_t1 = dw_mfma(tx, ty, lane, dw_t1);
      db_t1 += tile_colsum(ty, lane);
    }
    LP_SCHED_FENCE();
    if (a.grad_grid) {
      acc = (f32x16){0};
      acc = layer_t(wt + M::WT1, dh1, acc);  // rows >= C are zero weights
      const int rep = (int)(blockIdx.x % (unsigned)(a.n_grad_replicas + 1));
      float* gg = rep == 0 ? a.grad_grid : a.grad_grid_replicas + (int64_t)(rep - 1) * a.grid.n_rows * C;
      // ---- grid gradient: dx0 transposed through LDS to [channel][ray]; every lane then holds one
      //      channel of all 32 rays (tiles tx/ty are free here and serve as scratch) ----
      float* dxT = tx;
      float* wT = tx + C * DX_LD;
#pragma unroll
      for (int q = 0; q < C / 2; ++q) dxT[featq(q, h) * DX_LD + r] = acc[q];
      float dxr[32];
      {
        const float4* dsrc = reinterpret_cast<const float4*>(dxT + (lane % C) * DX_LD);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float4 v = dsrc[j];
          dxr[4 * j + 0] = v.x; dxr[4 * j + 1] = v.y; dxr[4 * j + 2] = v.z; dxr[4 * j + 3] = v.w;
        }
      }
      const bool live = valid && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));
      const int ng = (GM == GM_TRIPLANE) ? 3 : (GM == GM_VOXEL) ? 1 : a.grid.n_grids;
      if (!(mp.dbg & 2)) {
#pragma unroll 1
        for (int g = 0; g < ng; ++g) scatter_grid<C>(gg, a.grid.grids[g], ray.b, x, y, z, live, lane, dxr, wT, mp.dbg);
      }
    }
  }

  // ---- epilogue ----
  if (valid && a.grad_encoding) {
    float4* dst = reinterpret_cast<float4*>(a.grad_encoding + ray_id * HID + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[2 * j] = make_float4(denc[4 * j], denc[4 * j + 1], denc[4 * j + 2], denc[4 * j + 3]);
  }
  if (want_params) {
    float* G = a.grad_mlp_params;
    // hidden-layer biases: lane j (both halves hold a partial over 16 rays each)
    const int j = lane & 31;
    atomic_add_f32(G + mp.b_t1 + j, db_t1);
    atomic_add_f32(G + mp.b_t2 + j, db_t2);
    atomic_add_f32(G + mp.b_o1 + j, db_o1);
    atomic_add_f32(G + mp.b_c1 + j, db_c1);
    // head output layers: lane (f, h) holds the partial over the 16 rays of its half
    atomic_add_f32(G + mp.w_o2 + j, dwo2);
    for (int c = 0; c < a.color_chn; ++c) atomic_add_f32(G + mp.w_c2 + (int64_t)j * mp.ldc2 + c, dwc2[c]);
    float v = dbo2, c0 = dbc2[0], c1 = dbc2[1], c2 = dbc2[2], c3 = dbc2[3];
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      v += __shfl_xor(v, m);
      c0 += __shfl_xor(c0, m);
      c1 += __shfl_xor(c1, m);
      c2 += __shfl_xor(c2, m);
      c3 += __shfl_xor(c3, m);
    }
    if (lane == 0) {
      atomic_add_f32(G + mp.b_o2, v);
      const float cv[4] = {c0, c1, c2, c3};
      for (int c = 0; c < a.color_chn; ++c) atomic_add_f32(G + mp.b_c2 + c, cv[c]);
    }
    // the four 32x32 tiles: summed over the waves of the block in LDS (once per kernel), then one
    // global flush per block
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int o = featq(q, h) * 32 + j;
      atomicAdd(lds + M::DW + o, dw_t1[q]);
      atomicAdd(lds + M::DW + 1024 + o, dw_t2[q]);
      atomicAdd(lds + M::DW + 2048 + o, dw_o1[q]);
      atomicAdd(lds + M::DW + 3072 + o, dw_c1[q]);
    }
  }
  __syncthreads();
  if (want_params) {
    float* G = a.grad_mlp_params;
    for (int i = threadIdx.x; i < 1024; i += 256) {
      if (i < C * 32) atomic_add_f32(G + mp.w_t1 + i, lds[M::DW + i]);
      atomic_add_f32(G + mp.w_t2 + i, lds[M::DW + 1024 + i]);
      atomic_add_f32(G + mp.w_o1 + i, lds[M::DW + 2048 + i]);
      atomic_add_f32(G + mp.w_c1 + i, lds[M::DW + 3072 + i]);
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

bool renderer_mfma_supported(const LpRendererArgs& a, const char** why) {
  *why = "";
  const int C = a.grid.channels;
  if (a.color_grid.n_grids > 0) { *why = "separate colour grid"; return false; }
  if (C != 16 && C != 32) { *why = "grid channels not 16 or 32"; return false; }
  if (a.trunk.n_layers != 2 || a.opacity.n_layers != 2 || a.color.n_layers != 2) {
    *why = "layer counts other than trunk 2 / opacity 2 / colour 2";
    return false;
  }
  if (a.trunk.dims[1] != HID || a.trunk.dims[2] != HID || a.opacity.dims[1] != HID || a.color.dims[1] != HID) {
    *why = "hidden width other than 32";
    return false;
  }
  if (a.color_chn > 4) { *why = "more than 4 colour channels"; return false; }
  if (a.grid.n_rows >= (int64_t)1 << 31) { *why = "grid-list has 2^31 rows or more"; return false; }
  if (a.march.num_samples_inf > MAX_INF) { *why = "more than 256 beyond-far samples"; return false; }
  return true;
}

static MfmaParams make_params(const LpRendererArgs& a) {
  MfmaParams p;
  const int C = a.grid.channels;
  p.w_t1 = a.trunk.offset;
  p.w_t2 = p.w_t1 + (int64_t)C * HID;
  p.b_t1 = p.w_t2 + HID * HID;
  p.b_t2 = p.b_t1 + HID;
  p.w_o1 = a.opacity.offset;
  p.w_o2 = p.w_o1 + HID * HID;
  p.b_o1 = p.w_o2 + HID;
  p.b_o2 = p.b_o1 + HID;
  p.ldc2 = a.color.dims[2];
  p.w_c1 = a.color.offset;
  p.w_c2 = p.w_c1 + HID * HID;
  p.b_c1 = p.w_c2 + (int64_t)HID * p.ldc2;
  p.b_c2 = p.b_c1 + HID;
  static const int dbg = getenv("LP_MFMA_DEBUG") ? atoi(getenv("LP_MFMA_DEBUG")) : 0;
  p.dbg = dbg;
  return p;
}

static int grid_mode(const LpRendererArgs& a) {
  auto is_voxel = [](const LpGrid& g) { return g.D > 1 && g.H > 1 && g.W > 1; };
  static const bool force_generic = getenv("LP_MFMA_GENERIC_GRIDS") != nullptr;  // debugging aid
  if (force_generic) return GM_GENERIC;
  if (a.grid.n_grids == 1 && is_voxel(a.grid.grids[0])) return GM_VOXEL;
  if (a.grid.n_grids == 3 && !is_voxel(a.grid.grids[0]) && !is_voxel(a.grid.grids[1]) && !is_voxel(a.grid.grids[2]))
    return GM_TRIPLANE;
  return GM_GENERIC;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) {
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  return LP_OK;
}

static unsigned n_blocks(const LpRendererArgs& a) {
  return (unsigned)((a.rays.n_rays + WAVES * RAYS_PER_WAVE - 1) / (WAVES * RAYS_PER_WAVE));
}

template <int C, int GM>
static int launch_fwd(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
  const size_t lds = Lds::FWD_END * sizeof(float);
  int rc;
  if ((rc = set_lds(renderer_fwd_mfma<C, GM>, lds))) return rc;
  hipLaunchKernelGGL((renderer_fwd_mfma<C, GM>), dim3(n_blocks(a)), dim3(256), lds, stream, a, mp);
  return LP_OK;
}

template <int C, int GM, int OCC>
static int launch_bwd(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
  const size_t lds = Lds::BWD_END * sizeof(float);
  // software-pipelined gather in the backward (tuning knob; costs registers)
  static const bool pipe = getenv("LP_MFMA_BWD_PIPE") != nullptr;
  int rc;
  if (pipe) {
    if ((rc = set_lds(renderer_bwd_mfma<C, GM, OCC, true>, lds))) return rc;
    hipLaunchKernelGGL((renderer_bwd_mfma<C, GM, OCC, true>), dim3(n_blocks(a)), dim3(256), lds, stream, a, mp);
  } else {
    if ((rc = set_lds(renderer_bwd_mfma<C, GM, OCC, false>, lds))) return rc;
    hipLaunchKernelGGL((renderer_bwd_mfma<C, GM, OCC, false>), dim3(n_blocks(a)), dim3(256), lds, stream, a, mp);
  }
  return LP_OK;
}

#define LP_DISPATCH_GM(CALL, CV)                                   \
  switch (gm) {                                                    \
    case GM_TRIPLANE: rc = CALL(CV, GM_TRIPLANE); break;           \
    case GM_VOXEL: rc = CALL(CV, GM_VOXEL); break;                 \
    default: rc = CALL(CV, GM_GENERIC); break;                     \
  }

int renderer_forward_mfma(const LpRendererArgs& a, hipStream_t stream) {
  const MfmaParams mp = make_params(a);
  if (n_blocks(a) == 0) return LP_OK;
  const int gm = grid_mode(a);
  int rc;
#define LP_FWD(CV, GMV) launch_fwd<CV, GMV>(a, mp, stream)
  if (a.grid.channels == 16) { LP_DISPATCH_GM(LP_FWD, 16) } else { LP_DISPATCH_GM(LP_FWD, 32) }
#undef LP_FWD
  if (rc) return rc;
  return check_launch("renderer_fwd_mfma");
}

// grad_grid[i] += sum_r replicas[r][i]  (float4 lanes; n is a multiple of 4 because C is)
__global__ void fold_replicas_kernel(float* __restrict__ dst, const float* __restrict__ rep, int64_t n4, int n_rep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 s = reinterpret_cast<float4*>(dst)[i];
  for (int r = 0; r < n_rep; ++r) {
    const float4 v = reinterpret_cast<const float4*>(rep)[(int64_t)r * n4 + i];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  reinterpret_cast<float4*>(dst)[i] = s;
}

int fold_grad_replicas(const LpRendererArgs& a, hipStream_t stream) {
  if (!a.grad_grid || !a.grad_grid_replicas || a.n_grad_replicas <= 0) return LP_OK;
  const int64_t n = a.grid.n_rows * a.grid.channels;
  if (n % 4 != 0) return set_error(LP_EINVAL, "grad replicas need rows*C divisible by 4");
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(fold_replicas_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, a.grad_grid,
                     a.grad_grid_replicas, n4, a.n_grad_replicas);
  return check_launch("fold_replicas_kernel");
}

int renderer_backward_mfma(const LpRendererArgs& a, hipStream_t stream) {
  const MfmaParams mp = make_params(a);
  if (n_blocks(a) == 0) return LP_OK;
  const int gm = grid_mode(a);
  static const bool v1 = getenv("LP_MFMA_BWD_V1") != nullptr;  // first-generation kernel (A/B timing)
  if (!v1) {
    const int rc2 = renderer_backward_mfma2(a, mp, gm, stream);
    return rc2 ? rc2 : fold_grad_replicas(a, stream);
  }
  // waves per SIMD the backward kernel is register-allocated for (tuning knob)
  static const int occ = [] {
    const char* e = getenv("LP_MFMA_BWD_OCC");
    return (e && e[0] == '1') ? 1 : 2;
  }();
  int rc;
#define LP_BWD1(CV, GMV) launch_bwd<CV, GMV, 1>(a, mp, stream)
#define LP_BWD2(CV, GMV) launch_bwd<CV, GMV, 2>(a, mp, stream)
  if (a.grid.channels == 16) {
    if (occ == 1) { LP_DISPATCH_GM(LP_BWD1, 16) } else { LP_DISPATCH_GM(LP_BWD2, 16) }
  } else {
    if (occ == 1) { LP_DISPATCH_GM(LP_BWD1, 32) } else { LP_DISPATCH_GM(LP_BWD2, 32) }
  }
#undef LP_BWD1
#undef LP_BWD2
  if (rc) return rc;
  if ((rc = check_launch("renderer_bwd_mfma"))) return rc;
  return fold_grad_replicas(a, stream);
}

}  // namespace lp
