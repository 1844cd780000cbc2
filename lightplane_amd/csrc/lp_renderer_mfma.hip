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
#include <stdlib.h>

#include "lp_device.h"
#include "lp_host.h"

namespace lp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

LP_DEV constexpr int featq(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

#define LP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// An integer the optimiser must treat as unknown (it is always 0).  Added to LDS offsets inside
// the sample loop it stops LICM from hoisting the ~200 loop-invariant weight / bias reads out of
// the loop (which costs >200 VGPRs and spills); the reads stay ds_read (LDS address space kept).
LP_DEV int opaque_zero() {
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}

constexpr int HID = 32;        // hidden width of the shape family
constexpr int TILE_LD = 33;    // padded row stride of the per-wave transposition tiles
constexpr int W_LD = 33;       // padded row stride of the weight matrices in LDS
constexpr int WAVES = 4;       // waves per workgroup
constexpr int RAYS_PER_WAVE = 32;
constexpr int MAX_INF = 256;   // beyond-far samples tabulated in LDS

// grid-list shape the kernel is specialised for
constexpr int GM_GENERIC = 0;   // run-time loop over the grid-list
constexpr int GM_TRIPLANE = 1;  // exactly three plane grids
constexpr int GM_VOXEL = 2;     // exactly one voxel grid

// float offsets of the parameter blocks inside mlp_params (computed on the host)
struct MfmaParams {
  int64_t w_t1, w_t2, b_t1, b_t2;  // trunk
  int64_t w_o1, w_o2, b_o1, b_o2;  // opacity
  int64_t w_c1, w_c2, b_c1, b_c2;  // colour
  int ldc2;                        // row stride of w_c2 (padded colour width)
};

// LDS map (floats).  Weight matrices are kept ONCE, row-major [in][W_LD] with a padded row
// stride of 33: the forward operand W[feat(kk,h)][l&31] walks a row (conflict-free), the
// backward operand W[l&31][feat(kk,h)] walks a column with stride 33 (conflict-free as well).
struct Lds {
  static constexpr int WT1 = 0;                  // [32][33] (rows >= C are zero)
  static constexpr int WT2 = WT1 + 32 * W_LD;
  static constexpr int WO1 = WT2 + 32 * W_LD;
  static constexpr int WC1 = WO1 + 32 * W_LD;
  static constexpr int BIAS = WC1 + 32 * W_LD;   // b_t1, b_t2, b_o1, b_c1 : 4 x 32
  static constexpr int WO2 = BIAS + 4 * 32;      // [32]
  static constexpr int WC2 = WO2 + 32;           // [32][4]
  static constexpr int HB = WC2 + 32 * 4;        // bo2, bc2[0..3], pad -> 8
  static constexpr int INF = HB + 8;             // [MAX_INF] depth scale of the beyond-far samples
  static constexpr int FWD_END = INF + MAX_INF;
  // backward only: block-wide dW sum (epilogue), then per-wave scratch
  static constexpr int DW = FWD_END;             // 4 x [32][32]: t1, t2, o1, c1
  static constexpr int WAVE0 = DW + 4 * 1024;
  static constexpr int TX = 0;                   // per-wave: two transposition tiles [32][33] ...
  static constexpr int TY = 32 * TILE_LD;
  static constexpr int TS = 2 * 32 * TILE_LD;    // ... + [32 rays][8]: dro, drc[0..3] of the current sample
  static constexpr int PER_WAVE = 2 * 32 * TILE_LD + 32 * 8;
  static constexpr int BWD_END = WAVE0 + WAVES * PER_WAVE;
};

template <int C, bool BWD>
LP_DEV void stage_weights(const LpRendererArgs& a, const MfmaParams& mp, float* lds) {
  using M = Lds;
  const float* P = a.mlp_params;
  const int tid = threadIdx.x;
  for (int i = tid; i < 32 * 32; i += 256) {
    const int row = i >> 5, col = i & 31;
    const int d = row * W_LD + col;
    lds[M::WT1 + d] = (row < C) ? P[mp.w_t1 + i] : 0.0f;
    lds[M::WT2 + d] = P[mp.w_t2 + i];
    lds[M::WO1 + d] = P[mp.w_o1 + i];
    lds[M::WC1 + d] = P[mp.w_c1 + i];
    if (BWD) {
      lds[M::DW + i] = 0.0f;
      lds[M::DW + 1024 + i] = 0.0f;
      lds[M::DW + 2048 + i] = 0.0f;
      lds[M::DW + 3072 + i] = 0.0f;
    }
  }
  for (int i = tid; i < 32; i += 256) {
    lds[M::BIAS + i] = P[mp.b_t1 + i];
    lds[M::BIAS + 32 + i] = P[mp.b_t2 + i];
    lds[M::BIAS + 64 + i] = P[mp.b_o1 + i];
    lds[M::BIAS + 96 + i] = P[mp.b_c1 + i];
    lds[M::WO2 + i] = P[mp.w_o2 + i];
#pragma unroll
    for (int c = 0; c < 4; ++c)
      lds[M::WC2 + i * 4 + c] = (c < a.color_chn) ? P[mp.w_c2 + (int64_t)i * mp.ldc2 + c] : 0.0f;
  }
  for (int i = tid; i < MAX_INF; i += 256)
    lds[M::INF + i] = (i < a.march.num_samples_inf) ? inf_scale(i, a.march) : 0.0f;
  if (tid == 0) {
    lds[M::HB + 0] = P[mp.b_o2];
#pragma unroll
    for (int c = 0; c < 4; ++c) lds[M::HB + 1 + c] = (c < a.color_chn) ? P[mp.b_c2 + c] : 0.0f;
  }
}

// bias of layer `which` (0 t1, 1 t2, 2 o1, 3 c1) in accumulator-register order for half h
LP_DEV f32x16 load_bias(const float* lds, int which, int h, int zo) {
  const float4* b = reinterpret_cast<const float4*>(lds + Lds::BIAS + which * 32 + 4 * h + zo);
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = b[2 * j];  // floats 8j + 4h .. +3
    acc[4 * j + 0] = v.x; acc[4 * j + 1] = v.y; acc[4 * j + 2] = v.z; acc[4 * j + 3] = v.w;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------
// grid-list gather: interpolated feature of this lane's ray, channels feat(q,h), q < C/2
// ---------------------------------------------------------------------------------------
template <int C>
LP_DEV void gather_tap(const float* data, int row, float w, int h, float (&x0)[C / 2]) {
  // out-of-range taps carry weight 0 and read row 0: no branch
  const float4* src = reinterpret_cast<const float4*>(data + (int64_t)(row < 0 ? 0 : row) * C + 4 * h);
#pragma unroll
  for (int j = 0; j < C / 8; ++j) {
    const float4 v = src[2 * j];  // channels 8j + 4h .. +3
    x0[4 * j + 0] = fmaf(w, v.x, x0[4 * j + 0]);
    x0[4 * j + 1] = fmaf(w, v.y, x0[4 * j + 1]);
    x0[4 * j + 2] = fmaf(w, v.z, x0[4 * j + 2]);
    x0[4 * j + 3] = fmaf(w, v.w, x0[4 * j + 3]);
  }
}

template <int C, int GM>
LP_DEV void gather_features(const LpRendererArgs& a, const Ray& ray, float x, float y, float z, int h,
                            float (&x0)[C / 2]) {
#pragma unroll
  for (int q = 0; q < C / 2; ++q) x0[q] = 0.0f;
  const float keep = (a.march.mask_out_of_bounds && !point_in_bounds(x, y, z)) ? 0.0f : 1.0f;
  if (GM == GM_TRIPLANE) {
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      Taps t;
      plane_taps<false>(a.grid.grids[g], ray.b, x, y, z, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) gather_tap<C>(a.grid.data, t.row[k], t.w[k] * keep, h, x0);
    }
  } else if (GM == GM_VOXEL) {
    Taps t;
    voxel_taps<false>(a.grid.grids[0], ray.b, x, y, z, t);
#pragma unroll
    for (int k = 0; k < 8; ++k) gather_tap<C>(a.grid.data, t.row[k], t.w[k] * keep, h, x0);
  } else {
    for (int g = 0; g < a.grid.n_grids; ++g) {
      Taps t;
      grid_taps<false>(a.grid.grids[g], ray.b, x, y, z, t);
#pragma unroll
      for (int k = 0; k < 4; ++k) gather_tap<C>(a.grid.data, t.row[k], t.w[k] * keep, h, x0);
      if (t.n == 8) {
#pragma unroll
        for (int k = 4; k < 8; ++k) gather_tap<C>(a.grid.data, t.row[k], t.w[k] * keep, h, x0);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------

// One layer, forward form: acc (pre-loaded with the bias) += sum_kk A(kk) * in[kk] with
// A(kk) = W[feat(kk,h)][l&31].  `w` already points at W + (4h)*W_LD + (l&31) (+ opaque zero).
template <int K>
LP_DEV f32x16 layer(const float* w, const float* in, f32x16 acc) {
#pragma unroll
  for (int kk = 0; kk < K; ++kk) acc = LP_MFMA(w[featq(kk, 0) * W_LD], in[kk], acc);
  return acc;
}
// Backward (dX) form: A(kk) = W[l&31][feat(kk,h)].  `w` points at W + (l&31)*W_LD + 4h.
LP_DEV f32x16 layer_t(const float* w, const float* in, f32x16 acc) {
#pragma unroll
  for (int kk = 0; kk < 16; ++kk) acc = LP_MFMA(w[featq(kk, 0)], in[kk], acc);
  return acc;
}

struct Heads {
  float raw_o;
  float raw_c[4];
};

// opacity / colour output layers on the VALU (N = 1 and N <= 4): each lane covers its 16
// features, the partner lane (l ^ 32) the other 16.
LP_DEV Heads heads_forward(const float* lds_, int h, const float (&ho)[16], const float (&hc)[16], int zo) {
  using M = Lds;
  const float* lds = lds_ + zo;
  float po = 0.0f, pc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 wo = *reinterpret_cast<const float4*>(lds + M::WO2 + 8 * j + 4 * h);
    const float wov[4] = {wo.x, wo.y, wo.z, wo.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 4 * j + i;
      po = fmaf(ho[q], wov[i], po);
      const float4 wc = *reinterpret_cast<const float4*>(lds + M::WC2 + (8 * j + 4 * h + i) * 4);
      pc[0] = fmaf(hc[q], wc.x, pc[0]);
      pc[1] = fmaf(hc[q], wc.y, pc[1]);
      pc[2] = fmaf(hc[q], wc.z, pc[2]);
      pc[3] = fmaf(hc[q], wc.w, pc[3]);
    }
  }
  Heads o;
  o.raw_o = (po + __shfl_xor(po, 32)) + lds[M::HB];
#pragma unroll
  for (int c = 0; c < 4; ++c) o.raw_c[c] = (pc[c] + __shfl_xor(pc[c], 32)) + lds[M::HB + 1 + c];
  return o;
}

// Activations of one sample (accumulator-register order).
template <int C>
struct Act {
  float x0[C / 2];
  float h1[16], e[16], ho[16], hc[16];
};

// Nothing may be scheduled across this point.  Used to cut the sample loop body into groups of
// "one layer's MFMAs + one plane's gather": inside a group the scheduler interleaves freely, but it
// can no longer hoist all 24 dwordx4 loads of a sample to the top (96 live VGPRs).
#define LP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)

// Ask the scheduler for the issue order "1 MFMA, a few VALU, (1 global load), (1 LDS read)" N times:
// a dependent v_mfma_f32_32x32x2_f32 chain stalls its wave 64 cycles per link (in-order issue), so
// every instruction placed between two links is free.
template <int N, int VALU_PER, int VMEM_EVERY>
LP_DEV void interleave_hint() {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                       // MFMA
    if (VMEM_EVERY > 0 && (i % VMEM_EVERY) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                       // DS read (next operand)
    __builtin_amdgcn_sched_group_barrier(0x002, VALU_PER, 0);                // VALU
  }
}

LP_DEV void load_encoding(const LpRendererArgs& a, int64_t rid, int h, float (&enc)[16]) {
  const float4* src = reinterpret_cast<const float4*>(a.rays.encoding + rid * HID + 4 * h);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = src[2 * j];
    enc[4 * j + 0] = v.x; enc[4 * j + 1] = v.y; enc[4 * j + 2] = v.z; enc[4 * j + 3] = v.w;
  }
}

// geometry of one sample + its (prefetched) grid feature
template <int C>
struct Sample {
  float depth, occ, x, y, z;
  float x0[C / 2];
};

template <int C>
LP_DEV void sample_geometry(const LpRendererArgs& a, const float* lds, const Ray& ray, int s, Sample<C>& o) {
  o.depth = sample_depth_tab(s, a.march, ray.near_t, ray.far_t, lds + Lds::INF);
  sample_point(ray, o.depth, a.march.contract_coords != 0, o.x, o.y, o.z);
  o.occ = 1.0f;
  if (a.scaffold) o.occ = scaffold_lookup(a.scaffold, a.scaffold_shape, ray.b, o.x, o.y, o.z);
}

template <int C, int GM>
LP_DEV void fetch_sample(const LpRendererArgs& a, const float* lds, const Ray& ray, int s, int h, Sample<C>& o) {
  sample_geometry<C>(a, lds, ray, s, o);
  gather_features<C, GM>(a, ray, o.x, o.y, o.z, h, o.x0);
}

// Decoder of the CURRENT sample (input t.x0; fills t.h1 / t.e / t.ho / t.hc) interleaved with the
// gather of sample `s_next` into `nx` (software pipeline).  Triplane: plane g is gathered next to
// hidden layer g+1; voxel: taps 0-3 / 4-7 next to layers 2 / 3; generic grid-lists: whole gather
// first.
template <int C, int GM_, bool PREFETCH = true>
LP_DEV Heads decode_prefetch(const LpRendererArgs& a, const float* lds, const Ray& ray, int lane,
                             const float (&enc)[16], Act<C>& t, int s_next, Sample<C>& nx, int zo) {
  using M = Lds;
  // without prefetch this is the plain decoder (GM = -1 disables every gather below)
  constexpr int GM = PREFETCH ? GM_ : -1;
  const int h = lane >> 5;
  // operand base of this lane; `zo` (always 0) keeps the weight reads inside the sample loop
  const float* wl = lds + (4 * h) * W_LD + (lane & 31) + zo;
  float keep = 1.0f;
  if (PREFETCH) {
    sample_geometry<C>(a, lds, ray, s_next, nx);
    keep = (a.march.mask_out_of_bounds && !point_in_bounds(nx.x, nx.y, nx.z)) ? 0.0f : 1.0f;
  }
  Taps tp[GM == GM_TRIPLANE ? 3 : 1];
  if (GM == GM_TRIPLANE) {
#pragma unroll
    for (int g = 0; g < 3; ++g) plane_taps<false>(a.grid.grids[g], ray.b, nx.x, nx.y, nx.z, tp[g]);
  } else if (GM == GM_VOXEL) {
    voxel_taps<false>(a.grid.grids[0], ray.b, nx.x, nx.y, nx.z, tp[0]);
  } else if (GM == GM_GENERIC) {
    gather_features<C, GM_GENERIC>(a, ray, nx.x, nx.y, nx.z, h, nx.x0);
  }
  if (GM == GM_TRIPLANE || GM == GM_VOXEL) {
#pragma unroll
    for (int q = 0; q < C / 2; ++q) nx.x0[q] = 0.0f;
  }
  LP_SCHED_FENCE();
  f32x16 acc = layer<C / 2>(wl + M::WT1, t.x0, load_bias(lds, 0, h, zo));
#pragma unroll
  for (int q = 0; q < 16; ++q) t.h1[q] = fmaxf(acc[q], 0.0f);
  LP_SCHED_FENCE();
  // ---- group 1: trunk layer 2  ||  plane 0 / voxel taps 0-3 ----
  if (GM == GM_TRIPLANE || GM == GM_VOXEL) {
#pragma unroll
    for (int k = 0; k < 4; ++k) gather_tap<C>(a.grid.data, tp[0].row[k], tp[0].w[k] * keep, h, nx.x0);
  }
  acc = layer<16>(wl + M::WT2, t.h1, load_bias(lds, 1, h, zo));
#pragma unroll
  for (int q = 0; q < 16; ++q) t.e[q] = fmaxf(acc[q], 0.0f);
  if (GM == GM_TRIPLANE || GM == GM_VOXEL) interleave_hint<16, 5, 2>();
  LP_SCHED_FENCE();
  // ---- group 2: opacity hidden layer  ||  plane 1 / voxel taps 4-7 ----
  if (GM == GM_TRIPLANE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) gather_tap<C>(a.grid.data, tp[1].row[k], tp[1].w[k] * keep, h, nx.x0);
  } else if (GM == GM_VOXEL) {
#pragma unroll
    for (int k = 4; k < 8; ++k) gather_tap<C>(a.grid.data, tp[0].row[k], tp[0].w[k] * keep, h, nx.x0);
  }
  acc = layer<16>(wl + M::WO1, t.e, load_bias(lds, 2, h, zo));
#pragma unroll
  for (int q = 0; q < 16; ++q) t.ho[q] = fmaxf(acc[q], 0.0f);
  if (GM == GM_TRIPLANE || GM == GM_VOXEL) interleave_hint<16, 5, 2>();
  LP_SCHED_FENCE();
  // ---- group 3: colour hidden layer  ||  plane 2 ----
  if (GM == GM_TRIPLANE) {
#pragma unroll
    for (int k = 0; k < 4; ++k) gather_tap<C>(a.grid.data, tp[2].row[k], tp[2].w[k] * keep, h, nx.x0);
  }
  float ein[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) ein[q] = t.e[q] + enc[q];
  acc = layer<16>(wl + M::WC1, ein, load_bias(lds, 3, h, zo));
#pragma unroll
  for (int q = 0; q < 16; ++q) t.hc[q] = fmaxf(acc[q], 0.0f);
  if (GM == GM_TRIPLANE) interleave_hint<16, 6, 2>();
  LP_SCHED_FENCE();
  return heads_forward(lds, h, t.ho, t.hc, zo);
}

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

// Row-contiguous, run-length merged scatter of one grid's taps (table already in LDS).
// Tap slot k of all 32 rays is walked in ray order by one group of C lanes (lane = channel);
// neighbouring rays mostly fall into the same cell: contributions to the same row are summed in a
// register and leave as ONE row-contiguous atomic per run.
template <int C>
LP_DEV void scatter_taps(float* grad, const float2* tab, const float* tx, int n_taps, int lane) {
  constexpr int GRPS = 64 / C;  // rows (= taps) handled per instruction
  const int sub = lane % C, grp = lane / C;
  for (int k0 = 0; k0 < n_taps; k0 += GRPS) {
    const int k = k0 + grp;
    int cur = -1;
    float run = 0.0f;
#pragma unroll 4
    for (int rr = 0; rr < 32; ++rr) {
      const float2 e = tab[rr * 8 + k];
      const int row = __float_as_int(e.x);
      if (row != cur) {
        if (cur >= 0) atomic_add_f32(grad + (int64_t)cur * C + sub, run);
        cur = row;
        run = 0.0f;
      }
      run = fmaf(e.y, tx[rr * TILE_LD + sub], run);
    }
    if (cur >= 0) atomic_add_f32(grad + (int64_t)cur * C + sub, run);
  }
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
      // ---- grid gradient: transpose dx0 through LDS (tile tx), taps through the table in ty ----
#pragma unroll
      for (int q = 0; q < C / 2; ++q) tx[r * TILE_LD + featq(q, h)] = acc[q];
      const bool live = valid && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));
      float2* tab = reinterpret_cast<float2*>(ty);  // [32 rays][8 taps] {row bits, weight}
      const int ng = (GM == GM_TRIPLANE) ? 3 : (GM == GM_VOXEL) ? 1 : a.grid.n_grids;
      for (int g = 0; g < ng; ++g) {
        Taps tp;
        if (GM == GM_TRIPLANE) plane_taps<false>(a.grid.grids[g], ray.b, x, y, z, tp);
        else if (GM == GM_VOXEL) voxel_taps<false>(a.grid.grids[g], ray.b, x, y, z, tp);
        else grid_taps<false>(a.grid.grids[g], ray.b, x, y, z, tp);
        if (h == 0) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const bool ok = live && tp.row[k] >= 0;
            tab[r * 8 + k] = make_float2(__int_as_float(ok ? tp.row[k] : -1), ok ? tp.w[k] : 0.0f);
          }
        }
        scatter_taps<C>(a.grad_grid, tab, tx, (GM == GM_TRIPLANE) ? 4 : (GM == GM_VOXEL) ? 8 : tp.n, lane);
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

int renderer_backward_mfma(const LpRendererArgs& a, hipStream_t stream) {
  const MfmaParams mp = make_params(a);
  if (n_blocks(a) == 0) return LP_OK;
  const int gm = grid_mode(a);
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
  return check_launch("renderer_bwd_mfma");
}

}  // namespace lp
