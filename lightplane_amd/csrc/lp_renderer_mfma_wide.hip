// lp_renderer_mfma_wide.hip -- MFMA Renderer kernels for hidden width 64 (gfx950).
//
// Shape family: single grid-list with C in {16, 32} channels, trunk [C,64,64], opacity [64,64,1],
// colour [64,64,>=Cc] with Cc <= 4 -- e.g. the reference's own example configuration
// (examples/config/synthetic_overfit.json: triplane x 32 channels, mlp_hidden_chn 64).
//
// Same scheme as the width-32 kernels (lp_renderer_mfma.hip / lp_renderer_mfma_bwd.hip): one wave =
// 32 rays, lane (h = l>>5, r = l&31) owns ray r; a 64-wide activation is two blocks of 32 features
// and the lane keeps features 32*blk + feat(q,h) of both (32 registers).  A layer is, per output
// block, one chain of v_mfma_f32_32x32x2_f32 over all input blocks; the accumulator registers of a
// layer are the B operands of the next one.  Backward: far -> near recompute; the weight gradients
// are shared by the workgroup -- every wave publishes its X / dY tiles (feature-major, [64][36]) and
// wave w accumulates the 16-column slab w of every layer's dW over all 128 rays with
// v_mfma_f32_16x16x4_f32 (up to 4 quadrants = 16 accumulator registers per layer).  The register
// footprint (~320) allows one wave per SIMD; the kernels are matrix-pipe heavy (4x the FLOPs of the
// width-32 family), so that costs less than it would there.
#include "lp_mfma_common.h"

namespace lp {

typedef float f32x4w __attribute__((ext_vector_type(4)));
#define LP_MFMA16W(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int TW_LD = 36;  // row stride of the feature-major tiles [features][32 rays + 4]

template <int NB>
struct LdsW {
  static constexpr int H = 32 * NB;
  static constexpr int WLD = H + 1;              // padded row stride of the weight matrices
  static constexpr int WT1 = 0;                  // [32][WLD]  (rows >= C are zero)
  static constexpr int WT2 = WT1 + 32 * WLD;     // [H][WLD]
  static constexpr int WO1 = WT2 + H * WLD;
  static constexpr int WC1 = WO1 + H * WLD;
  static constexpr int BIAS = WC1 + H * WLD;     // b_t1, b_t2, b_o1, b_c1 : 4 x H
  static constexpr int WO2 = BIAS + 4 * H;       // [H]
  static constexpr int WC2 = WO2 + H;            // [H][4]
  static constexpr int HB = WC2 + 4 * H;         // bo2, bc2[0..3], pad -> 8
  static constexpr int INF = HB + 8;             // [MAX_INF]
  static constexpr int FWD_END = INF + MAX_INF;
  // backward: per-wave area
  static constexpr int XT = 0;                   // X  tile [H][36] (also the dx0 tile of the scatter)
  static constexpr int YT = H * TW_LD;           // dY tile [H][36] (also the scatter's weight table)
  static constexpr int TS = 2 * H * TW_LD;       // [5][32]: d raw_o, d raw_c[0..3] by ray
  static constexpr int PER_WAVE = TS + 5 * 32;
  static constexpr int BWD_END = FWD_END + WAVES * PER_WAVE;
};
static_assert(LdsW<2>::BWD_END * 4 <= 160 * 1024, "backward LDS must fit one CU");
static_assert(LdsW<2>::BIAS % 4 == 0 && LdsW<2>::WC2 % 4 == 0 && LdsW<2>::FWD_END % 4 == 0 && LdsW<2>::PER_WAVE % 4 == 0,
              "16-byte alignment of the float4 regions");

template <int C, int NB>
LP_DEV void stage_weights_w(const LpRendererArgs& a, const MfmaParams& mp, float* lds) {
  using M = LdsW<NB>;
  constexpr int H = M::H;
  const float* P = a.mlp_params;
  const int tid = threadIdx.x;
  for (int i = tid; i < 32 * H; i += 256) {
    const int row = i / H, col = i % H;
    lds[M::WT1 + row * M::WLD + col] = (row < C) ? P[mp.w_t1 + i] : 0.0f;
  }
  for (int i = tid; i < H * H; i += 256) {
    const int row = i / H, col = i % H;
    const int d = row * M::WLD + col;
    lds[M::WT2 + d] = P[mp.w_t2 + i];
    lds[M::WO1 + d] = P[mp.w_o1 + i];
    lds[M::WC1 + d] = P[mp.w_c1 + i];
  }
  for (int i = tid; i < H; i += 256) {
    lds[M::BIAS + i] = P[mp.b_t1 + i];
    lds[M::BIAS + H + i] = P[mp.b_t2 + i];
    lds[M::BIAS + 2 * H + i] = P[mp.b_o1 + i];
    lds[M::BIAS + 3 * H + i] = P[mp.b_c1 + i];
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

// One layer, forward form.  in: NBI blocks of 16 registers, the LAST block has KL k-steps (trunk
// layer 1: C/2, else 16).  w: matrix + (4h)*WLD + (l&31) (+ opaque zero); bias: vector + 4h (+ zero).
template <int NB, int NBI, int KL>
LP_DEV void layer_w(const float* w, const float* bias, const float* in, float* out, bool relu) {
  constexpr int WLD = LdsW<NB>::WLD;
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
    f32x16 acc;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = *reinterpret_cast<const float4*>(bias + 32 * ob + 8 * j);
      acc[4 * j + 0] = v.x; acc[4 * j + 1] = v.y; acc[4 * j + 2] = v.z; acc[4 * j + 3] = v.w;
    }
#pragma unroll
    for (int ib = 0; ib < NBI; ++ib) {
#pragma unroll
      for (int kk = 0; kk < ((ib == NBI - 1) ? KL : 16); ++kk)
        acc = LP_MFMA(w[(32 * ib + featq(kk, 0)) * WLD + 32 * ob], in[16 * ib + kk], acc);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) out[16 * ob + q] = relu ? relu_f(acc[q]) : acc[q];
  }
}

// Backward (dX) form: dx[ib] (+)= sum_ob W[32 ib + (l&31)][32 ob + feat(kk,h)] * dy[ob][kk].
// w: matrix + (l&31)*WLD + 4h (+ opaque zero).  NBI input blocks (= output of this function).
template <int NB, int NBI, bool ACCUM>
LP_DEV void layer_t_w(const float* w, const float* dy, float* dx) {
  constexpr int WLD = LdsW<NB>::WLD;
#pragma unroll
  for (int ib = 0; ib < NBI; ++ib) {
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = ACCUM ? dx[16 * ib + q] : 0.0f;
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) acc = LP_MFMA(w[32 * ib * WLD + 32 * ob + featq(kk, 0)], dy[16 * ob + kk], acc);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) dx[16 * ib + q] = acc[q];
  }
}

// opacity / colour output layers on the VALU
template <int NB, int NC = 4>
LP_DEV Heads heads_forward_w(const float* lds, int h, const float* ho, const float* hc) {
  using M = LdsW<NB>;
  float po = 0.0f, pc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 wo = *reinterpret_cast<const float4*>(lds + M::WO2 + 32 * b + 8 * j + 4 * h);
      const float wov[4] = {wo.x, wo.y, wo.z, wo.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = 16 * b + 4 * j + i;
        po = fmaf(ho[q], wov[i], po);
        const float4 wc = *reinterpret_cast<const float4*>(lds + M::WC2 + (32 * b + 8 * j + 4 * h + i) * 4);
        pc[0] = fmaf(hc[q], wc.x, pc[0]);
        pc[1] = fmaf(hc[q], wc.y, pc[1]);
        pc[2] = fmaf(hc[q], wc.z, pc[2]);
        if (NC > 3) pc[3] = fmaf(hc[q], wc.w, pc[3]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  Heads o;
  o.raw_o = (po + __shfl_xor(po, 32)) + lds[M::HB];
#pragma unroll
  for (int c = 0; c < 4; ++c) o.raw_c[c] = (c < NC) ? (pc[c] + __shfl_xor(pc[c], 32)) + lds[M::HB + 1 + c] : 0.0f;
  return o;
}

template <int NB>
LP_DEV void load_encoding_w(const LpRendererArgs& a, int64_t rid, int h, float* enc) {
  constexpr int H = 32 * NB;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const float4* src = reinterpret_cast<const float4*>(a.rays.encoding + rid * H + 32 * b + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = src[2 * j];
      enc[16 * b + 4 * j + 0] = v.x; enc[16 * b + 4 * j + 1] = v.y;
      enc[16 * b + 4 * j + 2] = v.z; enc[16 * b + 4 * j + 3] = v.w;
    }
  }
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <int C, int GM, int NB>
__global__ void __launch_bounds__(256, 2) renderer_fwd_mfma_w(const LpRendererArgs a, const MfmaParams mp) {
  using M = LdsW<NB>;
  constexpr int H = M::H;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights_w<C, NB>(a, mp, lds);
  __syncthreads();
  const float* lds_inf = lds + (M::INF - Lds::INF);  // sample_geometry() reads its table at Lds::INF
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  // small batch (mp.seg_fwd, see renderer_fwd_bf3 SEGF in lp_renderer_mfma.hip): a workgroup marches one segment from
  // transmittance 1 and leaves segment-local state records for renderer_fwd_combine
  const bool segf = mp.seg_fwd != 0;
  const int seg_len = LP_SEG_LEN * mp.seg_blocks;
  const int n_seg = segf ? (a.march.num_samples + seg_len - 1) / seg_len : 1;
  const int blk = segf ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
  const int seg = segf ? (int)blockIdx.x - blk * n_seg : 0;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[16 * NB];
  load_encoding_w<NB>(a, rid, h, enc);
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_ckpt = ckpt_count(a.march);
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;
  float nlt = 0.0f, nlt_lo = 0.0f, t_prev = 1.0f, len = 0.0f, depth_prev = 0.0f;
  int s_last = s_tot - 1;  // last sample marched
  float facc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const int s_lo = segf ? seg * seg_len : 0;
  const int s_hi = segf ? ((s_lo + seg_len < s_tot) ? s_lo + seg_len : s_tot) : s_tot;
  if (segf && s_lo > 0) {  // interval length of the segment's first sample
    Sample<C> pv;
    sample_geometry<C>(a, lds_inf, ray, s_lo - 1, pv);
    depth_prev = pv.depth;
  }
  for (int s = s_lo; s < s_hi; ++s) {
    Sample<C> nx;
    fetch_sample<C, GM, true>(a, lds_inf, ray, s, h, nx);
    const float depth = nx.depth, occ = nx.occ;
    const int zo = opaque_zero();
    const float* wl = lds + (4 * h) * M::WLD + r + zo;
    const float* bl = lds + M::BIAS + 4 * h + zo;
    float h1[16 * NB], e[16 * NB], ho[16 * NB];
    LP_SCHED_FENCE();
    layer_w<NB, 1, C / 2>(wl + M::WT1, bl, nx.x0, h1, true);
    LP_SCHED_FENCE();
    layer_w<NB, NB, 16>(wl + M::WT2, bl + H, h1, e, true);
    LP_SCHED_FENCE();
    layer_w<NB, NB, 16>(wl + M::WO1, bl + 2 * H, e, ho, true);
    LP_SCHED_FENCE();
#pragma unroll
    for (int q = 0; q < 16 * NB; ++q) e[q] += enc[q];
    layer_w<NB, NB, 16>(wl + M::WC1, bl + 3 * H, e, h1, true);  // h1 := hc
    LP_SCHED_FENCE();
    const Heads hd = heads_forward_w<NB>(lds + zo, h, ho, h1);
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    depth_prev = depth;
    float raw = hd.raw_o;
    if (a.noise_sigma > 0.0f) raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    nlt_add(nlt, nlt_lo, opacity * delta);
    if (!segf && a.neg_log_t_ckpt && valid && h == 0) {
      const int ck = ckpt_index(s, a.march);
      if (ck >= 0) *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + ck) * 2) = make_float2(nlt, nlt_lo);
    }
    const float tr = __expf(-nlt);
    const float w = t_prev - tr;
    t_prev = tr;
    len = fmaf(w, depth, len);
#pragma unroll
    for (int c = 0; c < 4; ++c) facc[c] = fmaf(w, sigmoid_f(hd.raw_c[c]) * occ, facc[c]);
    // state records of the segment-parallel backward (absolute; segf: relative to the segment's start)
    if (a.seg_prefix && valid && h == 0 && (((s + 1) % LP_SEG_LEN) == 0 || s == a.march.num_samples - 1)) {
      float4* dst = reinterpret_cast<float4*>(a.seg_prefix + (ray_id * segment_count(a.march) + s / LP_SEG_LEN) * 8);
      dst[0] = make_float4(len, facc[0], facc[1], facc[2]);
      dst[1] = make_float4(facc[3], nlt, nlt_lo, 0.0f);
    }
    if (a.stop_neg_log_t > 0.0f && __ballot(valid && nlt < a.stop_neg_log_t) == 0) {  // early termination
      s_last = s;
      break;
    }
  }
  if (!segf && valid && h == 0) {
    write_ray_outputs(a, ray_id, len, nlt, facc);
    if (a.neg_log_t_ckpt)
      *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + n_ckpt - 1) * 2) = make_float2((float)s_last, nlt_lo);
  }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------

LP_DEV constexpr int pi16w(int m) { return m < 4 ? 2 * m : (m < 12 ? 2 * (m - 4) + 1 : 2 * (m - 8)); }
LP_DEV void lds_barrier_w() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NB>
LP_DEV void tile_store_w(float* tile, int r, int h, const float* v) {
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int q = 0; q < 16; ++q) tile[(32 * b + featq(q, h)) * TW_LD + r] = v[16 * b + q];
}

// dW slab of one layer: this wave owns the 16 output columns `b_off` points at and NQ row quadrants
// (16 input features each); acc[mi] += X^T dY over the 128 rays of the workgroup.
template <int NQ, int PER_WAVE>
LP_DEV void dw_slab(const float* wave0, int a_off, int b_off, f32x4w (&acc)[NQ], float& db) {
  float s = 0.0f;
#pragma unroll 1
  for (int v = 0; v < WAVES; ++v) {
    const float* base = wave0 + v * PER_WAVE;
    const float4 b0 = *reinterpret_cast<const float4*>(base + b_off);
    const float4 b1 = *reinterpret_cast<const float4*>(base + b_off + 4);
    float4 a0[NQ], a1[NQ];
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) {
      a0[mi] = *reinterpret_cast<const float4*>(base + a_off + 16 * mi * TW_LD);
      a1[mi] = *reinterpret_cast<const float4*>(base + a_off + 16 * mi * TW_LD + 4);
    }
    // the NQ accumulators are independent: interleave them so that no MFMA waits for its predecessor
    // (v_mfma_f32_16x16x4_f32: 32 cycles issue, 40 cycles dependent)
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a0[mi].x, b0.x, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a0[mi].y, b0.y, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a0[mi].z, b0.z, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a0[mi].w, b0.w, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a1[mi].x, b1.x, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a1[mi].y, b1.y, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a1[mi].z, b1.z, acc[mi]);
#pragma unroll
    for (int mi = 0; mi < NQ; ++mi) acc[mi] = LP_MFMA16W(a1[mi].w, b1.w, acc[mi]);
    s += ((b0.x + b0.y) + (b0.z + b0.w)) + ((b1.x + b1.y) + (b1.z + b1.w));
  }
  db += s;
}

// NC = 3: RGB, the padding column of the colour path is compiled out
template <int C, int GM, int NB, bool PLAIN, int NC = 4>
__global__ void __launch_bounds__(256, 1) renderer_bwd_mfma_w(const LpRendererArgs a, const MfmaParams mp) {
  static_assert(NB == 2, "the dW slab assignment (wave w <-> output columns 16w..) assumes 64 output features");
  using M = LdsW<NB>;
  constexpr int H = M::H;
  constexpr int NQ1 = C / 16;  // row quadrants of trunk layer 1
  constexpr int NQH = H / 16;  // row quadrants of the 64-input layers
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights_w<C, NB>(a, mp, lds);
  const float* lds_inf = lds + (M::INF - Lds::INF);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  float* const wave0 = lds + M::FWD_END;
  float* const wv = wave0 + wave * M::PER_WAVE;
  float* const xt = wv + M::XT;
  float* const yt = wv + M::YT;
  float* const ts = wv + M::TS;

  // segment-parallel sweep of a small batch (LpRendererArgs.seg_prefix, see renderer_bwd_bf3): run-time switch
  const bool seg_on = a.seg_prefix != nullptr;
  const int n_rec = seg_on ? segment_count(a.march) : 1;
  const int seg_len = LP_SEG_LEN * mp.seg_blocks;
  const int n_seg = seg_on ? (a.march.num_samples + seg_len - 1) / seg_len : 1;
  const int blk = seg_on ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
  const int seg = seg_on ? (int)blockIdx.x - blk * n_seg : 0;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[16 * NB], denc[16 * NB];
  load_encoding_w<NB>(a, rid, h, enc);
#pragma unroll
  for (int q = 0; q < 16 * NB; ++q) denc[q] = 0.0f;
  // closing pair of the checkpoint list (see renderer_bwd_bf3, lp_renderer_mfma_bwd.hip): workgroup-uniform first sample of the loop
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_ckpt = ckpt_count(a.march);
  int s_last_w = s_tot - 1;
  float nlt_lo = 0.0f;
  if (a.neg_log_t_ckpt) {
    const float2 e2 = *reinterpret_cast<const float2*>(a.neg_log_t_ckpt + (rid * n_ckpt + n_ckpt - 1) * 2);
    s_last_w = __builtin_amdgcn_readfirstlane((int)e2.x);
    s_last_w = s_last_w < 0 ? 0 : (s_last_w > s_tot - 1 ? s_tot - 1 : s_last_w);
    nlt_lo = e2.y;
  }
  int s_begin = s_tot - 1;
  if (PLAIN) {  // the PLAIN instantiation is only launched without early termination
    __syncthreads();
  } else {
    if (lane == 0) ts[0] = (float)s_last_w;
    __syncthreads();
    s_begin = 0;
#pragma unroll
    for (int v = 0; v < WAVES; ++v) {
      const int sv = (int)wave0[v * M::PER_WAVE + M::TS];
      s_begin = sv > s_begin ? sv : s_begin;
    }
    __syncthreads();  // ts[] is reused by the sample loop
  }
  const int s_lo = seg_on ? seg * seg_len : 0;
  if (seg_on) s_begin = (s_lo + seg_len - 1 < s_tot - 1) ? s_lo + seg_len - 1 : s_tot - 1;
  float gfeat[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    gfeat[c] = (valid && a.grad_feature && c < a.color_chn) ? a.grad_feature[rid * a.color_chn + c] : 0.0f;
  const float g_len = (valid && a.grad_ray_length) ? a.grad_ray_length[rid] : 0.0f;
  const float g_nlt = epilogue_grad_nlt(a, rid, valid, a.neg_log_t[rid],
                                        (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f, gfeat, 4);

  const bool want_params = a.grad_mlp_params != nullptr;
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;

  // dW slab of this wave: output columns 16*wave .. of every layer; MFMA lane (m16, ka)
  const int m16 = lane & 15, ka = lane >> 4;
  const int a_off = M::XT + pi16w(m16) * TW_LD + 8 * ka;
  const int b_off = M::YT + (16 * wave + pi16w(m16)) * TW_LD + 8 * ka;
  f32x4w dq_t1[NQ1], dq_t2[NQH], dq_o1[NQH], dq_c1[NQH];
#pragma unroll
  for (int i = 0; i < NQ1; ++i) dq_t1[i] = (f32x4w){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NQH; ++i) dq_t2[i] = dq_o1[i] = dq_c1[i] = (f32x4w){0, 0, 0, 0};
  float db_t1 = 0.0f, db_t2 = 0.0f, db_o1 = 0.0f, db_c1 = 0.0f;
  // output layers of the heads: lane (f = l&31, half h) owns features f + 32 b, partial over 16 rays
  float dwo2[NB], dwc2[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    dwo2[b] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dwc2[b][c] = 0.0f;
  }
  float dbo2 = 0.0f, dbc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};

  const bool gg = a.grad_grid_list[0] != nullptr;  // the host fills every entry or none

  float nlt = a.neg_log_t[rid];
  float suffix = 0.0f, p_next = 0.0f;
  if (seg_on) {  // start of a segment: -log T and the sums behind its last sample, from the forward's state records
    const float4* pj = reinterpret_cast<const float4*>(a.seg_prefix + (rid * n_rec + s_begin / LP_SEG_LEN) * 8);
    const float4* pt = reinterpret_cast<const float4*>(a.seg_prefix + (rid * n_rec + n_rec - 1) * 8);
    const float4 j0 = pj[0], j1 = pj[1], t0 = pt[0];
    nlt = j1.y;
    nlt_lo = j1.z;
    if (seg < n_seg - 1) {
      float rest = g_len * (t0.x - j0.x);
      rest = fmaf(gfeat[0], t0.y - j0.y, rest);
      rest = fmaf(gfeat[1], t0.z - j0.z, rest);
      rest = fmaf(gfeat[2], t0.w - j0.w, rest);
      rest = fmaf(gfeat[3], pt[1].x - j1.x, rest);
      suffix = -rest;
    }
  }
  Sample<C> nx;
  fetch_sample<C, GM, true, PLAIN>(a, lds_inf, ray, s_begin, h, nx);
  for (int s = s_begin; s >= s_lo; --s) {
    const bool on = PLAIN || s <= s_last_w;  // wave-uniform; false only for samples a sibling wave still marches
    const float depth = nx.depth, occ = nx.occ, x = nx.x, y = nx.y, z = nx.z;
    float x0[C / 2];
#pragma unroll
    for (int q = 0; q < C / 2; ++q) x0[q] = nx.x0[q];
    const int zo = opaque_zero();
    const float* ldz = lds + zo;
    const float* wl = lds + (4 * h) * M::WLD + r + zo;
    const float* bl = lds + M::BIAS + 4 * h + zo;
    const float* wt = lds + r * M::WLD + 4 * h + zo;

    // ---------------- forward recompute ----------------
    float h1[16 * NB], e[16 * NB], ho[16 * NB], hc[16 * NB];
    layer_w<NB, 1, C / 2>(wl + M::WT1, bl, x0, h1, true);
    LP_SCHED_FENCE();
    layer_w<NB, NB, 16>(wl + M::WT2, bl + H, h1, e, true);
    LP_SCHED_FENCE();
    layer_w<NB, NB, 16>(wl + M::WO1, bl + 2 * H, e, ho, true);
    LP_SCHED_FENCE();
    {
      float ein[16 * NB];
#pragma unroll
      for (int q = 0; q < 16 * NB; ++q) ein[q] = e[q] + enc[q];
      layer_w<NB, NB, 16>(wl + M::WC1, bl + 3 * H, ein, hc, true);
    }
    LP_SCHED_FENCE();
    const Heads hd = heads_forward_w<NB, NC>(ldz, h, ho, hc);
    LP_SCHED_FENCE();
    if (want_params) {
      tile_store_w<NB>(xt, r, h, ho);
      tile_store_w<NB>(yt, r, h, hc);
    }
    LP_SCHED_FENCE();

    // ---------------- compositing, backward ----------------
    const float depth_prev =
        PLAIN ? ray.near_t + lin01((s > 0) ? s - 1 : 0, a.march.num_samples) * (ray.far_t - ray.near_t)
              : sample_depth_tab((s > 0) ? s - 1 : 0, a.march, ray.near_t, ray.far_t, lds + M::INF);
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    float raw = hd.raw_o;
    if (!PLAIN && a.noise_sigma > 0.0f)
      raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    if (on && a.neg_log_t_ckpt) {
      const int ck = PLAIN ? ((((s + 1) % LP_NLT_CKPT) == 0 || s == s_tot - 1) ? s / LP_NLT_CKPT : -1)
                           : ckpt_index(s, a.march);
      if (ck >= 0) {
        const float2 c2 = *reinterpret_cast<const float2*>(a.neg_log_t_ckpt + (rid * n_ckpt + ck) * 2);
        nlt = c2.x;
        nlt_lo = c2.y;
      }
    }
    const float t_i = __expf(-nlt);
    nlt_add(nlt, nlt_lo, on ? -(opacity * delta) : 0.0f);
    if (!(nlt > 0.0f)) { nlt = 0.0f; nlt_lo = 0.0f; }
    const float t_im1 = __expf(-nlt);
    const float w = t_im1 - t_i;
    float sg[4];
    float p_i = g_len * depth;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sg[c] = (c < NC) ? sigmoid_f(hd.raw_c[c]) : 0.0f;
      if (c < NC) p_i = fmaf(gfeat[c], sg[c] * occ, p_i);
    }
    suffix = on ? fmaf(t_i, p_i - p_next, suffix) : suffix;
    p_next = on ? p_i : p_next;
    const float d_a = suffix + g_nlt;
    const bool contrib = valid && on;
    const float dro = contrib ? d_a * delta * a.gain * occ * d_softplus_f(raw) : 0.0f;
    float drc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) drc[c] = (c < NC && contrib) ? w * gfeat[c] * occ * sg[c] * (1.0f - sg[c]) : 0.0f;

    // ---------------- output layers of the heads (VALU) ----------------
    // d ho is formed where it is needed (opacity hidden layer): keep only the ReLU mask of ho
    unsigned ho_mask = 0;
#pragma unroll
    for (int q = 0; q < 16 * NB; ++q) ho_mask |= (ho[q] > 0.0f) ? (1u << q) : 0u;
    float dhc[16 * NB];
    {
      const float* wc2 = lds + M::WC2 + 16 * h + opaque_zero();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int q = 16 * b + 4 * j + i;
            const float4 wc = *reinterpret_cast<const float4*>(wc2 + (32 * b + 8 * j + i) * 4);
            float v = drc[0] * wc.x;
            v = fmaf(drc[1], wc.y, v);
            v = fmaf(drc[2], wc.z, v);
            if (NC > 3) v = fmaf(drc[3], wc.w, v);
            dhc[q] = (hc[q] > 0.0f) ? v : 0.0f;
          }
          LP_SCHED_FENCE();
        }
      }
    }
    if (h == 0) {
      dbo2 += dro;
#pragma unroll
      for (int c = 0; c < NC; ++c) dbc2[c] += drc[c];
    }
    if (want_params) {
      // dW of the two output layers from the wave-private ho / hc tiles
      if (h == 0) {
        ts[r] = dro;
#pragma unroll
        for (int c = 0; c < NC; ++c) ts[(1 + c) * 32 + r] = drc[c];
      }
      const float* tf = ts + 16 * h;
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const float* xf = xt + (32 * b + r) * TW_LD + 16 * h;
        const float* yf = yt + (32 * b + r) * TW_LD + 16 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 hov = *reinterpret_cast<const float4*>(xf + 4 * i);
          const float4 hcv = *reinterpret_cast<const float4*>(yf + 4 * i);
          const float4 d0 = *reinterpret_cast<const float4*>(tf + 4 * i);
          dwo2[b] = fmaf(hov.x, d0.x, dwo2[b]); dwo2[b] = fmaf(hov.y, d0.y, dwo2[b]);
          dwo2[b] = fmaf(hov.z, d0.z, dwo2[b]); dwo2[b] = fmaf(hov.w, d0.w, dwo2[b]);
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const float4 dc = *reinterpret_cast<const float4*>(tf + (1 + c) * 32 + 4 * i);
            dwc2[b][c] = fmaf(hcv.x, dc.x, dwc2[b][c]); dwc2[b][c] = fmaf(hcv.y, dc.y, dwc2[b][c]);
            dwc2[b][c] = fmaf(hcv.z, dc.z, dwc2[b][c]); dwc2[b][c] = fmaf(hcv.w, dc.w, dwc2[b][c]);
          }
          LP_SCHED_FENCE();
        }
      }
    }
    LP_SCHED_FENCE();

    // Every layer: publish the X / dY tiles, dX chain (MFMA), barrier, dW slab, barrier.
    // ---------------- colour hidden layer ----------------
    float dx[16 * NB];
    if (want_params) {
      float ein[16 * NB];
#pragma unroll
      for (int q = 0; q < 16 * NB; ++q) ein[q] = e[q] + enc[q];
      tile_store_w<NB>(xt, r, h, ein);
      tile_store_w<NB>(yt, r, h, dhc);
    }
    layer_t_w<NB, NB, false>(wt + M::WC1, dhc, dx);
    if (want_params) {
      lds_barrier_w();
      dw_slab<NQH, M::PER_WAVE>(wave0, a_off, b_off, dq_c1, db_c1);
    }
#pragma unroll
    for (int q = 0; q < 16 * NB; ++q) denc[q] += dx[q];
    if (want_params) lds_barrier_w();
    LP_SCHED_FENCE();
    // ---------------- opacity hidden layer ----------------
    float dho[16 * NB];
    {
      const float* wo2 = lds + M::WO2 + 4 * h + opaque_zero();
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 wo = *reinterpret_cast<const float4*>(wo2 + 32 * b + 8 * j);
          const int q = 16 * b + 4 * j;
          dho[q + 0] = (ho_mask & (1u << (q + 0))) ? dro * wo.x : 0.0f;
          dho[q + 1] = (ho_mask & (1u << (q + 1))) ? dro * wo.y : 0.0f;
          dho[q + 2] = (ho_mask & (1u << (q + 2))) ? dro * wo.z : 0.0f;
          dho[q + 3] = (ho_mask & (1u << (q + 3))) ? dro * wo.w : 0.0f;
        }
      }
    }
    if (want_params) {
      tile_store_w<NB>(xt, r, h, e);
      tile_store_w<NB>(yt, r, h, dho);
    }
    layer_t_w<NB, NB, true>(wt + M::WO1, dho, dx);
    if (want_params) {
      lds_barrier_w();
      dw_slab<NQH, M::PER_WAVE>(wave0, a_off, b_off, dq_o1, db_o1);
    }
    float de[16 * NB];
#pragma unroll
    for (int q = 0; q < 16 * NB; ++q) de[q] = (e[q] > 0.0f) ? dx[q] : 0.0f;
    if (want_params) lds_barrier_w();
    LP_SCHED_FENCE();
    // ---------------- trunk layer 2 ----------------
    if (want_params) {
      tile_store_w<NB>(xt, r, h, h1);
      tile_store_w<NB>(yt, r, h, de);
    }
    layer_t_w<NB, NB, false>(wt + M::WT2, de, dx);
    if (want_params) {
      lds_barrier_w();
      dw_slab<NQH, M::PER_WAVE>(wave0, a_off, b_off, dq_t2, db_t2);
    }
    float dh1[16 * NB];
#pragma unroll
    for (int q = 0; q < 16 * NB; ++q) dh1[q] = (h1[q] > 0.0f) ? dx[q] : 0.0f;
    if (want_params) lds_barrier_w();
    LP_SCHED_FENCE();
    // ---------------- trunk layer 1 ----------------
    if (want_params) {
#pragma unroll
      for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * TW_LD + r] = x0[q];
      tile_store_w<NB>(yt, r, h, dh1);
    }
    float dx0[16];
    if (gg) layer_t_w<NB, 1, false>(wt + M::WT1, dh1, dx0);  // rows >= C are zero weights
    if (want_params) {
      lds_barrier_w();
      dw_slab<NQ1, M::PER_WAVE>(wave0, a_off, b_off, dq_t1, db_t1);
      lds_barrier_w();
    }
    if (gg) {
#pragma unroll
      for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * DX_LD + r] = dx0[q];
    }
    LP_SCHED_FENCE();
    // ---------------- next (nearer) sample + grid gradient ----------------
    const bool live = valid && on && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));
    if (s > s_lo) fetch_sample<C, GM, true, PLAIN>(a, lds_inf, ray, s - 1, h, nx);
    LP_SCHED_FENCE();
    if (gg && !(mp.dbg & 2)) {
      const int ng = (GM == GM_TRIPLANE) ? 3 : (GM == GM_VOXEL) ? 1 : a.grid.n_grids;
#pragma unroll 1
      for (int g = 0; g < ng; ++g) scatter_grid<C, GM>(a.grad_grid_list[g], a.grid.grids[g], ray.b, x, y, z, live, lane, xt, yt, mp.dbg);
    }
  }

  // ---------------- epilogue ----------------
  if (valid && a.grad_encoding && !seg_on) {
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      float4* dst = reinterpret_cast<float4*>(a.grad_encoding + ray_id * H + 32 * b + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        dst[2 * j] = make_float4(denc[16 * b + 4 * j], denc[16 * b + 4 * j + 1], denc[16 * b + 4 * j + 2],
                                 denc[16 * b + 4 * j + 3]);
    }
  } else if (valid && a.grad_encoding) {  // the segments of a ray add up
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          atomic_add_f32(a.grad_encoding + ray_id * H + 32 * b + 4 * h + 8 * j + i, denc[16 * b + 4 * j + i]);
  }
  if (want_params) {
    float* G = a.grad_mlp_params;
    const int j = lane & 31;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      atomic_add_f32(G + mp.w_o2 + 32 * b + j, dwo2[b]);
      for (int c = 0; c < a.color_chn; ++c) atomic_add_f32(G + mp.w_c2 + (int64_t)(32 * b + j) * mp.ldc2 + c, dwc2[b][c]);
    }
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
    // dW slabs: register i of lane (n16 = l&15, ka = l>>4), quadrant mi is dW[16 mi + pi(4ka+i)][16 wave + pi(n16)]
    const int col = 16 * wave + pi16w(m16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int prow = pi16w(4 * ka + i);
#pragma unroll
      for (int mi = 0; mi < NQH; ++mi) {
        const int row = 16 * mi + prow;
        atomic_add_f32(G + mp.w_t2 + row * H + col, dq_t2[mi][i]);
        atomic_add_f32(G + mp.w_o1 + row * H + col, dq_o1[mi][i]);
        atomic_add_f32(G + mp.w_c1 + row * H + col, dq_c1[mi][i]);
      }
#pragma unroll
      for (int mi = 0; mi < NQ1; ++mi) atomic_add_f32(G + mp.w_t1 + (16 * mi + prow) * H + col, dq_t1[mi][i]);
    }
    db_t1 += __shfl_xor(db_t1, 16); db_t1 += __shfl_xor(db_t1, 32);
    db_t2 += __shfl_xor(db_t2, 16); db_t2 += __shfl_xor(db_t2, 32);
    db_o1 += __shfl_xor(db_o1, 16); db_o1 += __shfl_xor(db_o1, 32);
    db_c1 += __shfl_xor(db_c1, 16); db_c1 += __shfl_xor(db_c1, 32);
    if (ka == 0) {
      atomic_add_f32(G + mp.b_t1 + col, db_t1);
      atomic_add_f32(G + mp.b_t2 + col, db_t2);
      atomic_add_f32(G + mp.b_o1 + col, db_o1);
      atomic_add_f32(G + mp.b_c1 + col, db_c1);
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

bool renderer_mfma_wide_supported(const LpRendererArgs& a, const char** why) {
  *why = "";
  const int C = a.grid.channels;
  constexpr int H = 64;
  if (a.color_grid.n_grids > 0) { *why = "separate colour grid"; return false; }
  if (C != 16 && C != 32) { *why = "grid channels not 16 or 32"; return false; }
  if (a.trunk.n_layers != 2 || a.opacity.n_layers != 2 || a.color.n_layers != 2) {
    *why = "layer counts other than trunk 2 / opacity 2 / colour 2";
    return false;
  }
  if (a.trunk.dims[1] != H || a.trunk.dims[2] != H || a.opacity.dims[1] != H || a.color.dims[1] != H) {
    *why = "hidden width other than 64";
    return false;
  }
  if (a.color_chn > 4) { *why = "more than 4 colour channels"; return false; }
  if (a.grid.n_rows * C * 4 >= (int64_t)1 << 32) { *why = "grid-list of 4 GB or more"; return false; }
  if (a.march.num_samples_inf > MAX_INF) { *why = "more than 256 beyond-far samples"; return false; }
  return true;
}

static MfmaParams make_params_w(const LpRendererArgs& a, int H) {
  MfmaParams p;
  const int C = a.grid.channels;
  p.w_t1 = a.trunk.offset;
  p.w_t2 = p.w_t1 + (int64_t)C * H;
  p.b_t1 = p.w_t2 + H * H;
  p.b_t2 = p.b_t1 + H;
  p.w_o1 = a.opacity.offset;
  p.w_o2 = p.w_o1 + H * H;
  p.b_o1 = p.w_o2 + H;
  p.b_o2 = p.b_o1 + H;
  p.ldc2 = a.color.dims[2];
  p.w_c1 = a.color.offset;
  p.w_c2 = p.w_c1 + H * H;
  p.b_c1 = p.w_c2 + (int64_t)H * p.ldc2;
  p.b_c2 = p.b_c1 + H;
  static const int dbg = getenv("LP_MFMA_DEBUG") ? atoi(getenv("LP_MFMA_DEBUG")) : 0;
  p.dbg = dbg;
  p.seg_blocks = 1;
  p.seg_fwd = 0;
  return p;
}

static int grid_mode_w(const LpRendererArgs& a) {
  auto is_voxel = [](const LpGrid& g) { return g.D > 1 && g.H > 1 && g.W > 1; };
  if (a.grid.n_grids == 1 && is_voxel(a.grid.grids[0])) return GM_VOXEL;
  if (is_canonical_triplane(a.grid)) return GM_TRIPLANE;
  return GM_GENERIC;
}

// Segment-parallel march of a small batch (LpRendererArgs.seg_prefix; same rules as renderer_mfma_segments)
int renderer_mfma_wide_segments(const LpRendererArgs& a) {
  static const int forced = getenv("LP_SEGMENTS") ? atoi(getenv("LP_SEGMENTS")) : -1;
  if (forced == 0 || a.march.num_samples_inf != 0 || a.stop_neg_log_t > 0.0f) return 1;
  const int n_seg = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
  if (n_seg < 2) return 1;
  if (forced < 0 && a.rays.n_rays > 32768) return 1;
  return n_seg;
}

// LP_SEG_LEN-sample blocks per segment: as many segments as keep the launch within `slots` resident workgroups (one round)
static int wide_seg_blocks(const LpRendererArgs& a, unsigned slots) {
  static const int forced = getenv("LP_SEG_BLOCKS") ? atoi(getenv("LP_SEG_BLOCKS")) : 0;
  const unsigned nb = (unsigned)((a.rays.n_rays + WAVES * RAYS_PER_WAVE - 1) / (WAVES * RAYS_PER_WAVE));
  const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
  int m = 1;
  while (m < n_rec && (uint64_t)nb * ((n_rec + m - 1) / m) > slots) ++m;
  if (forced > 0) m = forced < n_rec ? forced : n_rec;
  return m;
}

// mp.seg_blocks > 0 together with `segmented`: the launch is (ray blocks x segments)
template <typename K>
static int launch_w(K kernel, size_t lds, const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream, bool segmented = false) {
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  unsigned nb = (unsigned)((a.rays.n_rays + WAVES * RAYS_PER_WAVE - 1) / (WAVES * RAYS_PER_WAVE));
  if (segmented) {
    const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
    nb *= (unsigned)((n_rec + mp.seg_blocks - 1) / mp.seg_blocks);
  }
  hipLaunchKernelGGL(kernel, dim3(nb), dim3(256), lds, stream, a, mp);
  return LP_OK;
}

#define LP_DISPATCH_W(KERNEL, LDS, SEG)                                                       \
  do {                                                                                     \
    const int gm = grid_mode_w(a);                                                         \
    if (a.grid.channels == 16) {                                                           \
      if (gm == GM_TRIPLANE) rc = launch_w(KERNEL<16, GM_TRIPLANE, 2>, LDS, a, mp, stream, SEG); \
      else if (gm == GM_VOXEL) rc = launch_w(KERNEL<16, GM_VOXEL, 2>, LDS, a, mp, stream, SEG);  \
      else rc = launch_w(KERNEL<16, GM_GENERIC, 2>, LDS, a, mp, stream, SEG);                    \
    } else {                                                                               \
      if (gm == GM_TRIPLANE) rc = launch_w(KERNEL<32, GM_TRIPLANE, 2>, LDS, a, mp, stream, SEG); \
      else if (gm == GM_VOXEL) rc = launch_w(KERNEL<32, GM_VOXEL, 2>, LDS, a, mp, stream, SEG);  \
      else rc = launch_w(KERNEL<32, GM_GENERIC, 2>, LDS, a, mp, stream, SEG);                    \
    }                                                                                      \
  } while (0)

int renderer_forward_mfma_wide(const LpRendererArgs& a, hipStream_t stream) {
  if (a.rays.n_rays == 0) return LP_OK;
  MfmaParams mp = make_params_w(a, 64);
  int rc;
  // small batch with state records: segment march (two workgroups per CU: 512 slots) + combine pass
  static const bool seg_fwd = getenv("LP_SEG_FWD") == nullptr || atoi(getenv("LP_SEG_FWD")) != 0;
  const bool segf = a.seg_prefix && seg_fwd && !a.seg_forward_off;
  if (segf) {
    mp.seg_blocks = wide_seg_blocks(a, 512);
    mp.seg_fwd = 1;
  }
  LP_DISPATCH_W(renderer_fwd_mfma_w, LdsW<2>::FWD_END * sizeof(float), segf);
  if (rc) return rc;
  if (segf && (rc = renderer_forward_combine_launch(a, mp.seg_blocks, stream))) return rc;
  return check_launch("renderer_fwd_mfma_w");
}

int renderer_backward_mfma_wide(const LpRendererArgs& a, hipStream_t stream) {
  if (a.rays.n_rays == 0) return LP_OK;
  MfmaParams mp = make_params_w(a, 64);
  const bool segb = a.seg_prefix != nullptr;  // one workgroup per CU: 256 slots
  if (segb) mp.seg_blocks = wide_seg_blocks(a, 256);
  int rc;
  const bool plain = !(a.noise_sigma > 0.0f) && !a.march.contract_coords && !a.scaffold && a.march.num_samples_inf == 0 &&
                     !(a.stop_neg_log_t > 0.0f);
  const size_t lds_b = LdsW<2>::BWD_END * sizeof(float);
  static const bool no_nc3 = getenv("LP_MFMA_NO_NC3") != nullptr;
  const bool rgb = a.color_chn <= 3 && !no_nc3;
#define LP_BW(CV, GMV)                                                                                     \
  (rgb ? (plain ? launch_w(renderer_bwd_mfma_w<CV, GMV, 2, true, 3>, lds_b, a, mp, stream, segb)                 \
                : launch_w(renderer_bwd_mfma_w<CV, GMV, 2, false, 3>, lds_b, a, mp, stream, segb))               \
       : (plain ? launch_w(renderer_bwd_mfma_w<CV, GMV, 2, true>, lds_b, a, mp, stream, segb)                    \
                : launch_w(renderer_bwd_mfma_w<CV, GMV, 2, false>, lds_b, a, mp, stream, segb)))
  {
    const int gm = grid_mode_w(a);
    if (a.grid.channels == 16) rc = gm == GM_TRIPLANE ? LP_BW(16, GM_TRIPLANE) : gm == GM_VOXEL ? LP_BW(16, GM_VOXEL) : LP_BW(16, GM_GENERIC);
    else rc = gm == GM_TRIPLANE ? LP_BW(32, GM_TRIPLANE) : gm == GM_VOXEL ? LP_BW(32, GM_VOXEL) : LP_BW(32, GM_GENERIC);
  }
#undef LP_BW
  if (rc) return rc;
  return check_launch("renderer_bwd_mfma_w");
}

}  // namespace lp
