// lp_renderer_mfma.hip -- Renderer forward on the CDNA4 matrix cores (gfx950) + host dispatch of the
// MFMA kernels (the backward kernel lives in lp_renderer_mfma_bwd.hip).
//
// Shape family = the TUNED default shape: single grid-list with C in {16, 32} channels, trunk [C,32,32],
// opacity [32,32,1], colour [32,32,>=Cc] with Cc <= 4 -- every BASELINE.json configuration.  Everything else runs the
// layer-looped family (lp_renderer_loop*.hip) or lp_renderer_generic.hip.
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
#include "lp_mfma_common.h"
#include "lp_bf3.h"

namespace lp {

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
// (Rounds 1-3 kept a first generation here -- renderer_fwd_mfma / renderer_fwd_mfma_np on v_mfma_f32_32x32x2_f32, with FLEX and
// two-grid instantiations for the decoders that are not the default shape.  Round 4 retired it: the default shape runs the
// bf16x3 kernels below, every other shallow decoder the layer-looped family's two-waves-per-SIMD instantiations
// (lp_renderer_loop_shallow.hip), which measured faster on every flex shape and within 5 % on the two-grid ones --
// profiles/r04_loop_shallow_ab.txt.)
// NC = 3: at most three colour channels (RGB), the fourth lane of the colour path is compiled out
// SEGF (small batches, LpRendererArgs.seg_prefix): a workgroup marches ONE segment (mp.seg_blocks blocks of LP_SEG_LEN
// samples) of its 128 rays, starting from transmittance 1, and only writes the segment-local running sums into the ray's
// state records; renderer_fwd_combine chains the segments.  Compositing is associative: a segment with local sums
// (L, F, N = -log T over the segment) behind a prefix of transmittance T_0 contributes (T_0 L, T_0 F) and N.
template <int C, int GM, int OCC, int NC, bool SEGF = false>
__global__ void __launch_bounds__(256, OCC) renderer_fwd_bf3(const LpRendererArgs a, const MfmaParams mp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights_bf3<C>(a, mp, lds, false, 256);
  __syncthreads();
  const float* sm = lds - Lds::BIAS;  // small fp32 block: sm[Lds::X]
  const char* fimg = reinterpret_cast<const char*>(lds) + LdsBf3<C>::FWD_IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int seg_len = LP_SEG_LEN * mp.seg_blocks;
  const int n_seg = SEGF ? (a.march.num_samples + seg_len - 1) / seg_len : 1;
  const int blk = SEGF ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
  const int seg = SEGF ? (int)blockIdx.x - blk * n_seg : 0;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float cb[16];  // per-ray pre-activation of the colour hidden layer (replaces the ray encoding in the loop)
  {
    float enc[16];
    load_encoding(a, rid, h, enc);
    color_prebias_bf3<C>(sm, fimg, lane, enc, cb);
  }
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_ckpt = ckpt_count(a.march);
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;
  float nlt = 0.0f, nlt_lo = 0.0f, t_prev = 1.0f, len = 0.0f, depth_prev = 0.0f;
  int s_last = s_tot - 1;  // last sample marched
  float facc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  Sample<C> nx;
  Act<C> t;
  const int s_lo = SEGF ? seg * seg_len : 0;
  const int s_hi = SEGF ? ((s_lo + seg_len < s_tot) ? s_lo + seg_len : s_tot) : s_tot;
  if (SEGF && s_lo > 0) {  // interval length of the segment's first sample
    sample_geometry<C>(a, sm, ray, s_lo - 1, nx);
    depth_prev = nx.depth;
  }
  for (int s = s_lo; s < s_hi; ++s) {
    fetch_sample<C, GM, true>(a, sm, ray, s, h, nx);
    const float depth = nx.depth, occ = nx.occ;
#pragma unroll
    for (int q = 0; q < C / 2; ++q) t.x0[q] = nx.x0[q];
    const int zo = opaque_zero();
    const Heads hd = decode_bf3<C, NC>(sm, fimg, lane, cb, t, zo);
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    depth_prev = depth;
    float raw = hd.raw_o;
    if (a.noise_sigma > 0.0f) raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    nlt_add(nlt, nlt_lo, opacity * delta);
    if (!SEGF && a.neg_log_t_ckpt && valid && h == 0) {
      const int ck = ckpt_index(s, a.march);
      if (ck >= 0) *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + ck) * 2) = make_float2(nlt, nlt_lo);
    }
    const float tr = __expf(-nlt);
    const float w = t_prev - tr;
    t_prev = tr;
    len = fmaf(w, depth, len);
#pragma unroll
    for (int c = 0; c < NC; ++c) facc[c] = fmaf(w, sigmoid_f(hd.raw_c[c]) * occ, facc[c]);
    // segment-parallel backward (LpRendererArgs.seg_prefix): the state after every block of LP_SEG_LEN samples
    // (SEGF: relative to the segment's start; renderer_fwd_combine makes it absolute)
    if (a.seg_prefix && valid && h == 0 && (((s + 1) % LP_SEG_LEN) == 0 || s == a.march.num_samples - 1)) {
      float4* dst = reinterpret_cast<float4*>(a.seg_prefix + (ray_id * segment_count(a.march) + s / LP_SEG_LEN) * 8);
      dst[0] = make_float4(len, facc[0], facc[1], facc[2]);
      dst[1] = make_float4(NC == 4 ? facc[3] : 0.0f, nlt, nlt_lo, 0.0f);
    }
    // early termination (off unless stop_neg_log_t > 0): every ray of this wave is opaque
    if (a.stop_neg_log_t > 0.0f && __ballot(valid && nlt < a.stop_neg_log_t) == 0) {
      s_last = s;
      break;
    }
  }
  if (!SEGF && valid && h == 0) {
    write_ray_outputs(a, ray_id, len, nlt, facc);
    if (a.neg_log_t_ckpt)
      *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + n_ckpt - 1) * 2) = make_float2((float)s_last, nlt_lo);
  }
}

// Chains the segments of a segmented forward (one thread per ray): turns the segment-local state records into the
// absolute ones the backward reads, writes the -log T checkpoints and the ray's outputs.
__global__ void __launch_bounds__(256) renderer_fwd_combine(const LpRendererArgs a, int seg_blocks) {
  const int64_t ray_id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (ray_id >= a.rays.n_rays) return;
  const int S = a.march.num_samples;
  const int n_rec = segment_count(a.march), n_ckpt = ckpt_count(a.march);
  float4* rec = reinterpret_cast<float4*>(a.seg_prefix + ray_id * n_rec * 8);
  // state at the start of the current segment
  float n0 = 0.0f, n0_lo = 0.0f, l0 = 0.0f, f0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, t0 = 1.0f;
  float nlt = 0.0f, nlt_lo = 0.0f, len = 0.0f, f[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int b = 0; b < n_rec; ++b) {
    if (b > 0 && b % seg_blocks == 0) {  // a new segment starts behind everything accumulated so far
      n0 = nlt; n0_lo = nlt_lo; l0 = len; t0 = __expf(-nlt);
      for (int c = 0; c < 4; ++c) f0[c] = f[c];
    }
    const float4 r0 = rec[2 * b], r1 = rec[2 * b + 1];
    nlt = n0; nlt_lo = n0_lo;
    nlt_add(nlt, nlt_lo, r1.y);
    nlt_add(nlt, nlt_lo, r1.z);
    len = fmaf(t0, r0.x, l0);
    f[0] = fmaf(t0, r0.y, f0[0]);
    f[1] = fmaf(t0, r0.z, f0[1]);
    f[2] = fmaf(t0, r0.w, f0[2]);
    f[3] = fmaf(t0, r1.x, f0[3]);
    rec[2 * b] = make_float4(len, f[0], f[1], f[2]);
    rec[2 * b + 1] = make_float4(f[3], nlt, nlt_lo, 0.0f);
    const int s_end = ((b + 1) * LP_SEG_LEN < S ? (b + 1) * LP_SEG_LEN : S) - 1;  // last sample of the block
    if (a.neg_log_t_ckpt) {
      const int ck = ckpt_index(s_end, a.march);
      if (ck >= 0) *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + ck) * 2) = make_float2(nlt, nlt_lo);
    }
  }
  write_ray_outputs(a, ray_id, len, nlt, f);
  if (a.neg_log_t_ckpt)
    *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + n_ckpt - 1) * 2) = make_float2((float)(S - 1), nlt_lo);
}

// ---------------------------------------------------------------------------------------
// forward with a TRANSPOSED MARCH (LpRendererArgs.march_order == LP_MARCH_SAMPLES_PER_WAVE): a wave = ONE ray x 32 consecutive
// samples, its rays one after the other -- the forward twin of lp_renderer_mfma_bwd_tm.hip.  For batches of unrelated rays: the 12 / 8
// corner gathers of a wave's lanes then touch consecutive cells of one ray (L1 hits) instead of 32 unrelated places, and a small
// batch is dealt over the chip by rays per wave (mp.tm_rpw) instead of by the segmented march + combine pass.
// Compositing along the lanes: -log T = carry (a float pair, as in the sequential march) + an inclusive wave scan of opacity *
// delta; the ray's sums are lane partials reduced once per ray.  Writes the same -log T checkpoints (one per block of LP_NLT_CKPT =
// 32 samples + the closing pair).  No beyond-far samples, no early termination, >= 32 samples (renderer_tm_eligible()).
// ---------------------------------------------------------------------------------------
LP_DEV float fwd_scan_up(float v, int r) {  // inclusive prefix sum over the 32 lanes of a half (r = lane & 31)
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float t = __shfl_up(v, d, 32);
    v += (r >= d) ? t : 0.0f;
  }
  return v;
}

template <int C, int GM, int NC>
__global__ void __launch_bounds__(256, 3) renderer_fwd_bf3_tm(const LpRendererArgs a, const MfmaParams mp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights_bf3<C>(a, mp, lds, false, 256);
  __syncthreads();
  const float* sm = lds - Lds::BIAS;
  const char* fimg = reinterpret_cast<const char*>(lds) + LdsBf3<C>::FWD_IMG;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  const int S = a.march.num_samples;
  const int n_blk = (S + 31) >> 5;
  const int n_ckpt = ckpt_count(a.march);
  const int rpw = mp.tm_rpw;
  const int64_t ray0 = ((int64_t)blockIdx.x * WAVES + wave) * rpw;
  for (int k = 0; k < rpw; ++k) {
    const int64_t ray_id = ray0 + k;
    if (ray_id >= a.rays.n_rays) break;  // (wave-uniform; no workgroup barrier below)
    const Ray ray = load_ray(a.rays, ray_id);
    float cb[16];
    {
      float enc[16];
      load_encoding(a, ray_id, h, enc);
      color_prebias_bf3<C>(sm, fimg, lane, enc, cb);
    }
    const float delta0 = (S > 1) ? (ray.far_t - ray.near_t) / (float)(S - 1) : 1.0f;
    float n_hi = 0.0f, n_lo = 0.0f;  // -log T behind the previous block
    float len = 0.0f, facc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    Sample<C> nx;
    Act<C> t;
    for (int bs = 0; bs < n_blk; ++bs) {
      const int s = bs * 32 + r;
      const bool on = s < S;
      const int sc = on ? s : S - 1;
      fetch_sample<C, GM, true>(a, sm, ray, sc, h, nx);
      const float depth = nx.depth, occ = nx.occ;
#pragma unroll
      for (int q = 0; q < C / 2; ++q) t.x0[q] = nx.x0[q];
      const int zo = opaque_zero();
      const Heads hd = decode_bf3<C, NC>(sm, fimg, lane, cb, t, zo);
      const float depth_prev = sample_depth_tab((sc > 0) ? sc - 1 : 0, a.march, ray.near_t, ray.far_t, sm + Lds::INF);
      const float delta = (sc == 0) ? delta0 : depth - depth_prev;
      float raw = hd.raw_o;
      if (a.noise_sigma > 0.0f) raw = raw + sample_noise(ray_id, sc, a.rays.n_rays, S, a.noise_seed) * a.noise_sigma;
      const float od = on ? a.gain * softplus_f(raw) * occ * delta : 0.0f;
      const float incl = fwd_scan_up(od, r);
      const float nlt_s = n_hi + (incl + n_lo), nlt_p = n_hi + ((incl - od) + n_lo);
      const float w = on ? __expf(-nlt_p) - __expf(-nlt_s) : 0.0f;
      len = fmaf(w, depth, len);
#pragma unroll
      for (int c = 0; c < NC; ++c) facc[c] = fmaf(w, sigmoid_f(hd.raw_c[c]) * occ, facc[c]);
      // carry: the block's total joins the float pair; lane 0 of half 0 writes the checkpoint behind the block
      nlt_add(n_hi, n_lo, __shfl(incl, 31, 32));
      if (a.neg_log_t_ckpt && lane == 0) *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + bs) * 2) = make_float2(n_hi, n_lo);
    }
    // the ray's sums: lane partials -> one value
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
      len += __shfl_xor(len, m, 32);
#pragma unroll
      for (int c = 0; c < NC; ++c) facc[c] += __shfl_xor(facc[c], m, 32);
    }
    if (lane == 0) {
      write_ray_outputs(a, ray_id, len, n_hi, facc);
      if (a.neg_log_t_ckpt) *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + n_ckpt - 1) * 2) = make_float2((float)(S - 1), n_lo);
    }
  }
}

// forward + backward of these arguments may march samples per wavefront
bool renderer_tm_eligible(const LpRendererArgs& a) {
  static const bool off = getenv("LP_TM_OFF") != nullptr;  // A/B
  return !off && a.march_order == LP_MARCH_SAMPLES_PER_WAVE && a.march.num_samples_inf == 0 && !(a.stop_neg_log_t > 0.0f) &&
         a.march.num_samples >= 32;
}

// rays per wave of the transposed march: 32 once the batch fills one round of resident workgroups; a smaller batch is spread over
// the chip (measured on 65 536 random rays, fwd + bwd: 32 rays per wave 7.15 ms, 16: 7.52, 8: 7.78, 4: 8.86 -- profiles/r06_transposed_march.txt)
int renderer_tm_rays_per_wave(const LpRendererArgs& a, int resident_workgroups) {
  static const int forced = getenv("LP_TM_RPW") ? atoi(getenv("LP_TM_RPW")) : 0;
  int rpw = RAYS_PER_WAVE;
  while (rpw > 1 && (a.rays.n_rays + WAVES * rpw - 1) / (WAVES * rpw) < resident_workgroups) rpw >>= 1;
  if (forced >= 1 && forced <= RAYS_PER_WAVE) rpw = forced;
  return rpw;
}

int renderer_forward_combine_launch(const LpRendererArgs& a, int seg_blocks, hipStream_t stream) {
  hipLaunchKernelGGL(renderer_fwd_combine, dim3((unsigned)((a.rays.n_rays + 255) / 256)), dim3(256), 0, stream, a, seg_blocks);
  return LP_OK;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

// Shape family of these kernels: grid-list(s) with C in {16, 32} channels below 2^31 rows (any byte size); trunk of 1 or 2 layers -- or
// none with a separate colour grid-list (the reference's two-grid decoder: the heads read relu(sampled feature)
// of their own grid) --, opacity / colour heads of 1 or 2 layers, every hidden width equal to H in {16, 32};
// <= 4 colour channels.  The default shape (2/2/2 layers, H = 32, one grid-list) runs the tuned kernels; the
// others run the same kernels with the absent layers skipped and narrow widths zero-padded to 32 ("flex").
bool renderer_mfma_supported(const LpRendererArgs& a, const char** why) {
  *why = "";
  const int C = a.grid.channels;
  const bool tg = a.color_grid.n_grids > 0;  // two-grid decoder: no trunk, the heads read the two sampled features
  if (C != 16 && C != 32) { *why = "grid channels not 16 or 32"; return false; }
  if (a.trunk.n_layers > 2 || a.opacity.n_layers < 1 || a.opacity.n_layers > 2 || a.color.n_layers < 1 ||
      a.color.n_layers > 2 || (!tg && a.trunk.n_layers < 1) || (tg && a.trunk.n_layers != 0)) {
    *why = "layer counts outside trunk 1-2 (0 with a colour grid) / opacity 1-2 / colour 1-2";
    return false;
  }
  // one hidden width H for every hidden layer (trunk layers, hidden layers of the heads)
  int H = 0;
  bool same = true;
  auto hidden = [&](int w) { if (H == 0) H = w; else same = same && (w == H); };
  for (int l = 1; l <= a.trunk.n_layers; ++l) hidden(a.trunk.dims[l]);
  if (a.opacity.n_layers == 2) hidden(a.opacity.dims[1]);
  if (a.color.n_layers == 2) hidden(a.color.dims[1]);
  if (H == 0) H = C;  // two-grid decoder with single-layer heads: no hidden layer at all
  if (H != 16 && H != 32) { *why = "hidden width other than 16 / 32"; return false; }
  if (!same) { *why = "hidden widths differ between layers"; return false; }
  if (a.color_chn > 4) { *why = "more than 4 colour channels"; return false; }
  if (!grid_list_rows_ok(a.grid)) { *why = "grid-list of 2^31 rows or more (or a grid slice of 2 GB or more)"; return false; }
  if (tg && !grid_list_rows_ok(a.color_grid)) { *why = "colour grid-list of 2^31 rows or more (or a grid slice of 2 GB or more)"; return false; }
  if (a.march.num_samples_inf > MAX_INF) { *why = "more than 256 beyond-far samples"; return false; }
  // the tuned kernels are written for ONE decoder shape; the flex / two-grid subsets this family used to take (fp32-MFMA
  // kernels with run-time layer flags) belong to the layer-looped family since round 4
  if (!(H == HID && a.trunk.n_layers == 2 && a.opacity.n_layers == 2 && a.color.n_layers == 2 && !tg)) {
    *why = "not the tuned default shape (2/2/2 layers x 32 hidden): layer-looped family";
    return false;
  }
  return true;
}

static MfmaParams make_params(const LpRendererArgs& a) {
  MfmaParams p;
  const int C = a.grid.channels;
  p.tg = a.color_grid.n_grids > 0;
  p.t1 = a.trunk.n_layers >= 1;
  p.t2 = a.trunk.n_layers == 2;
  p.oh = a.opacity.n_layers == 2;
  p.ch = a.color.n_layers == 2;
  const int H = p.t1 ? a.trunk.dims[1] : (p.oh ? a.opacity.dims[1] : (p.ch ? a.color.dims[1] : C));
  p.hid = H;
  p.hin = p.tg ? C : H;  // input width of the heads = width of the ray encoding
  // trunk: W_1 [C,H] (, W_2 [H,H]), b_1 (, b_2)
  p.w_t1 = a.trunk.offset;
  p.w_t2 = p.w_t1 + (int64_t)C * H;
  p.b_t1 = p.w_t2 + (p.t2 ? H * H : 0);
  p.b_t2 = p.b_t1 + H;
  // opacity: (W_1 [hin,H],) W_out [H or hin,1], (b_1,) b_out
  p.w_o1 = a.opacity.offset;
  p.w_o2 = p.w_o1 + (p.oh ? p.hin * H : 0);
  p.b_o1 = p.w_o2 + (p.oh ? H : p.hin);
  p.b_o2 = p.b_o1 + (p.oh ? H : 0);
  // colour: (W_1 [hin,H],) W_out [H or hin,ldc2], (b_1,) b_out
  p.ldc2 = a.color.dims[a.color.n_layers];
  p.w_c1 = a.color.offset;
  p.w_c2 = p.w_c1 + (p.ch ? p.hin * H : 0);
  p.b_c1 = p.w_c2 + (int64_t)(p.ch ? H : p.hin) * p.ldc2;
  p.b_c2 = p.b_c1 + (p.ch ? H : 0);
  static const int dbg = getenv("LP_MFMA_DEBUG") ? atoi(getenv("LP_MFMA_DEBUG")) : 0;
  p.dbg = dbg;
  p.seg_blocks = 1;
  p.seg_fwd = 0;
  p.relu_dump = g_relu_dump;
  p.tm_rpw = RAYS_PER_WAVE;
  return p;
}


// Segment-parallel backward (LpRendererArgs.seg_prefix): available where renderer_fwd_bf3 / renderer_bwd_bf3 run (default
// decoder shape), without beyond-far samples and early termination; worth it while the batch leaves wave slots
// idle (a 4-wave workgroup per 128 rays, two workgroups per CU: 65 536 rays fill the chip once).  Measured on MI355X
// (scripts/bench_small_batch.py, S = 128, backward kernel): 4 096 rays 1.80 -> 0.51 ms, 16 384 rays 1.80 -> 0.88 ms,
// 32 768 rays 1.75 -> 1.53 ms, 49 152 rays 2.12 -> 2.22 ms: on up to 32 768 rays.
// LP_SEGMENTS=0 / 1 switches it off / on regardless of the batch size (A/B, tests).
// LP_ARITH_FP32 instantiations exist for four-wave workgroups (<= 64 beyond-far samples), one sweep per ray
bool renderer_mfma_f32_supported(const LpRendererArgs& a) {
  const char* why = "";
  return renderer_mfma_supported(a, &why) && a.march.num_samples_inf <= 64;
}

int renderer_mfma_segments(const LpRendererArgs& a) {
  static const int forced = getenv("LP_SEGMENTS") ? atoi(getenv("LP_SEGMENTS")) : -1;
  if (forced == 0 || a.arithmetic != LP_ARITH_DEFAULT) return 1;
  if (renderer_tm_eligible(a)) return 1;  // the transposed march deals a small batch over the chip by rays per wave: no segments
  if (a.march.num_samples_inf != 0 || a.stop_neg_log_t > 0.0f) return 1;
  const int n_seg = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
  if (n_seg < 2) return 1;
  if (forced < 0 && a.rays.n_rays > 32768) return 1;
  return n_seg;
}

static int grid_mode(const LpRendererArgs& a) {
  auto is_voxel = [](const LpGrid& g) { return g.D > 1 && g.H > 1 && g.W > 1; };
  static const bool force_generic = getenv("LP_MFMA_GENERIC_GRIDS") != nullptr;  // debugging aid
  if (force_generic) return GM_GENERIC;
  if (a.grid.n_grids == 1 && is_voxel(a.grid.grids[0])) return GM_VOXEL;
  if (is_canonical_triplane(a.grid)) return GM_TRIPLANE;
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
  int rc;
  static const bool no_nc3 = getenv("LP_MFMA_NO_NC3") != nullptr;  // A/B knob
  static const bool tm_fwd_off = getenv("LP_TM_FWD_OFF") != nullptr;  // A/B: transposed backward, rays-per-wavefront forward
  if (renderer_tm_eligible(a) && !tm_fwd_off) {  // samples per wavefront (batches of unrelated rays)
    MfmaParams mt = mp;
    mt.tm_rpw = renderer_tm_rays_per_wave(a, 768);  // (three workgroups per CU)
    const size_t lds3 = (size_t)LdsBf3<C>::FWD_END;
    const unsigned nb = (unsigned)((a.rays.n_rays + WAVES * mt.tm_rpw - 1) / (WAVES * mt.tm_rpw));
    if (a.color_chn <= 3 && !no_nc3) {
      if ((rc = set_lds(renderer_fwd_bf3_tm<C, GM, 3>, lds3))) return rc;
      hipLaunchKernelGGL((renderer_fwd_bf3_tm<C, GM, 3>), dim3(nb), dim3(256), lds3, stream, a, mt);
    } else {
      if ((rc = set_lds(renderer_fwd_bf3_tm<C, GM, 4>, lds3))) return rc;
      hipLaunchKernelGGL((renderer_fwd_bf3_tm<C, GM, 4>), dim3(nb), dim3(256), lds3, stream, a, mt);
    }
    return LP_OK;
  }
  static const int bf3_occ = getenv("LP_BF3_OCC") ? atoi(getenv("LP_BF3_OCC")) : 0;
  {
    const size_t lds3 = (size_t)LdsBf3<C>::FWD_END;
    const int occ = bf3_occ ? bf3_occ : (a.rays.n_rays > 2 * 32768 ? 3 : 2);
    // small batch with state records (seg_prefix survives lp_api.hip only where renderer_mfma_segments() > 1): one
    // workgroup per (128 rays, segment) + the combine pass.  LP_SEG_FWD=0: one march per ray (the records are written all
    // the same).  Segment length: as for the backward, as many segments as fit one round of resident workgroups.
    static const bool seg_fwd = getenv("LP_SEG_FWD") == nullptr || atoi(getenv("LP_SEG_FWD")) != 0;
    if (a.seg_prefix && seg_fwd && !a.seg_forward_off) {
      MfmaParams ms = mp;
      const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
      const int m = seg_blocks_for(n_blocks(a), n_rec, 512u);
      ms.seg_blocks = m;
      const unsigned nb = n_blocks(a) * (unsigned)((n_rec + m - 1) / m);
      if (a.color_chn <= 3 && !no_nc3) {
        if ((rc = set_lds(renderer_fwd_bf3<C, GM, 2, 3, true>, lds3))) return rc;
        hipLaunchKernelGGL((renderer_fwd_bf3<C, GM, 2, 3, true>), dim3(nb), dim3(256), lds3, stream, a, ms);
      } else {
        if ((rc = set_lds(renderer_fwd_bf3<C, GM, 2, 4, true>, lds3))) return rc;
        hipLaunchKernelGGL((renderer_fwd_bf3<C, GM, 2, 4, true>), dim3(nb), dim3(256), lds3, stream, a, ms);
      }
      hipLaunchKernelGGL(renderer_fwd_combine, dim3((unsigned)((a.rays.n_rays + 255) / 256)), dim3(256), 0, stream, a, m);
      return LP_OK;
    }
#define LP_BF3_LAUNCH(OCCV, NCV)                                                                                  \
    do {                                                                                                          \
      if ((rc = set_lds(renderer_fwd_bf3<C, GM, OCCV, NCV>, lds3))) return rc;                                   \
      hipLaunchKernelGGL((renderer_fwd_bf3<C, GM, OCCV, NCV>), dim3(n_blocks(a)), dim3(256), lds3, stream, a, mp); \
    } while (0)
    const bool nc3 = a.color_chn <= 3 && !no_nc3;
    if (occ >= 4 && C == 16) { if (nc3) LP_BF3_LAUNCH(4, 3); else LP_BF3_LAUNCH(4, 4); }
    else if (occ == 3 || occ >= 4) { if (nc3) LP_BF3_LAUNCH(3, 3); else LP_BF3_LAUNCH(3, 4); }
    else { if (nc3) LP_BF3_LAUNCH(2, 3); else LP_BF3_LAUNCH(2, 4); }
#undef LP_BF3_LAUNCH
    return LP_OK;
  }
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
  return renderer_backward_mfma2(a, mp, grid_mode(a), stream);  // lp_renderer_mfma_bwd.hip
}

}  // namespace lp
