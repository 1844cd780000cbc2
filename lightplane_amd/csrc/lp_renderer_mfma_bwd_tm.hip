// lp_renderer_mfma_bwd_tm.hip -- the tuned Renderer backward with a TRANSPOSED MARCH: a wave = ONE ray x 32 consecutive samples
// (lane (h, r): sample 32 b + r of the wave's current ray), instead of 32 rays x one sample (lp_renderer_mfma_bwd.h).
//
// Why.  The grid-gradient scatter merges the taps of CONSECUTIVE LANES that fall into the same cell into one row-contiguous atomic
// per run (lp_mfma_common.h, scatter_triplane / scatter_grid / the voxel column walk).  With lane = ray that is spatial coherence:
// neighbouring pixels of an image.  Random ray batches -- what NeRF-style training feeds and what the reference's own speed
// benchmark draws (tests/renderer_speed_benchmark.py:228-246, tests/utils.py:230-268) -- have none: every ray is its own run,
// 12 taps x 32 rays x C / 16 atomic segments per wave-sample, and the backward sits at the chip's atomic-segment rate (refbench
// 256^2: 403 M segments, 114 k of 200 k cycles per wave-sample in the scatter, profiles/r06_refbench256_phase_cycles.txt).  But
// consecutive SAMPLES of one ray do share cells (256 samples across a 32-cell plane: ~6 per cell).  With lane = sample the very
// same run merge -- ballot run heads, scalar-branched walk, one atomic per run and row -- merges them: temporal coherence through
// the unchanged scatter code.  The gathers of a wave become consecutive cells of one ray as well.
//
// What changes against renderer_bwd_bf3 (everything else -- bf16x3 recompute, two-limb dX chains, workgroup-shared bf16 dW
// quadrants, limb-image layout -- is the same code):
//  * loop: 32 rays of the wave one after the other, each as ceil(S / 32) blocks of 32 samples, far -> near; every wave of a
//    workgroup runs 32 * ceil(S / 32) iterations (the dW barriers stay workgroup-uniform);
//  * compositing: -log T of the block's samples = the forward's checkpoint behind the previous block (LP_NLT_CKPT = 32 = the block
//    length) + an inclusive wave scan of opacity * delta; the suffix term sum_{i > s} w_i p_i is a reverse scan of
//    T_i (p_i - p_{i+1}) with a carry from the block behind;
//  * the ray encoding is wave-uniform: cb = b_c1 + W_c1^T enc and enc live in a 256-byte record per wave; the colour layer's weight
//    gradient takes X = e + enc directly (no per-ray epilogue product), d enc = W_c1 sum_s d hc is formed once per ray after a
//    cross-lane reduction.
// Scope: no beyond-far samples, no early termination (opacity noise, contraction, scaffold and the out-of-bounds mask are covered:
// PLAIN = false), at least 32 samples, four-wave workgroups, default arithmetic.  Selected by LpRendererArgs.march_order == LP_MARCH_SAMPLES_PER_WAVE (the Python
// front-end sets it for batches whose consecutive rays do not share an origin, see lightplane_amd/renderer.py).
#include "lp_renderer_mfma_bwd.h"

namespace lp {

LP_DEV float tm_scan_up(float v, int r) {  // inclusive prefix sum over the 32 lanes of a half (r = lane & 31)
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float t = __shfl_up(v, d, 32);
    v += (r >= d) ? t : 0.0f;
  }
  return v;
}
LP_DEV float tm_scan_down(float v, int r) {  // inclusive suffix sum
#pragma unroll
  for (int d = 1; d < 32; d <<= 1) {
    const float t = __shfl_down(v, d, 32);
    v += (r + d < 32) ? t : 0.0f;
  }
  return v;
}

template <int C>
struct TmLds {
  static constexpr int IMG_END = LdsBf3Rm<C>::END;
  static constexpr int REC = IMG_END + WAVES * LdsB3::PER_WAVE * 4;   // per-wave ray records: cb [2][16], enc [2][16] floats
  static constexpr int TOTAL = REC + WAVES * 64 * 4;
  static_assert(2 * TOTAL <= 160 * 1024, "two workgroups per CU");
};

template <int C, int GM, int NC, bool PLAIN, bool DUMP = false>
__global__ void __launch_bounds__(256, 2) renderer_bwd_bf3_tm(const LpRendererArgs a, const MfmaParams mp) {
  using M = Lds;
  using R = LdsBf3Rm<C>;
  using B = LdsB3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_weights_rm<C>(a, mp, lds, 256);
  const float* const sm = lds - M::BIAS;
  const char* const rimg = reinterpret_cast<const char*>(lds);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, r = lane & 31;
  float* const wave0 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + TmLds<C>::IMG_END);
  float* const wv = wave0 + wave * B::PER_WAVE;
  float* const xt = wv + B::XT;
  float* const yt = wv + B::YT;
  float* const ts = wv + B::TS;
  float* const rec = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + TmLds<C>::REC) + wave * 64;
  __syncthreads();

  const int S = a.march.num_samples;
  const int n_blk = (S + 31) >> 5;
  const int n_ckpt = ckpt_count(a.march);
  const int rpw = mp.tm_rpw;  // rays of this wave (a small batch is dealt over more workgroups: fewer rays per wave)
  const int64_t ray0 = ((int64_t)blockIdx.x * WAVES + wave) * rpw;
  const bool want_params = a.grad_mlp_params != nullptr;
  const bool gg = a.grad_grid_list[0] != nullptr;

  // dW quadrant of this wave (bf16 quadrants, see renderer_bwd_bf3)
  const int mi = (wave & 3) >> 1, ni = wave & 1;
  const int m16 = lane & 15, ka = lane >> 4;
  auto rho = [](int k) { return (k & 0x15) | ((k & 2) << 2) | ((k & 8) >> 2); };
  char* const xrow = reinterpret_cast<char*>(wv) + B::XT * 4 + rm_off(rho(r), 4 * h);
  char* const yrow = xrow + (B::YT - B::XT) * 4;
  const int xq_off = B::XT * 4 + rm_off(rho(8 * ka + (m16 >> 2)), 4 * (m16 & 3)) + 32 * mi;
  const int yq_off = B::YT * 4 + rm_off(rho(8 * ka + (m16 >> 2)), 4 * (m16 & 3)) + 32 * ni;
  f32x4 dq_b = {0, 0, 0, 0};
  auto onehot = [&](int li) -> unsigned { return (lane & 15) == li ? 0x3F803F80u : 0u; };
  const int t1_v0 = (C == 16) ? 2 * (wave >> 1) : 0, t1_v1 = (C == 16) ? 2 * (wave >> 1) + 2 : 4;
  f32x4 dq_t1 = {0, 0, 0, 0}, dq_t2 = {0, 0, 0, 0}, dq_o1 = {0, 0, 0, 0}, dq_c1 = {0, 0, 0, 0};
  float dwo2 = 0.0f, dwc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float dbo2 = 0.0f, dbc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const char* const wave0b = reinterpret_cast<const char*>(wave0);

  float dsum[16];  // D = sum over the current ray's samples (this lane's) of d hc
#pragma unroll
  for (int q = 0; q < 16; ++q) dsum[q] = 0.0f;

  // the wave's current ray (wave-uniform values held per lane) and block
  int k = 0, bs = n_blk - 1;
  bool new_ray = true;
  int64_t ray_id = ray0;
  bool valid = ray_id < a.rays.n_rays;
  int64_t rid = valid ? ray_id : 0;
  Ray ray = load_ray(a.rays, rid);
  float gfeat[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float g_len = 0.0f, g_nlt = 0.0f, delta0 = 1.0f;
  float suffix_carry = 0.0f, p_carry = 0.0f;

  auto sample_of = [&](int blk_s) { const int s = blk_s * 32 + r; return s < S ? s : S - 1; };  // (lanes beyond S re-do the last sample, dead)
  Sample<C> nx;
  fetch_sample<C, GM, false, PLAIN>(a, sm, ray, sample_of(bs), h, nx);
  const int n_it = rpw * n_blk;
  for (int it = 0; it < n_it; ++it) {
    if (new_ray) {  // wave-uniform: per-ray scalars, cb = b_c1 + W_c1^T enc and enc into the wave's record
#pragma unroll
      for (int c = 0; c < 4; ++c) gfeat[c] = (valid && a.grad_feature && c < a.color_chn) ? a.grad_feature[rid * a.color_chn + c] : 0.0f;
      g_len = (valid && a.grad_ray_length) ? a.grad_ray_length[rid] : 0.0f;
      g_nlt = epilogue_grad_nlt(a, rid, valid, a.neg_log_t[rid], (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f, gfeat, 4);
      delta0 = (S > 1) ? (ray.far_t - ray.near_t) / (float)(S - 1) : 1.0f;
      float enc[16], cb[16];
      load_encoding(a, rid, h, enc);
      color_prebias_bf3(sm, AColsFwd{rimg + R::L_C1, R::ST_32}, lane, enc, cb);
      if (r == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          *reinterpret_cast<float4*>(rec + 16 * h + 4 * i) = make_float4(cb[4 * i], cb[4 * i + 1], cb[4 * i + 2], cb[4 * i + 3]);
          *reinterpret_cast<float4*>(rec + 32 + 16 * h + 4 * i) = make_float4(enc[4 * i], enc[4 * i + 1], enc[4 * i + 2], enc[4 * i + 3]);
        }
      }
      suffix_carry = 0.0f;
      p_carry = 0.0f;
    }
    const int s = bs * 32 + r;
    const bool on = s < S;
    const float depth = nx.depth, occ = nx.occ, x = nx.x, y = nx.y, z = nx.z;
    const int b_cur = ray.b;
    float x0[C / 2];
#pragma unroll
    for (int q = 0; q < C / 2; ++q) x0[q] = nx.x0[q];
    const int zo = opaque_zero();
    const float* ldz = sm + zo;
    const float* recz = rec + zo;
    auto Af = [&](auto layer) {
      constexpr int l = decltype(layer)::value;
      return AColsFwd{rimg + zo + (l == 0 ? R::L_T1 : l == 1 ? R::L_T2 : l == 2 ? R::L_O1 : R::L_C1), l == 0 ? R::ST_T1 : R::ST_32};
    };
    auto Ab = [&](auto layer) {
      constexpr int l = decltype(layer)::value;
      return ARowsBwd{rimg + zo + (l == 0 ? R::L_T1 : l == 1 ? R::L_T2 : l == 2 ? R::L_O1 : R::L_C1), l == 0 ? R::ST_T1 : R::ST_32, l == 0 ? C - 1 : 31};
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---------------- forward recompute (bf16x3) ----------------
    float h1[16];
    float e[16];
    unsigned ho_mask = 0, hc_mask = 0;
    Heads hd;
    {
      f32x16 acc = layer_bf3v<C / 16>(Af(I0{}), lane, x0, load_bias(sm, 0, h, zo));
#pragma unroll
      for (int q = 0; q < 16; ++q) h1[q] = relu_f(acc[q]);
      acc = layer_bf3v<2>(Af(I1{}), lane, h1, load_bias(sm, 1, h, zo));
#pragma unroll
      for (int q = 0; q < 16; ++q) e[q] = relu_f(acc[q]);
      float ho[16], hc[16];
      {
        f32x16 acc_o = load_bias(sm, 2, h, zo), acc_c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(recz + 16 * h + 4 * i);
          acc_c[4 * i] = v.x; acc_c[4 * i + 1] = v.y; acc_c[4 * i + 2] = v.z; acc_c[4 * i + 3] = v.w;
        }
        layer2_bf3v<2>(Af(I2{}), Af(I3{}), lane, e, acc_o, acc_c);
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          ho[q] = relu_f(acc_o[q]);
          hc[q] = relu_f(acc_c[q]);
        }
      }
      hd = heads_forward<NC>(sm, h, ho, hc, zo);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        ho_mask = mask_bit(ho_mask, ho[q], q);
        hc_mask = mask_bit(hc_mask, hc[q], q);
      }
      if constexpr (DUMP) {
        unsigned m1 = 0, m2 = 0, mo = 0, mc = 0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const unsigned bit = 1u << featq(q, h);
          m1 |= (h1[q] > 0.0f) ? bit : 0u;
          m2 |= (e[q] > 0.0f) ? bit : 0u;
          mo |= ((ho_mask >> q) & 1u) ? bit : 0u;
          mc |= ((hc_mask >> q) & 1u) ? bit : 0u;
        }
        m1 |= __shfl_xor(m1, 32); m2 |= __shfl_xor(m2, 32); mo |= __shfl_xor(mo, 32); mc |= __shfl_xor(mc, 32);
        if (valid && on && h == 0) {
          uint32_t* d = mp.relu_dump + (rid * S + s) * 5;
          d[0] = m1; d[1] = m2; d[2] = mo; d[3] = mc; d[4] = 1u;
        }
      }
      LP_SCHED_FENCE();
      if (want_params) {
        tile_store_fm(xt, r, h, ho);
        tile_store_fm(yt, r, h, hc);
      }
      LP_SCHED_FENCE();
    }

    // ---------------- compositing, backward: wave scans along the ray ----------------
    const int sc = on ? s : S - 1;
    const float depth_prev = PLAIN ? ray.near_t + lin01((sc > 0) ? sc - 1 : 0, S) * (ray.far_t - ray.near_t)
                                   : sample_depth_tab((sc > 0) ? sc - 1 : 0, a.march, ray.near_t, ray.far_t, sm + M::INF);
    const float delta = (sc == 0) ? delta0 : depth - depth_prev;
    float raw = hd.raw_o;
    if (!PLAIN && a.noise_sigma > 0.0f) raw = raw + sample_noise(rid, sc, a.rays.n_rays, S, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    const float od = on ? opacity * delta : 0.0f;
    float n_hi = 0.0f, n_lo = 0.0f;  // -log T behind the previous block: the forward's checkpoint (a float pair)
    if (bs > 0) {
      const float2 c2 = *reinterpret_cast<const float2*>(a.neg_log_t_ckpt + (rid * n_ckpt + bs - 1) * 2);
      n_hi = c2.x;
      n_lo = c2.y;
    }
    const float incl = tm_scan_up(od, r);
    float nlt_s = n_hi + (incl + n_lo);           // after sample s
    float nlt_p = n_hi + ((incl - od) + n_lo);    // before it
    if (!(nlt_s > 0.0f)) nlt_s = 0.0f;
    if (!(nlt_p > 0.0f)) nlt_p = 0.0f;
    const float t_i = __expf(-nlt_s), t_im1 = __expf(-nlt_p);
    const float w = t_im1 - t_i;
    float sg[4];
    float p_i = g_len * depth;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      sg[c] = (c < NC) ? sigmoid_f(hd.raw_c[c]) : 0.0f;
      if (c < NC) p_i = fmaf(gfeat[c], sg[c] * occ, p_i);
    }
    p_i = on ? p_i : 0.0f;
    float p_up = __shfl_down(p_i, 1, 32);          // p of the next (farther) sample
    p_up = (r == 31) ? p_carry : p_up;
    const float term = on ? t_i * (p_i - p_up) : 0.0f;
    const float suffix = suffix_carry + tm_scan_down(term, r);
    const float d_a = suffix + g_nlt;
    // carries for the block in front (wave-uniform: lane 0 of either half)
    const float next_suffix_carry = __shfl(suffix, 0, 32);
    const float next_p_carry = __shfl(p_i, 0, 32);
    const bool contrib = valid && on;
    const float dro = contrib ? d_a * delta * a.gain * occ * d_softplus_f(raw) : 0.0f;
    float drc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) drc[c] = (c < NC && contrib) ? w * gfeat[c] * occ * sg[c] * (1.0f - sg[c]) : 0.0f;

    // ---------------- output layers of the heads (VALU) ----------------
    float dhc[16];
    {
      const float* wc2 = sm + M::WC2 + 16 * h + opaque_zero();
#pragma unroll
      for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 4 * j + i;
          const float4 wc = *reinterpret_cast<const float4*>(wc2 + (8 * j + i) * 4);
          float v = drc[0] * wc.x;
          v = fmaf(drc[1], wc.y, v);
          v = fmaf(drc[2], wc.z, v);
          if (NC > 3) v = fmaf(drc[3], wc.w, v);
          dhc[q] = mask_apply(hc_mask, q, v);
        }
        LP_SCHED_FENCE();
      }
    }
    if (h == 0) {
      dbo2 += dro;
#pragma unroll
      for (int c = 0; c < NC; ++c) dbc2[c] += drc[c];
    }
    if (want_params) {
      if (h == 0) {
        ts[r] = dro;
#pragma unroll
        for (int c = 0; c < NC; ++c) ts[(1 + c) * 32 + r] = drc[c];
      }
      const float* xf = xt + r * T_LD + 16 * h;
      const float* yf = yt + r * T_LD + 16 * h;
      const float* tf = ts + 16 * h;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 hov = *reinterpret_cast<const float4*>(xf + 4 * i);
        const float4 hcv = *reinterpret_cast<const float4*>(yf + 4 * i);
        const float4 d0 = *reinterpret_cast<const float4*>(tf + 4 * i);
        dwo2 = fmaf(hov.x, d0.x, dwo2); dwo2 = fmaf(hov.y, d0.y, dwo2);
        dwo2 = fmaf(hov.z, d0.z, dwo2); dwo2 = fmaf(hov.w, d0.w, dwo2);
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const float4 dc = *reinterpret_cast<const float4*>(tf + (1 + c) * 32 + 4 * i);
          dwc2[c] = fmaf(hcv.x, dc.x, dwc2[c]); dwc2[c] = fmaf(hcv.y, dc.y, dwc2[c]);
          dwc2[c] = fmaf(hcv.z, dc.z, dwc2[c]); dwc2[c] = fmaf(hcv.w, dc.w, dwc2[c]);
        }
        LP_SCHED_FENCE();
      }
    }
    LP_SCHED_FENCE();

    // ---------------- colour hidden layer: X = e + enc (the ray encoding is wave-uniform here) ----------------
    __builtin_amdgcn_s_setprio(1);
    f32x16 acc = (f32x16){0};
    {
      if (want_params) {
        float xe[16];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 v = *reinterpret_cast<const float4*>(recz + 32 + 16 * h + 4 * i);
          xe[4 * i] = e[4 * i] + v.x; xe[4 * i + 1] = e[4 * i + 1] + v.y; xe[4 * i + 2] = e[4 * i + 2] + v.z; xe[4 * i + 3] = e[4 * i + 3] + v.w;
        }
        limb_tile_store<2>(xrow, xe);
      }
      acc = layer_dxv<2, 2>(Ab(I3{}), lane, dhc, acc, want_params ? yrow : nullptr);
#pragma unroll
      for (int q = 0; q < 16; ++q) dsum[q] += dhc[q];
      if (want_params) {
        lds_barrier();
        dq_c1 = dw_quadrant_bf<B::PER_WAVE * 4>(wave0b, xq_off, yq_off, 0, 4, dq_c1, dq_b, onehot(0));
        lds_barrier();
      }
    }
    LP_SCHED_FENCE();
    // ---------------- opacity hidden layer: X = e ----------------
    {
      float dho[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 wo = *reinterpret_cast<const float4*>(ldz + M::WO2 + 8 * j + 4 * h);
        dho[4 * j + 0] = mask_apply(ho_mask, 4 * j + 0, dro * wo.x);
        dho[4 * j + 1] = mask_apply(ho_mask, 4 * j + 1, dro * wo.y);
        dho[4 * j + 2] = mask_apply(ho_mask, 4 * j + 2, dro * wo.z);
        dho[4 * j + 3] = mask_apply(ho_mask, 4 * j + 3, dro * wo.w);
      }
      if (want_params) limb_tile_store<2>(xrow, e);
      acc = layer_dxv<2, 2>(Ab(I2{}), lane, dho, acc, want_params ? yrow : nullptr);
      if (want_params) {
        lds_barrier();
        dq_o1 = dw_quadrant_bf<B::PER_WAVE * 4>(wave0b, xq_off, yq_off, 0, 4, dq_o1, dq_b, onehot(1));
        lds_barrier();
      }
    }
    float de[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) de[q] = (e[q] > 0.0f) ? acc[q] : 0.0f;
    LP_SCHED_FENCE();
    // ---------------- trunk layer 2 ----------------
    float dh1[16];
    {
      if (want_params) limb_tile_store<2>(xrow, h1);
      acc = layer_dxv<2, 2>(Ab(I1{}), lane, de, (f32x16){0}, want_params ? yrow : nullptr);
      if (want_params) {
        lds_barrier();
        dq_t2 = dw_quadrant_bf<B::PER_WAVE * 4>(wave0b, xq_off, yq_off, 0, 4, dq_t2, dq_b, onehot(2));
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) dh1[q] = (h1[q] > 0.0f) ? acc[q] : 0.0f;
      if (want_params) lds_barrier();
    }
    LP_SCHED_FENCE();
    // ---------------- trunk layer 1 ----------------
    {
      if (want_params) limb_tile_store<C / 16>(xrow, x0);
      if (gg) {
        acc = layer_dxv<2, 2>(Ab(I0{}), lane, dh1, (f32x16){0}, want_params ? yrow : nullptr);
      } else if (want_params) {
        limb_tile_store<2>(yrow, dh1);
      }
      if (want_params) {
        lds_barrier();
        dq_t1 = dw_quadrant_bf<B::PER_WAVE * 4>(wave0b, (C == 16) ? xq_off - 32 * mi : xq_off, yq_off, t1_v0, t1_v1, dq_t1, dq_b, onehot(3));
        lds_barrier();
      }
    }
    if (gg) {
#pragma unroll
      for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * DX_LD + r] = acc[q];
    }
    LP_SCHED_FENCE();
    __builtin_amdgcn_s_setprio(0);
    const bool live = contrib && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));  // a masked sample scatters nothing

    // ---------------- the ray is done: d enc = W_c1 D, D = the ray's sum of d hc over lanes and blocks ----------------
    suffix_carry = next_suffix_carry;
    p_carry = next_p_carry;
    if (bs == 0) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        float v = dsum[q];
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 32);
        dsum[q] = v;
      }
      const f32x16 de_acc = layer_bf3v<2>(ARowsBwd{rimg + R::L_C1, R::ST_32, 31}, lane, dsum, (f32x16){0});
      if (valid && a.grad_encoding && r == 0) {
        float4* dst = reinterpret_cast<float4*>(a.grad_encoding + ray_id * HID + 4 * h);
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[2 * j] = make_float4(de_acc[4 * j], de_acc[4 * j + 1], de_acc[4 * j + 2], de_acc[4 * j + 3]);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) dsum[q] = 0.0f;
      k += 1;
      bs = n_blk - 1;
      new_ray = true;
      ray_id = ray0 + k;
      valid = ray_id < a.rays.n_rays;
      rid = valid ? ray_id : 0;
      if (k < rpw) ray = load_ray(a.rays, rid);
    } else {
      bs -= 1;
      new_ray = false;
    }
    // ---------------- next iteration's samples + grid gradient of this one ----------------
    if (it + 1 < n_it) fetch_sample<C, GM, true, PLAIN>(a, sm, ray, sample_of(bs), h, nx);
    LP_SCHED_FENCE();
    if (gg && !(mp.dbg & 2)) {
      if constexpr (GM == GM_TRIPLANE) {
        scatter_triplane<C>(a.grad_grid_list, a.grid, b_cur, x, y, z, live, lane, xt, yt, mp.dbg);
      } else {
        const int ng = (GM == GM_VOXEL) ? 1 : a.grid.n_grids;
#pragma unroll 1
        for (int g = 0; g < ng; ++g)
          scatter_grid<C, GM>(a.grad_grid_list[g], a.grid.grids[g], b_cur, x, y, z, live, lane, xt, yt, mp.dbg);
      }
    }
  }

  // ---------------- epilogue: output layers, biases, this wave's dW quadrants ----------------
  if (want_params) {
    float* G = a.grad_mlp_params;
    const int j = lane & 31;
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
    const int col = 16 * ni + m16;  // 16x16x32 accumulator: column = lane & 15, rows 4 (lane >> 4) + i
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int prow = 4 * ka + i;
      atomic_add_f32(G + mp.w_t2 + (16 * mi + prow) * HID + col, dq_t2[i]);
      atomic_add_f32(G + mp.w_o1 + (16 * mi + prow) * HID + col, dq_o1[i]);
      atomic_add_f32(G + mp.w_c1 + (16 * mi + prow) * HID + col, dq_c1[i]);
      const int row1 = (C == 16) ? prow : 16 * mi + prow;
      if (row1 < C) atomic_add_f32(G + mp.w_t1 + row1 * HID + col, dq_t1[i]);
    }
    if (ka == 0) {  // lanes 0..15: rows 0..3 of the one-hot products = the bias gradients of c1, o1, t2, t1
      if (mi == 0) {
        atomic_add_f32(G + mp.b_t2 + col, dq_b[2]);
        atomic_add_f32(G + mp.b_o1 + col, dq_b[1]);
        atomic_add_f32(G + mp.b_c1 + col, dq_b[0]);
      }
      if (C == 16 || mi == 0) atomic_add_f32(G + mp.b_t1 + col, dq_b[3]);
    }
  }
}

template <int C, int GM, int NC, bool PLAIN, bool DUMP>
static int launch_tm(const LpRendererArgs& a, const MfmaParams& mp_, hipStream_t stream) {
  // rays per wave: 32 once the batch fills one round of resident workgroups (2 per CU = 512); a smaller batch is spread over the chip
  // with fewer rays per wave (every workgroup pays the weight staging and the dW flush once).  Measured on 65 536 random rays
  // (refbench256, fwd + bwd): 32 rays per wave 7.15 ms, 16: 7.52, 8: 7.78, 4: 8.86 (profiles/r06_transposed_march.txt)
  MfmaParams mp = mp_;
  const int rpw = renderer_tm_rays_per_wave(a, 512);
  mp.tm_rpw = rpw;
  constexpr size_t lds = (size_t)TmLds<C>::TOTAL;
  const hipError_t e = hipFuncSetAttribute((const void*)renderer_bwd_bf3_tm<C, GM, NC, PLAIN, DUMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const unsigned nb = (unsigned)((a.rays.n_rays + WAVES * rpw - 1) / (WAVES * rpw));
  hipLaunchKernelGGL((renderer_bwd_bf3_tm<C, GM, NC, PLAIN, DUMP>), dim3(nb), dim3(256), lds, stream, a, mp);
  return LP_OK;
}
template <int C, int GM, bool PLAIN>
static int launch_tm_nc(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
#ifdef LP_TEST_HOOKS
  if (mp.relu_dump) return a.color_chn <= 3 ? launch_tm<C, GM, 3, PLAIN, true>(a, mp, stream) : launch_tm<C, GM, 4, PLAIN, true>(a, mp, stream);
#else
  if (mp.relu_dump) return set_error(LP_EUNSUPPORTED, "relu dump: this library was built without -DLP_TEST_HOOKS (no DUMP twins)");
#endif
  return a.color_chn <= 3 ? launch_tm<C, GM, 3, PLAIN, false>(a, mp, stream) : launch_tm<C, GM, 4, PLAIN, false>(a, mp, stream);
}
template <int C>
static int launch_tm_gm(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  const bool plain = bwd_is_plain(a);
  switch (gm) {
    case GM_TRIPLANE: return plain ? launch_tm_nc<C, GM_TRIPLANE, true>(a, mp, stream) : launch_tm_nc<C, GM_TRIPLANE, false>(a, mp, stream);
    case GM_VOXEL: return plain ? launch_tm_nc<C, GM_VOXEL, true>(a, mp, stream) : launch_tm_nc<C, GM_VOXEL, false>(a, mp, stream);
    default: return plain ? launch_tm_nc<C, GM_GENERIC, true>(a, mp, stream) : launch_tm_nc<C, GM_GENERIC, false>(a, mp, stream);
  }
}

// the transposed march covers: no beyond-far samples (<= 64 would fit the table, but their -log T jumps by orders of magnitude per
// sample: the in-block scan is not the place for them), no early termination, checkpoints present, at least one full block of samples.  (A small batch's forward may have
// marched segments in parallel -- seg_prefix -- which this backward does not need: it deals a small batch over the chip by rays per
// wave, and every ray's samples lie in one wave.)
bool renderer_bwd_tm_supported(const LpRendererArgs& a) {
  return renderer_tm_eligible(a) && a.neg_log_t_ckpt != nullptr && a.arithmetic == LP_ARITH_DEFAULT;
}

int renderer_bwd_bf3_tm_launch(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
  return a.grid.channels == 16 ? launch_tm_gm<16>(a, mp, gm, stream) : launch_tm_gm<32>(a, mp, gm, stream);
}

}  // namespace lp
