// lp_renderer_loop.h -- device code of the LAYER-LOOPED bf16x3 MFMA family of the Renderer: any decoder depth the reference sweeps.
//
// The tuned kernels (lp_renderer_mfma*.hip) are written for one decoder shape (trunk / opacity / colour of 2 layers, one
// hidden layer each next to the output layers) and its "flex" subsets.  The reference generates a kernel for ANY layer
// counts (triton_src/__init__.py:82-125) and its own sweep uses 2 or 4 layers per MLP
// (tests/test_renderer_with_autograd.py:49-51).  This family covers those shapes on the matrix cores:
//   trunk 1..MT layers (none with a separate colour grid-list), opacity / colour heads of 1..MH+1 layers (MH hidden layers +
//   the output layer), ONE hidden width H in {16, 32} (NB = 1; 16 is staged zero-padded) or 64 (NB = 2 blocks of 32
//   features), grid channels 16 / 32, <= 4 colour channels.
// Same arithmetic and data layout as the bf16x3 kernels of the default shape (lp_bf3.h): one wave = 32 rays, lane (h, r) owns
// ray r and 16 of every 32 features, every layer is the six limb products of v_mfma_f32_32x32x16_bf16 on row-major limb
// images in LDS (one image per 32 x 32 block of a weight matrix, read transposed by the forward and plainly by the dX
// chains), weight gradients are shared by the four waves of a workgroup through fp32 tiles and v_mfma_f32_16x16x4_f32
// quadrants (lp_renderer_mfma_bwd.hip), the grid gradient leaves through the run-merged scatter (lp_mfma_common.h).
// What is different: the layers of an MLP are a LOOP (unrolled to the family's maximum with wave-uniform guards, so every
// activation keeps a compile-time register name), the backward keeps every hidden activation of the recompute (up to
// (4 + 3 + 3) x 16 registers) and therefore runs at one wave per SIMD with the 512-register budget.
#pragma once
#include <type_traits>

#include "lp_host.h"
#include "lp_loop.h"

namespace lp {

constexpr int LOOP_MAX_T = 4;       // trunk layers
constexpr int LOOP_MAX_H = 3;       // hidden layers of a head (its output layer comes on top)

struct LoopParams {
  int n_t, n_o, n_c;   // layers on the matrix cores: every trunk layer, the hidden layers of the heads
  LoopLayer t[LOOP_MAX_T], o[LOOP_MAX_H], c[LOOP_MAX_H];
  LoopLayer co;                     // WC (5..32 colour channels): the colour output layer, on the matrix cores as well
  int64_t w_o2, b_o2, w_c2, b_c2;   // output layers of the heads
  int ldc2;                         // row stride of the colour output layer (padded colour width)
  int hid, hin;                     // hidden width; input width of the heads (= width of the ray encoding)
  int ho_w, hc_w;                   // input widths of the two output layers
  int wo2, wc2, hb, inf;            // float offsets inside the small block
  int img_end;                      // bytes before the per-wave tiles
  int dbg;
  // segment-parallel march of a small batch (LpRendererArgs.seg_prefix): LP_SEG_LEN-sample blocks per workgroup;
  // seg_fwd: this forward launch marches segments (segment-local state records, chained by renderer_fwd_combine)
  int seg_blocks, seg_fwd;
  // test hook (lp_renderer_backward_relu_dump): ReLU decisions of the backward's recompute, [ray][sample][dump_words] words -- NB words
  // per ReLU site in the reference's evaluation order, then the visited flag (include/lightplane_hip.h).  Only the DUMP twins read it.
  uint32_t* relu_dump;
  int dump_words;
  // per-wave LDS area of the backward: floats per wave, and the byte distance from the dY tile to the third tile of two-block layers
  // (0 = none: the decoder's images leave no room), see loop_layer_bwd
  int tile_stride, z_delta;
};

// per-wave LDS area behind the images (floats)
struct LoopTile {
  static constexpr int XT = 0;                 // X tile [32][36] (also: dx0 tile of the scatter)
  static constexpr int YT = 32 * LT_LD;        // dY tile [32][36] (also: the scatter's weight table)
  static constexpr int TS = 2 * 32 * LT_LD;    // [5][32]: d raw_o, d raw_c[0..3] by ray
  static constexpr int WT = TS + 5 * 32;       // [8][32]: the scatter's weight table when C = 64 (its dx0 tile [64][36] spans X and dY)
  static constexpr int PER_WAVE = WT + 8 * 32;
  // third tile of the two-block backward (hidden 64, bf16 quadrants): starts at TS -- TS and WT are idle during the layer rounds -- and
  // ends Z_EXTRA floats behind PER_WAVE
  static constexpr int ZT = TS;
  static constexpr int Z_EXTRA = ZT + 2 * rm_bytes(32) / 4 - PER_WAVE;
};


template <int NB>
LP_DEV void loop_stage(const LpRendererArgs& a, const LoopParams& lp, float* lds) {
  const float* P = a.mlp_params;
  const int tid = threadIdx.x;
  char* b = reinterpret_cast<char*>(lds);
  for (int l = 0; l < lp.n_t; ++l) loop_stage_layer<NB>(b, lds, P, lp.t[l], tid);
  for (int l = 0; l < lp.n_o; ++l) loop_stage_layer<NB>(b, lds, P, lp.o[l], tid);
  for (int l = 0; l < lp.n_c; ++l) loop_stage_layer<NB>(b, lds, P, lp.c[l], tid);
  if (a.color_chn > 4) loop_stage_layer<NB>(b, lds, P, lp.co, tid);
  for (int i = tid; i < 32 * NB; i += (int)blockDim.x) {
    lds[lp.wo2 + i] = (i < lp.ho_w) ? P[lp.w_o2 + i] : 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) lds[lp.wc2 + i * 4 + c] = (i < lp.hc_w && c < a.color_chn) ? P[lp.w_c2 + (int64_t)i * lp.ldc2 + c] : 0.0f;
  }
  for (int i = tid; i < LOOP_N_INF; i += (int)blockDim.x) lds[lp.inf + i] = (i < a.march.num_samples_inf) ? inf_scale(i, a.march) : 0.0f;
  if (tid == 0) {
    lds[lp.hb] = P[lp.b_o2];
#pragma unroll
    for (int c = 0; c < 4; ++c) lds[lp.hb + 1 + c] = (c < a.color_chn) ? P[lp.b_c2 + c] : 0.0f;
  }
}


// output layers of the heads on the VALU (N = 1 and N <= 4): each lane covers its 16 features per block, its partner lane
// (l ^ 32) the other 16
template <int NB>
LP_DEV Heads loop_heads_forward(const float* sm, const LoopParams& lp, int h, const float (&ho)[NB][16], const float (&hc)[NB][16]) {
  float po = 0.0f, pc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 wo = *reinterpret_cast<const float4*>(sm + lp.wo2 + 32 * blk + 8 * j + 4 * h);
      const float wov[4] = {wo.x, wo.y, wo.z, wo.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int q = 4 * j + i;
        po = fmaf(ho[blk][q], wov[i], po);
        const float4 wc = *reinterpret_cast<const float4*>(sm + lp.wc2 + (32 * blk + 8 * j + 4 * h + i) * 4);
        pc[0] = fmaf(hc[blk][q], wc.x, pc[0]);
        pc[1] = fmaf(hc[blk][q], wc.y, pc[1]);
        pc[2] = fmaf(hc[blk][q], wc.z, pc[2]);
        pc[3] = fmaf(hc[blk][q], wc.w, pc[3]);
      }
    }
  }
  Heads o;
  o.raw_o = (po + __shfl_xor(po, 32)) + sm[lp.hb];
#pragma unroll
  for (int c = 0; c < 4; ++c) o.raw_c[c] = (pc[c] + __shfl_xor(pc[c], 32)) + sm[lp.hb + 1 + c];
  return o;
}

// WC: only the opacity head ends on the VALU
template <int NB>
LP_DEV float loop_opacity_forward(const float* sm, const LoopParams& lp, int h, const float (&ho)[NB][16]) {
  float po = 0.0f;
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 wo = *reinterpret_cast<const float4*>(sm + lp.wo2 + 32 * blk + 8 * j + 4 * h);
      po = fmaf(ho[blk][4 * j + 0], wo.x, po);
      po = fmaf(ho[blk][4 * j + 1], wo.y, po);
      po = fmaf(ho[blk][4 * j + 2], wo.z, po);
      po = fmaf(ho[blk][4 * j + 3], wo.w, po);
    }
  }
  return (po + __shfl_xor(po, 32)) + sm[lp.hb];
}

// this lane's features of the ray encoding, NB blocks (features >= width read as 0)
template <int NB>
LP_DEV void loop_load_encoding(const LpRendererArgs& a, int64_t rid, int h, int width, float (&enc)[NB][16]) {
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
    const float4* src = reinterpret_cast<const float4*>(a.rays.encoding + rid * width + 32 * blk + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = (32 * blk + 8 * j + 4 * h < width) ? src[2 * j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      enc[blk][4 * j + 0] = v.x; enc[blk][4 * j + 1] = v.y; enc[blk][4 * j + 2] = v.z; enc[blk][4 * j + 3] = v.w;
    }
  }
}

// sampled feature x0 [C/2 registers] -> NB blocks (zero-padded); RELU: the two-grid decoder's heads read relu(sample)
// (C = 64: register 16 blk + q of the gather holds channel 32 blk + featq(q, h), as in the MLP-Splatter's sloop_input)
template <int C, int NB, bool RELU>
LP_DEV void loop_pad_input(const float (&x0)[C / 2], float (&out)[NB][16]) {
  static_assert(C <= 32 * NB, "a 64-channel grid needs the two-block instantiation");
#pragma unroll
  for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool in = 16 * blk + q < C / 2;
      const float v = in ? x0[in ? 16 * blk + q : 0] : 0.0f;
      out[blk][q] = RELU ? relu_f(v) : v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------------
// WC: 5..32 colour channels -- the colour output layer runs on the matrix cores like a hidden layer (without ReLU), lane
// (h, r) composites the 16 channels feat(q, h) of its ray
// GM: GM_TRIPLANE (canonical triplane: shared axis computations, border re-expression in the scatter) or GM_GENERIC
// NW: waves per workgroup.  4, or 8 for decoders whose weight images exclude a second four-wave workgroup per CU (2/2/2 x 64: 97 KB):
// the forward needs no per-wave LDS, so EIGHT waves can share one copy of the images and the SIMDs still host two waves each
// (the staging loops stride by 256 threads: the upper four waves re-write what the lower four write, same values)
template <int C, int NB, bool TG, bool WC = false, int GM = GM_GENERIC, int NW = WAVES>
// (the two-block instantiations are compiled for two waves per SIMD as well: the triplane form needed 215 + 48 registers -- seven too
// many -- and ran the reference example's 1/1/2 x 64 decoder, whose 41 KB of images allow two workgroups per CU, at one wave per
// SIMD: forward 1.85 -> 0.98 ms with the bound (211 VGPRs, no scratch; 64 grid channels: 256 VGPRs, 10 spilled); decoders whose
// images exclude a second workgroup keep one wave per SIMD whatever the bound says)
__global__ void __launch_bounds__(64 * NW, 2) renderer_fwd_loop(const LpRendererArgs a, const LoopParams lp) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  loop_stage<NB>(a, lp, lds);
  __syncthreads();
  const float* const geo = lds + lp.inf - Lds::INF;  // sample_geometry() reads its beyond-far table at geo + Lds::INF
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int h = lane >> 5, r = lane & 31;
  // small batch (lp.seg_fwd): a workgroup marches ONE segment of its 128 rays from transmittance 1 and leaves segment-local
  // state records; renderer_fwd_combine chains them (lp_renderer_mfma.hip)
  const bool segf = !WC && lp.seg_fwd != 0;
  const int seg_len = LP_SEG_LEN * lp.seg_blocks;
  const int n_seg = segf ? (a.march.num_samples + seg_len - 1) / seg_len : 1;
  const int blk = segf ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
  const int seg = segf ? (int)blockIdx.x - blk * n_seg : 0;
  const int64_t ray_id = ((int64_t)blk * NW + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[NB][16];
  loop_load_encoding<NB>(a, rid, h, lp.hin, enc);
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const int n_ckpt = ckpt_count(a.march);
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;
  float nlt = 0.0f, nlt_lo = 0.0f, t_prev = 1.0f, len = 0.0f, depth_prev = 0.0f;
  int s_last = s_tot - 1;
  constexpr int NCH = WC ? 16 : 4;  // colour channels this lane composites
  float facc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) facc[c] = 0.0f;
  Sample<C> nx;
  const int s_lo = segf ? seg * seg_len : 0;
  const int s_hi = segf ? ((s_lo + seg_len < s_tot) ? s_lo + seg_len : s_tot) : s_tot;
  if (segf && s_lo > 0) {  // interval length of the segment's first sample
    sample_geometry<C>(a, geo, ray, s_lo - 1, nx);
    depth_prev = nx.depth;
  }
  for (int s = s_lo; s < s_hi; ++s) {
    fetch_sample<C, GM, true>(a, geo, ray, s, h, nx);
    const float depth = nx.depth, occ = nx.occ;
    const int zo = opaque_zero();
    const char* lbase = reinterpret_cast<const char*>(lds) + zo;
    const float* sm = lds + zo;
    float cur[NB][16], ho[NB][16], hc[NB][16];
    loop_pad_input<C, NB, TG>(nx.x0, cur);
#pragma unroll
    for (int l = 0; l < LOOP_MAX_T; ++l) {
      if (!TG && l < lp.n_t) {
        float nxt[NB][16];
        loop_layer_fwd<NB>(lbase, sm, lp.t[l], lane, cur, nxt);
        loop_copy<NB>(nxt, cur);
      }
    }
    // colour head input: trunk output (two-grid decoder: relu(sampled colour feature)) + ray encoding
    float cin[NB][16];
    if (TG) {
      float xc0[C / 2];
      gather_list<C, false>(a.color_grid, a.march.mask_out_of_bounds != 0, ray, nx.x, nx.y, nx.z, h, xc0);
      loop_pad_input<C, NB, true>(xc0, cin);
    } else {
      loop_copy<NB>(cur, cin);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) cin[b][q] += enc[b][q];
    }
    loop_copy<NB>(cur, ho);
#pragma unroll
    for (int l = 0; l < LOOP_MAX_H; ++l) {
      if (l < lp.n_o) {
        float nxt[NB][16];
        loop_layer_fwd<NB>(lbase, sm, lp.o[l], lane, ho, nxt);
        loop_copy<NB>(nxt, ho);
      }
    }
    loop_copy<NB>(cin, hc);
#pragma unroll
    for (int l = 0; l < LOOP_MAX_H; ++l) {
      if (l < lp.n_c) {
        float nxt[NB][16];
        loop_layer_fwd<NB>(lbase, sm, lp.c[l], lane, hc, nxt);
        loop_copy<NB>(nxt, hc);
      }
    }
    float raw, raw_c[NCH];
    if constexpr (WC) {
      raw = loop_opacity_forward<NB>(sm, lp, h, ho);
      float cv[NB][16];
      loop_layer_fwd<NB, false>(lbase, sm, lp.co, lane, hc, cv);
#pragma unroll
      for (int c = 0; c < 16; ++c) raw_c[c] = cv[0][c];
    } else {
      const Heads hd = loop_heads_forward<NB>(sm, lp, h, ho, hc);
      raw = hd.raw_o;
#pragma unroll
      for (int c = 0; c < 4; ++c) raw_c[c] = hd.raw_c[c];
    }
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    depth_prev = depth;
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
    for (int c = 0; c < NCH; ++c) facc[c] = fmaf(w, sigmoid_f(raw_c[c]) * occ, facc[c]);
    // state records of the segment-parallel backward (absolute; segf: relative to the segment's start)
    if (!WC && a.seg_prefix && valid && h == 0 && (((s + 1) % LP_SEG_LEN) == 0 || s == a.march.num_samples - 1)) {
      float4* dst = reinterpret_cast<float4*>(a.seg_prefix + (ray_id * segment_count(a.march) + s / LP_SEG_LEN) * 8);
      dst[0] = make_float4(len, facc[0], facc[1], facc[2]);
      dst[1] = make_float4(facc[3], nlt, nlt_lo, 0.0f);
    }
    if (a.stop_neg_log_t > 0.0f && __ballot(valid && nlt < a.stop_neg_log_t) == 0) {
      s_last = s;
      break;
    }
  }
  if constexpr (WC) {
    if (valid) {  // both lanes of a ray write their channels; lane h = 0 the per-ray scalars (write_ray_outputs, lp_device.h)
      const bool epi = a.bg_color != nullptr || a.alpha != nullptr;
      const float T = epi ? expf(-nlt) : 0.0f;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int ch = featq(q, h);
        if (ch < a.color_chn) {
          float f = facc[q];
          if (a.bg_color) f = f + T * a.bg_color[ch];
          a.feature[ray_id * a.color_chn + ch] = f;
        }
      }
      if (h == 0) {
        a.ray_length[ray_id] = len;
        a.neg_log_t[ray_id] = nlt;
        if (a.alpha) a.alpha[ray_id] = (a.alpha_mode == 2) ? -nlt : 1.0f - T;
      }
    }
  } else if (valid && h == 0 && !segf) {
    write_ray_outputs(a, ray_id, len, nlt, facc);
  }
  if (valid && h == 0 && a.neg_log_t_ckpt && !segf)
    *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + n_ckpt - 1) * 2) = make_float2((float)s_last, nlt_lo);
}

// ---------------------------------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------------------------------
// MT / MH: the trunk layers / hidden head layers this instantiation holds registers for (the kernel's loops are unrolled to
// them; NB = 2 is instantiated for the 2 / 2 / 2 shape only: 64-wide activations are 32 registers each)
// DUMP (test hook, its own instantiations in lp_renderer_loop_dump.hip / lp_renderer_loop_shallow_dump.hip, built with the flags of
// their production twins): the ReLU decisions of the recompute are also written to lp.relu_dump -- same instruction sequence, stores
// added.
template <int C, int NB, bool TG, int MT, int MH, bool WC = false, int GM = GM_GENERIC, bool DUMP = false>
__global__ void __launch_bounds__(256, (NB == 1 && MT <= 2 && MH <= 1 && !WC) ? 2 : 1) renderer_bwd_loop(const LpRendererArgs a, const LoopParams lp) {
  using T = LoopTile;
  // gradient operand of the dX chains: two limbs (lp_bf3.h) where the chain is short -- at most two trunk layers and one hidden
  // layer per head, i.e. the shallow and the two-block (hidden 64) instantiations; deep decoders keep three: the per-layer error
  // compounds along a 7-layer chain (4/2/3 x 32 on 70 rays x 15 samples: grad_mlp_params 1.05e-4 with two limbs)
  constexpr int DXL = (MT <= 2 && MH <= 1) ? LP_DX_LIMBS : 3;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  loop_stage<NB>(a, lp, lds);
  const float* const geo = lds + lp.inf - Lds::INF;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int h = lane >> 5, r = lane & 31;
  float* const wave0 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + lp.img_end);
  const int PW = (NB == 2 && LP_LOOP_DW_BF16) ? lp.tile_stride : T::PER_WAVE;  // floats per wave
  const int zd = (NB == 2 && LP_LOOP_DW_BF16) ? lp.z_delta : 0;
  float* const wv = wave0 + wave * PW;
  float* const xt = wv + T::XT;
  float* const yt = wv + T::YT;
  float* const ts = wv + T::TS;
  // segment-parallel sweep of a small batch (LpRendererArgs.seg_prefix): workgroup = (128 rays, lp.seg_blocks blocks of
  // LP_SEG_LEN samples), see renderer_bwd_bf3 (lp_renderer_mfma_bwd.hip)
  const bool seg_on = !WC && a.seg_prefix != nullptr;
  const int n_rec = seg_on ? segment_count(a.march) : 1;
  const int seg_len = LP_SEG_LEN * lp.seg_blocks;
  const int n_seg = seg_on ? (a.march.num_samples + seg_len - 1) / seg_len : 1;
  const int blk = seg_on ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
  const int seg = seg_on ? (int)blockIdx.x - blk * n_seg : 0;
  const int64_t ray_id = ((int64_t)blk * WAVES + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
  float enc[NB][16];
  loop_load_encoding<NB>(a, rid, h, lp.hin, enc);
  // closing pair of the checkpoint list: last sample the forward marched for this wave (early termination) and the low word
  // of the final -log T.  The sample loop is workgroup-uniform (barriers): it starts at the largest index of the four waves.
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
  if (lane == 0) ts[0] = (float)s_last_w;
  __syncthreads();
  int s_begin = 0;
#pragma unroll
  for (int v = 0; v < WAVES; ++v) {
    const int sv = (int)wave0[v * PW + T::TS];
    s_begin = sv > s_begin ? sv : s_begin;
  }
  __syncthreads();  // ts[] is reused by the sample loop
  const int s_lo = seg_on ? seg * seg_len : 0;
  if (seg_on) s_begin = (s_lo + seg_len - 1 < s_tot - 1) ? s_lo + seg_len - 1 : s_tot - 1;

  float denc[NB][16];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int q = 0; q < 16; ++q) denc[b][q] = 0.0f;
  }
  constexpr int NCH = WC ? 16 : 4;  // colour channels of this lane: 0..3 (both lanes of a ray alike), or the 16 channels feat(q, h)
  float gfeat[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int ch = WC ? featq(c, h) : c;
    gfeat[c] = (valid && a.grad_feature && ch < a.color_chn) ? a.grad_feature[rid * a.color_chn + (ch < a.color_chn ? ch : 0)] : 0.0f;
  }
  const float g_len = (valid && a.grad_ray_length) ? a.grad_ray_length[rid] : 0.0f;
  float g_nlt;
  if constexpr (WC) {  // epilogue_grad_nlt (lp_device.h) with the background sum taken over both lanes of the ray
    g_nlt = (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f;
    if (a.bg_color != nullptr || a.grad_alpha != nullptr) {
      const float T = expf(-a.neg_log_t[rid]);
      if (a.grad_alpha && valid) {
        const float ga = a.grad_alpha[rid];
        g_nlt += (a.alpha_mode == 2) ? -ga : ga * T;
      }
      if (a.bg_color) {
        float sb = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int ch = featq(q, h);
          if (ch < a.color_chn) sb = fmaf(a.bg_color[ch], gfeat[q], sb);
        }
        sb += __shfl_xor(sb, 32);
        g_nlt -= T * sb;
      }
    }
  } else {
    g_nlt = epilogue_grad_nlt(a, rid, valid, a.neg_log_t[rid], (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f, gfeat, 4);
  }
  const bool want_params = a.grad_mlp_params != nullptr;
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;

  // dW quadrant of this wave inside every 32 x 32 block: rows 16 mi .., columns 16 ni ..; MFMA lane (m16, ka)
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
  LoopDw<NB> dw_t[MT], dw_o[MH], dw_c[MH], dw_co;
  loop_dw_zero<NB>(dw_co);
#pragma unroll
  for (int l = 0; l < MT; ++l) loop_dw_zero<NB>(dw_t[l]);
#pragma unroll
  for (int l = 0; l < MH; ++l) {
    loop_dw_zero<NB>(dw_o[l]);
    loop_dw_zero<NB>(dw_c[l]);
  }
  // output layers of the heads: lane (f = l & 31, half h) owns feature 32 blk + f, partial over the 16 rays of its half
  float dwo2[NB], dwc2[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    dwo2[b] = 0.0f;
#pragma unroll
    for (int c = 0; c < 4; ++c) dwc2[b][c] = 0.0f;
  }
  float dbo2 = 0.0f, dbc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool gg = a.grad_grid_list[0] != nullptr;
  const bool ggc = TG && a.grad_color_grid_list[0] != nullptr;

  float nlt = a.neg_log_t[rid];
  float suffix = 0.0f, p_next = 0.0f;
  if constexpr (!WC) {
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
  }
  Sample<C> nx;
  fetch_sample<C, GM, false>(a, geo, ray, s_begin, h, nx);
  for (int s = s_begin; s >= s_lo; --s) {
    const bool on = s <= s_last_w;  // wave-uniform; false only for samples a sibling wave still marches
    const float depth = nx.depth, occ = nx.occ, x = nx.x, y = nx.y, z = nx.z;
    const int zo = opaque_zero();
    const char* lbase = reinterpret_cast<const char*>(lds) + zo;
    const float* sm = lds + zo;

    // ---------------- forward recompute: every hidden activation is kept ----------------
    float xin[NB][16];          // input of the first layer(s): sampled feature (two-grid decoder: its relu)
    loop_pad_input<C, NB, TG>(nx.x0, xin);
    float tA[MT][NB][16];       // trunk activations (post-ReLU)
    float e[NB][16];            // trunk output = input of the heads
    loop_copy<NB>(xin, e);
#pragma unroll
    for (int l = 0; l < MT; ++l) {
      if (!TG && l < lp.n_t) {
        loop_layer_fwd<NB>(lbase, sm, lp.t[l], lane, e, tA[l]);
        loop_copy<NB>(tA[l], e);
      }
    }
    float xc[C / 2];            // two-grid decoder: sampled colour feature of this sample
    float cin[NB][16];          // input of the colour head
    if (TG) {
      gather_list<C, true>(a.color_grid, a.march.mask_out_of_bounds != 0, ray, x, y, z, h, xc);
      loop_pad_input<C, NB, true>(xc, cin);
    } else {
      loop_copy<NB>(e, cin);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) cin[b][q] += enc[b][q];
    }
    float oA[MH][NB][16], cA[MH][NB][16];
    float ho[NB][16], hc[NB][16];
    loop_copy<NB>(e, ho);
#pragma unroll
    for (int l = 0; l < MH; ++l) {
      if (l < lp.n_o) {
        loop_layer_fwd<NB>(lbase, sm, lp.o[l], lane, ho, oA[l]);
        loop_copy<NB>(oA[l], ho);
      }
    }
    loop_copy<NB>(cin, hc);
#pragma unroll
    for (int l = 0; l < MH; ++l) {
      if (l < lp.n_c) {
        loop_layer_fwd<NB>(lbase, sm, lp.c[l], lane, hc, cA[l]);
        loop_copy<NB>(cA[l], hc);
      }
    }
    float raw, raw_c[NCH];
    if constexpr (WC) {
      raw = loop_opacity_forward<NB>(sm, lp, h, ho);
      float cv[NB][16];
      loop_layer_fwd<NB, false>(lbase, sm, lp.co, lane, hc, cv);
#pragma unroll
      for (int c = 0; c < 16; ++c) raw_c[c] = cv[0][c];
    } else {
      const Heads hd = loop_heads_forward<NB>(sm, lp, h, ho, hc);
      raw = hd.raw_o;
#pragma unroll
      for (int c = 0; c < 4; ++c) raw_c[c] = hd.raw_c[c];
    }
    if constexpr (DUMP) {
      // sites in the reference's evaluation order (naive_renderer.py:328-501): single grid-list: trunk layers, opacity hidden
      // layers, colour hidden layers; two-grid decoder: relu(feature), opacity hidden layers, relu(colour feature), colour hidden
      // layers.  A post-ReLU activation is > 0 exactly where the unit is active -- the test every mask of this backward applies.
      uint32_t* const dsite = lp.relu_dump + (rid * (int64_t)s_tot + s) * lp.dump_words;
      int k = 0;
      auto put = [&](const float (&v)[NB][16]) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
          unsigned m = 0;
#pragma unroll
          for (int q = 0; q < 16; ++q) m |= (v[b][q] > 0.0f) ? (1u << featq(q, h)) : 0u;
          m |= __shfl_xor(m, 32);
          if (valid && h == 0) dsite[k * NB + b] = m;
        }
        ++k;
      };
      if (TG) put(xin);
#pragma unroll
      for (int l = 0; l < MT; ++l) {
        if (!TG && l < lp.n_t) put(tA[l]);
      }
#pragma unroll
      for (int l = 0; l < MH; ++l) {
        if (l < lp.n_o) put(oA[l]);
      }
      if (TG) {
        float xcp[NB][16];
        loop_pad_input<C, NB, true>(xc, xcp);
        put(xcp);
      }
#pragma unroll
      for (int l = 0; l < MH; ++l) {
        if (l < lp.n_c) put(cA[l]);
      }
      if (valid && h == 0) dsite[k * NB] = on ? 1u : 2u;
    }
    LP_SCHED_FENCE();

    // ---------------- compositing, backward ----------------
    const float depth_prev = sample_depth_tab((s > 0) ? s - 1 : 0, a.march, ray.near_t, ray.far_t, lds + lp.inf);
    const float delta = (s == 0) ? delta0 : depth - depth_prev;
    if (a.noise_sigma > 0.0f) raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    if (on && a.neg_log_t_ckpt) {
      const int ck = ckpt_index(s, a.march);
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
    float sg[NCH];
    float p_i = g_len * depth;
    if constexpr (WC) {
      float pc = 0.0f;
#pragma unroll
      for (int c = 0; c < 16; ++c) {
        sg[c] = sigmoid_f(raw_c[c]);
        pc = fmaf(gfeat[c], sg[c] * occ, pc);
      }
      p_i += pc + __shfl_xor(pc, 32);
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        sg[c] = sigmoid_f(raw_c[c]);
        p_i = fmaf(gfeat[c], sg[c] * occ, p_i);
      }
    }
    suffix = on ? fmaf(t_i, p_i - p_next, suffix) : suffix;
    p_next = on ? p_i : p_next;
    const float d_a = suffix + g_nlt;
    const bool contrib = valid && on;
    const float dro = contrib ? d_a * delta * a.gain * occ * d_softplus_f(raw) : 0.0f;
    float drc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) drc[c] = contrib ? w * gfeat[c] * occ * sg[c] * (1.0f - sg[c]) : 0.0f;

    // ---------------- output layers of the heads (VALU) ----------------
    if (h == 0) {
      dbo2 += dro;
      if constexpr (!WC) {
#pragma unroll
        for (int c = 0; c < 4; ++c) dbc2[c] += drc[c];
      }
    }
    if (want_params) {
      if (h == 0) {
        ts[r] = dro;
        if constexpr (!WC) {
#pragma unroll
          for (int c = 0; c < 4; ++c) ts[(1 + c) * 32 + r] = drc[c];
        }
      }
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
        // ho / hc of this block -> the (wave-private) tiles; lane (f = r, half h) reads the rays 16h .. 16h+15 of feature f
        loop_tile_store(xt, r, h, ho[blk]);
        if constexpr (!WC) loop_tile_store(yt, r, h, hc[blk]);
        const float* xf = xt + r * LT_LD + 16 * h;
        const float* yf = yt + r * LT_LD + 16 * h;
        const float* tf = ts + 16 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 hov = *reinterpret_cast<const float4*>(xf + 4 * i);
          const float4 hcv = *reinterpret_cast<const float4*>(yf + 4 * i);
          const float4 d0 = *reinterpret_cast<const float4*>(tf + 4 * i);
          dwo2[blk] = fmaf(hov.x, d0.x, dwo2[blk]); dwo2[blk] = fmaf(hov.y, d0.y, dwo2[blk]);
          dwo2[blk] = fmaf(hov.z, d0.z, dwo2[blk]); dwo2[blk] = fmaf(hov.w, d0.w, dwo2[blk]);
          if constexpr (!WC) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              const float4 dc = *reinterpret_cast<const float4*>(tf + (1 + c) * 32 + 4 * i);
              dwc2[blk][c] = fmaf(hcv.x, dc.x, dwc2[blk][c]); dwc2[blk][c] = fmaf(hcv.y, dc.y, dwc2[blk][c]);
              dwc2[blk][c] = fmaf(hcv.z, dc.z, dwc2[blk][c]); dwc2[blk][c] = fmaf(hcv.w, dc.w, dwc2[blk][c]);
            }
          }
          LP_SCHED_FENCE();
        }
        LP_SCHED_FENCE();
      }
    }
    // gradients of the heads' last hidden activations (masked by their ReLU where there is a hidden layer)
    float g[NB][16];   // running gradient of the head being back-propagated
    float de[NB][16];  // gradient of the heads' input e (both heads)
    if constexpr (WC) {
      // the colour output layer as a layer phase of its own: dW = hc^T d raw_c (workgroup-shared), d hc = W d raw_c
      __builtin_amdgcn_s_setprio(1);
      float dyc[NB][16];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) dyc[b][q] = (b == 0) ? drc[q] : 0.0f;
      }
      f32x16 dxc[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) dxc[b] = (f32x16){0};
      loop_layer_bwd<NB, DXL>(lbase, lp.co, lane, xt, yt, wave0, PW, a_off, b_off, want_params, true, hc, dyc, dw_co, dxc, zd);
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) g[b][q] = (lp.n_c == 0 || hc[b][q] > 0.0f) ? dxc[b][q] : 0.0f;
      }
    } else {
#pragma unroll
      for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int q = 4 * j + i;
            const float4 wc = *reinterpret_cast<const float4*>(sm + lp.wc2 + (32 * blk + 8 * j + 4 * h + i) * 4);
            float v = drc[0] * wc.x;
            v = fmaf(drc[1], wc.y, v);
            v = fmaf(drc[2], wc.z, v);
            v = fmaf(drc[3], wc.w, v);
            g[blk][q] = (lp.n_c == 0 || hc[blk][q] > 0.0f) ? v : 0.0f;
          }
        }
      }
    }
    LP_SCHED_FENCE();

    // ---------------- colour head, hidden layers last -> first ----------------
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int l = MH - 1; l >= 0; --l) {
      if (l < lp.n_c) {
        f32x16 dx[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) dx[b] = (f32x16){0};
        if (l > 0) {
          loop_layer_bwd<NB, DXL>(lbase, lp.c[l], lane, xt, yt, wave0, PW, a_off, b_off, want_params, true, cA[l > 0 ? l - 1 : 0], g, dw_c[l], dx, zd);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = (cA[l > 0 ? l - 1 : 0][b][q] > 0.0f) ? dx[b][q] : 0.0f;
          }
        } else {
          loop_layer_bwd<NB, DXL>(lbase, lp.c[0], lane, xt, yt, wave0, PW, a_off, b_off, want_params, true, cin, g, dw_c[0], dx, zd);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = dx[b][q];
          }
        }
      }
    }
    // g = d cin = d (e | relu(colour feature)) and d encoding
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) denc[b][q] += g[b][q];
    }
    if (TG) {
      // two-grid decoder: scatter d relu(colour feature) into the colour grid-list now, while the wave's tiles are idle
      // between two layer phases; the opacity branch then starts from zero
      if (ggc && !(lp.dbg & 2)) {
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * DX_LD + r] = (xc[q] > 0.0f) ? g[0][q] : 0.0f;
        const bool live_c = valid && on && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));
#pragma unroll 1
        for (int gi = 0; gi < a.color_grid.n_grids; ++gi)
          scatter_grid<C>(a.grad_color_grid_list[gi], a.color_grid.grids[gi], ray.b, x, y, z, live_c, lane, xt, yt, lp.dbg);
        __builtin_amdgcn_s_setprio(1);
      }
#pragma unroll
      for (int b = 0; b < NB; ++b) {
#pragma unroll
        for (int q = 0; q < 16; ++q) de[b][q] = 0.0f;
      }
    } else {
      loop_copy<NB>(g, de);
    }
    // ---------------- opacity head ----------------
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 wo = *reinterpret_cast<const float4*>(sm + lp.wo2 + 32 * blk + 8 * j + 4 * h);
        const float wov[4] = {wo.x, wo.y, wo.z, wo.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int q = 4 * j + i;
          g[blk][q] = (lp.n_o == 0 || ho[blk][q] > 0.0f) ? dro * wov[i] : 0.0f;
        }
      }
    }
#pragma unroll
    for (int l = MH - 1; l >= 0; --l) {
      if (l < lp.n_o) {
        f32x16 dx[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) dx[b] = (f32x16){0};
        if (l > 0) {
          loop_layer_bwd<NB, DXL>(lbase, lp.o[l], lane, xt, yt, wave0, PW, a_off, b_off, want_params, true, oA[l > 0 ? l - 1 : 0], g, dw_o[l], dx, zd);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = (oA[l > 0 ? l - 1 : 0][b][q] > 0.0f) ? dx[b][q] : 0.0f;
          }
        } else {
          loop_layer_bwd<NB, DXL>(lbase, lp.o[0], lane, xt, yt, wave0, PW, a_off, b_off, want_params, true, e, g, dw_o[0], dx, zd);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = dx[b][q];
          }
        }
      }
    }
    // d e of both heads, through the ReLU that produced e (the trunk's last layer, or relu(sample) of the two-grid decoder)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
#pragma unroll
      for (int q = 0; q < 16; ++q) g[b][q] = (e[b][q] > 0.0f) ? de[b][q] + g[b][q] : 0.0f;
    }
    // ---------------- trunk, last -> first ----------------
#pragma unroll
    for (int l = MT - 1; l >= 0; --l) {
      if (!TG && l < lp.n_t) {
        f32x16 dx[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) dx[b] = (f32x16){0};
        if (l > 0) {
          loop_layer_bwd<NB, DXL>(lbase, lp.t[l], lane, xt, yt, wave0, PW, a_off, b_off, want_params, true, tA[l > 0 ? l - 1 : 0], g, dw_t[l], dx, zd);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = (tA[l > 0 ? l - 1 : 0][b][q] > 0.0f) ? dx[b][q] : 0.0f;
          }
        } else {
          loop_layer_bwd<NB, DXL>(lbase, lp.t[0], lane, xt, yt, wave0, PW, a_off, b_off, want_params, gg, xin, g, dw_t[0], dx, zd);
#pragma unroll
          for (int b = 0; b < NB; ++b) {
#pragma unroll
            for (int q = 0; q < 16; ++q) g[b][q] = dx[b][q];
          }
        }
      }
    }
    // g = d (sampled feature) -> LDS [channel][ray] (the X tile is free behind the last barrier)
    if (gg) {  // (C = 64: rows 32 .. 63 of the [channel][ray] tile lie in the dY tile; the weight table moves behind the tiles)
#pragma unroll
      for (int q = 0; q < C / 2; ++q) xt[(32 * (q >> 4) + featq(q & 15, h)) * DX_LD + r] = g[q >> 4][q & 15];
    }
    LP_SCHED_FENCE();
    // ---------------- next (nearer) sample + grid gradient ----------------
    __builtin_amdgcn_s_setprio(0);
    const bool live = valid && on && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));
    if (s > s_lo) fetch_sample<C, GM, true>(a, geo, ray, s - 1, h, nx);
    LP_SCHED_FENCE();
    if (gg && !(lp.dbg & 2)) {
      const int ng = (GM == GM_TRIPLANE) ? 3 : a.grid.n_grids;
      float* const wtab = (C == 64) ? wv + T::WT : yt;
      if constexpr (GM == GM_TRIPLANE && C != 64) {  // all three planes in one call (its weight table needs 3 x 128 floats: the dY tile)
        scatter_triplane<C>(a.grad_grid_list, a.grid, ray.b, x, y, z, live, lane, xt, wtab, lp.dbg);
      } else {
#pragma unroll 1
        for (int gi = 0; gi < ng; ++gi)
          scatter_grid<C, GM>(a.grad_grid_list[gi], a.grid.grids[gi], ray.b, x, y, z, live, lane, xt, wtab, lp.dbg);
      }
    }
  }

  // ---------------- epilogue ----------------
  if (valid && a.grad_encoding && !seg_on) {
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      float4* dst = reinterpret_cast<float4*>(a.grad_encoding + ray_id * lp.hin + 32 * blk + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (32 * blk + 8 * j + 4 * h < lp.hin)
          dst[2 * j] = make_float4(denc[blk][4 * j], denc[blk][4 * j + 1], denc[blk][4 * j + 2], denc[blk][4 * j + 3]);
      }
    }
  } else if (valid && a.grad_encoding) {  // the segments of a ray add up (the caller zero-fills)
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      float* dst = a.grad_encoding + ray_id * lp.hin + 32 * blk + 4 * h;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (32 * blk + 8 * j + 4 * h < lp.hin) {
#pragma unroll
          for (int i = 0; i < 4; ++i) atomic_add_f32(dst + 8 * j + i, denc[blk][4 * j + i]);
        }
      }
    }
  }
  if (want_params) {
    float* G = a.grad_mlp_params;
    const int j = lane & 31;
#pragma unroll
    for (int blk = 0; blk < NB; ++blk) {
      const int f = 32 * blk + j;
      if (f < lp.ho_w) atomic_add_f32(G + lp.w_o2 + f, dwo2[blk]);
      if (!WC && f < lp.hc_w) {
        for (int c = 0; c < a.color_chn; ++c) atomic_add_f32(G + lp.w_c2 + (int64_t)f * lp.ldc2 + c, dwc2[blk][c]);
      }
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
      atomic_add_f32(G + lp.b_o2, v);
      const float cv[4] = {c0, c1, c2, c3};
      if (!WC) {
        for (int c = 0; c < a.color_chn; ++c) atomic_add_f32(G + lp.b_c2 + c, cv[c]);
      }
    }
    if (WC) loop_dw_flush<NB>(G, lp.co, dw_co, wave, lane);
#pragma unroll
    for (int l = 0; l < MT; ++l) {
      if (!TG && l < lp.n_t) loop_dw_flush<NB>(G, lp.t[l], dw_t[l], wave, lane);
    }
#pragma unroll
    for (int l = 0; l < MH; ++l) {
      if (l < lp.n_o) loop_dw_flush<NB>(G, lp.o[l], dw_o[l], wave, lane);
      if (l < lp.n_c) loop_dw_flush<NB>(G, lp.c[l], dw_c[l], wave, lane);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// launch helpers shared by the two translation units of the family (lp_renderer_loop.hip: forward kernels + the deep and
// the two-block backward instantiations, one wave per SIMD; lp_renderer_loop_shallow.hip: the backward of decoders of up to
// 2 trunk layers and one hidden layer per head at TWO waves per SIMD, compiled with the spill-avoiding switches of build.py)
// ---------------------------------------------------------------------------------------------------------------
template <typename K>
static int loop_set_lds(K kernel, size_t bytes) {
  const hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  return LP_OK;
}

static unsigned loop_blocks(const LpRendererArgs& a) {
  return (unsigned)((a.rays.n_rays + WAVES * RAYS_PER_WAVE - 1) / (WAVES * RAYS_PER_WAVE));
}

// (two-grid decoder: always the run-time-loop grid-list form -- with two gathers per sample its triplane form spills ~150
// registers; `if constexpr` inside a template keeps those instantiations from being compiled at all)
template <int C, int NB, bool TG, bool WC>
static int launch_fwd_loop(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
  int rc;
  if constexpr (NB == 2 && !TG && !WC) {
    // images that exclude a second four-wave workgroup per CU (> 80 KB): eight-wave workgroups, one per CU, two waves per SIMD
    static const bool no_nw8 = getenv("LP_LOOP_FWD_NW4") != nullptr;  // A/B
    if (lds > 80 * 1024 && !no_nw8 && !p.seg_fwd) {
      const unsigned nb8 = (unsigned)((a.rays.n_rays + 8 * RAYS_PER_WAVE - 1) / (8 * RAYS_PER_WAVE));
      if constexpr (C <= 32) {
        if (tri) {
          if ((rc = loop_set_lds(renderer_fwd_loop<C, NB, TG, WC, GM_TRIPLANE, 8>, lds))) return rc;
          hipLaunchKernelGGL((renderer_fwd_loop<C, NB, TG, WC, GM_TRIPLANE, 8>), dim3(nb8), dim3(512), lds, stream, a, p);
          return LP_OK;
        }
      }
      if ((rc = loop_set_lds(renderer_fwd_loop<C, NB, TG, WC, GM_GENERIC, 8>, lds))) return rc;
      hipLaunchKernelGGL((renderer_fwd_loop<C, NB, TG, WC, GM_GENERIC, 8>), dim3(nb8), dim3(512), lds, stream, a, p);
      return LP_OK;
    }
  }
  if constexpr (!TG && C <= 32) {  // (64 channels: the run-time grid-list form only)
    if (tri) {
      if ((rc = loop_set_lds(renderer_fwd_loop<C, NB, TG, WC, GM_TRIPLANE>, lds))) return rc;
      hipLaunchKernelGGL((renderer_fwd_loop<C, NB, TG, WC, GM_TRIPLANE>), dim3(nb), dim3(256), lds, stream, a, p);
      return LP_OK;
    }
  }
  if ((rc = loop_set_lds(renderer_fwd_loop<C, NB, TG, WC, GM_GENERIC>, lds))) return rc;
  hipLaunchKernelGGL((renderer_fwd_loop<C, NB, TG, WC, GM_GENERIC>), dim3(nb), dim3(256), lds, stream, a, p);
  return LP_OK;
}

template <int C, int NB, bool TG, int MT, int MH, bool WC, bool DUMP = false>
static int launch_bwd_loop(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
  int rc;
  if constexpr (!TG && C <= 32) {
    if (tri) {
      if ((rc = loop_set_lds(renderer_bwd_loop<C, NB, TG, MT, MH, WC, GM_TRIPLANE, DUMP>, lds))) return rc;
      hipLaunchKernelGGL((renderer_bwd_loop<C, NB, TG, MT, MH, WC, GM_TRIPLANE, DUMP>), dim3(nb), dim3(256), lds, stream, a, p);
      return LP_OK;
    }
  }
  if ((rc = loop_set_lds(renderer_bwd_loop<C, NB, TG, MT, MH, WC, GM_GENERIC, DUMP>, lds))) return rc;
  hipLaunchKernelGGL((renderer_bwd_loop<C, NB, TG, MT, MH, WC, GM_GENERIC, DUMP>), dim3(nb), dim3(256), lds, stream, a, p);
  return LP_OK;
}

// The instantiation table of the family's backward, shared by the production translation units and their DUMP twins.
// SHALLOW decoders (<= 2 trunk layers -- none with a colour grid --, heads with at most one hidden layer, hidden width 16 / 32,
// <= 4 colour channels) at two waves per SIMD: lp_renderer_loop_shallow.hip (+ _dump)
template <bool DUMP>
static int loop_bwd_table_shallow(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
  const bool tg = a.color_grid.n_grids > 0;
  if (a.grid.channels == 16) {
    if (tg) return launch_bwd_loop<16, 1, true, 1, 1, false, DUMP>(a, p, nb, lds, tri, stream);
    return launch_bwd_loop<16, 1, false, 2, 1, false, DUMP>(a, p, nb, lds, tri, stream);
  }
  if (tg) return launch_bwd_loop<32, 1, true, 1, 1, false, DUMP>(a, p, nb, lds, tri, stream);
  return launch_bwd_loop<32, 1, false, 2, 1, false, DUMP>(a, p, nb, lds, tri, stream);
}
// deep (up to 4 / 4 / 4 layers), wide-colour and two-block (hidden 64 / 64 grid channels) decoders at one wave per SIMD:
// lp_renderer_loop.hip (+ lp_renderer_loop_dump.hip); NB = blocks of 32 features
template <bool DUMP>
static int loop_bwd_table_deep(const LpRendererArgs& a, const LoopParams& p, int NB, unsigned nb, size_t lds, bool tri, hipStream_t stream) {
  const bool tg = a.color_grid.n_grids > 0, wc = a.color_chn > 4;
#define LP_LOOP_BWD(CV, NBV, TGV, MTV, MHV, WCV) return launch_bwd_loop<CV, NBV, TGV, MTV, MHV, WCV, DUMP>(a, p, nb, lds, tri, stream)
  if (a.grid.channels == 64) {
    if (p.n_t <= 1) LP_LOOP_BWD(64, 2, false, 1, 1, false);
    else LP_LOOP_BWD(64, 2, false, 2, 1, false);
  } else if (a.grid.channels == 16) {
    if (NB == 2 && tg) LP_LOOP_BWD(16, 2, true, 1, 1, false);  // two-grid decoder x 64: heads of at most two layers, no trunk
    else if (NB == 2 && p.n_t <= 1) LP_LOOP_BWD(16, 2, false, 1, 1, false);
    else if (NB == 2) LP_LOOP_BWD(16, 2, false, 2, 1, false);
    else if (tg && wc) LP_LOOP_BWD(16, 1, true, 1, LOOP_MAX_H, true);
    else if (tg) LP_LOOP_BWD(16, 1, true, 1, LOOP_MAX_H, false);
    else if (wc) LP_LOOP_BWD(16, 1, false, LOOP_MAX_T, LOOP_MAX_H, true);
    else LP_LOOP_BWD(16, 1, false, LOOP_MAX_T, LOOP_MAX_H, false);
  } else {
    // (one trunk layer -- the reference example's 1/1/2 x 64 --: an instantiation of its own keeps 32 activation + 18 dW registers
    // fewer and fits the 512-register budget without scratch; the two-trunk-layer one spills 44-46)
    if (NB == 2 && tg) LP_LOOP_BWD(32, 2, true, 1, 1, false);
    else if (NB == 2 && p.n_t <= 1) LP_LOOP_BWD(32, 2, false, 1, 1, false);
    else if (NB == 2) LP_LOOP_BWD(32, 2, false, 2, 1, false);
    else if (tg && wc) LP_LOOP_BWD(32, 1, true, 1, LOOP_MAX_H, true);
    else if (tg) LP_LOOP_BWD(32, 1, true, 1, LOOP_MAX_H, false);
    else if (wc) LP_LOOP_BWD(32, 1, false, LOOP_MAX_T, LOOP_MAX_H, true);
    else LP_LOOP_BWD(32, 1, false, LOOP_MAX_T, LOOP_MAX_H, false);
  }
#undef LP_LOOP_BWD
}

int renderer_backward_loop_shallow(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream);
// DUMP twins (with -DLP_TEST_HOOKS; LP_EUNSUPPORTED without)
int renderer_backward_loop_shallow_dump(const LpRendererArgs& a, const LoopParams& p, unsigned nb, size_t lds, bool tri, hipStream_t stream);
int renderer_backward_loop_deep_dump(const LpRendererArgs& a, const LoopParams& p, int NB, unsigned nb, size_t lds, bool tri, hipStream_t stream);

}  // namespace lp
