// lp_renderer_mfma_bwd.h -- Renderer backward of the tuned default shape on the CDNA4 matrix cores: the kernel template and its
// launcher.  Instantiated by lp_renderer_mfma_bwd.hip (16 grid channels + the host dispatch), lp_renderer_mfma_bwd_c32.hip (32
// channels), lp_renderer_mfma_bwd_aux.hip (the LP_ARITH_FP32 instantiations) and lp_renderer_mfma_bwd_dump.hip (with
// -DLP_TEST_HOOKS: the DUMP twins): four translation units instead of one -- the template is the longest compile of the library.
//
// Same maths and the same lane <-> (ray, feature) mapping as the forward kernel (lp_renderer_mfma.hip): far -> near
// sweep that recomputes the decoder of every sample (bf16x3, lp_bf3.h), then back-propagates through it.
// WEIGHT gradients.  dW = X^T dY contracts over rays; a wave that keeps the four 32x32 tiles of its own 32 rays needs 64
// accumulator registers for the whole kernel, which (with the activations of the recompute) does not fit the
// 256-register budget of two waves per SIMD.  So the four waves of a workgroup SHARE the contraction: every wave
// publishes its X / dY tiles of the current layer in LDS (feature-major, [feature][ray]), the workgroup synchronises, and
// wave w accumulates ONE 16x16 quadrant of the layer's dW over all 128 rays with v_mfma_f32_16x16x4_f32 (4 accumulator
// registers per layer, 16 in total).  Operands arrive as ds_read_b128 (8 rays per lane), the bias gradient is a
// by-product of the B operand, nothing is summed across waves at the end.
// LDS tiles are feature-major with a row stride of 36 floats: the transposing writes are 32 consecutive lanes per row
// (conflict-free), the quadrant reads are conflict-free once the 16 features of a quadrant are dealt to the MFMA lanes as
// even | odd | even (pi16() below).
// (The fp32-MFMA generation of this kernel -- renderer_bwd_mfma2, rounds 1-3, with FLEX / two-grid instantiations -- was
// retired in round 4: see lp_renderer_mfma.hip.)
#pragma once
#include "lp_mfma_common.h"
#include <type_traits>

#include "lp_bf3.h"

namespace lp {

#ifdef LP_ASM_MARKS
#define LP_MARK(n) asm volatile("; LPMARK " n)
#elif defined(LP_PHASE_TIMING)
// developer build (-DLP_PHASE_TIMING): per-phase shader-clock totals of the sample loop, summed over
// all waves into g_phase[] (read with lp_debug_phase_cycles)
static __device__ unsigned long long g_phase[16];  // (one copy per translation unit: lp_debug_phase_cycles reads the C = 16 unit's)
__device__ constexpr int phase_id(const char* n) {
  return n[0] == 'f' && n[1] == 'w' ? 0 : n[0] == 'c' && n[1] == 'o' ? 1 : n[0] == 'h' ? 2 : n[0] == 'c' ? 3
       : n[0] == 'o' ? 4 : n[0] == 't' && n[1] == '2' ? 5 : n[0] == 't' ? 6 : n[0] == 'f' ? 7 : n[0] == 's' ? 8 : 9;
}
#define LP_MARK(n)                                                     \
  {                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                 \
    const unsigned long long t_now = __builtin_readcyclecounter();     \
    ph[ph_cur] += t_now - t_last;                                      \
    t_last = t_now;                                                    \
    ph_cur = phase_id(n);                                              \
    __builtin_amdgcn_sched_barrier(0);                                 \
  }
#else
#define LP_MARK(n)
#endif
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define LP_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int T_LD = 36;  // row stride of the feature-major tiles [32 features][32 rays + 4]

// feature of a 16-wide quadrant handled by MFMA lane index m (bank-conflict-free b128 reads)
LP_DEV constexpr int pi16(int m) { return m < 4 ? 2 * m : (m < 12 ? 2 * (m - 4) + 1 : 2 * (m - 8)); }

// workgroup barrier that only drains LDS traffic (a __syncthreads() would also wait for the
// outstanding global atomics of the gradient scatter)
// (Round 3 timed the kernel with these barriers compiled out -- wrong weight gradients, 1.3 % faster; the
// switch is gone: a build that knowingly breaks correctness must not exist.)
LP_DEV void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// 16-register activation (accumulator order) -> feature-major tile
LP_DEV void tile_store_fm(float* tile, int r, int h, const float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[featq(q, h) * T_LD + r] = v[q];
}

// dW quadrant of one layer over the rays of the source waves [v0, v1): acc += X^T dY.
// a_off / b_off: float offsets of this lane's rows inside a wave area (tile + feature*T_LD + 8*(lane>>4)).
template <int STRIDE>
LP_DEV f32x4 dw_quadrant(const float* wave0, int a_off, int b_off, int v0, int v1, f32x4 acc, float& db) {
  float s = 0.0f;
  for (int v = v0; v < v1; ++v) {
    const float* base = wave0 + v * STRIDE;
    const float4 a0 = *reinterpret_cast<const float4*>(base + a_off);
    const float4 a1 = *reinterpret_cast<const float4*>(base + a_off + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(base + b_off);
    const float4 b1 = *reinterpret_cast<const float4*>(base + b_off + 4);
    acc = LP_MFMA16(a0.x, b0.x, acc);
    acc = LP_MFMA16(a0.y, b0.y, acc);
    acc = LP_MFMA16(a0.z, b0.z, acc);
    acc = LP_MFMA16(a0.w, b0.w, acc);
    acc = LP_MFMA16(a1.x, b1.x, acc);
    acc = LP_MFMA16(a1.y, b1.y, acc);
    acc = LP_MFMA16(a1.z, b1.z, acc);
    acc = LP_MFMA16(a1.w, b1.w, acc);
    s += ((b0.x + b0.y) + (b0.z + b0.w)) + ((b1.x + b1.y) + (b1.z + b1.w));
  }
  db += s;
  return acc;
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients on the bf16 pipe (round 5; default with LP_DX_LIMBS == 2, -DLP_DW_FP32 keeps the fp32 quadrants above).
// The fp32 16x16x4 products are 32 cycles each that no VALU instruction overlaps (112 per wave-sample = 22 % of the SIMD's
// time).  Here the PRODUCER lanes publish two-limb bf16 tiles [ray][feature] -- the dY limbs are the ones the dX chain forms
// anyway (layer_bf2v's `trow`), the X limbs cost one two-limb split per activation (limb_tile_store, 24 VALU per chunk) -- and
// wave w accumulates its 16 x 16 quadrant with v_mfma_f32_16x16x32_bf16: K = the 32 rays of one source wave, three limb
// products x1 y1, x2 y1, x1 y2 (dropped terms ~2^-16 |x y|, random in sign, summed over 10^6..10^7 ray-samples per entry).
// Operand layout: lane (m = l & 15, kq = l >> 4) supplies the rays 8 kq .. 8 kq + 7 of feature f0 + m: two
// ds_read_b64_tr_b16 of four consecutive tile rows each; A and B use the same ray order by construction.  Tile rows: ray k sits
// in row rho(k) (bits 1 and 3 of k swapped) of the rm_off layout: the ds_write_b128 of the producers and the transposed reads of
// the consumers are both bank-conflict free (scripts/lds_bank_model.py).  Result: lane holds dW[16 mi + 4 kq + i][16 ni + m].
// The bias gradient (column sums of dY) used to be a by-product of the fp32 B operand.  Now: one more product per limb with a
// ONE-HOT A operand (row `li` all ones, li = layer index 0..3): D[li][n] += sum_k dY[k][n], every other row += 0 -- the four
// layers share ONE f32x4 accumulator (lanes 0..15 hold rows 0..3 = the four layers' bias gradients of column 16 ni + m).
// Measured: profiles/r05_dw_bf16_ab.txt.
typedef __bf16 bf16x8_dw __attribute__((ext_vector_type(8)));
#define LP_MFMA16B(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_dw, (a)), __builtin_bit_cast(bf16x8_dw, (b)), (c), 0, 0, 0)
constexpr int LT_LIMB = rm_bytes(32);  // bytes of one limb tile [32 rays][32 features]

// eight rays (tile rows 8 kq .. 8 kq + 7) of one feature column: p = the supplier address of the first four rows
LP_DEV u32x4_t limb_tile_operand(const char* p) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + (rm_off(4, 0) - rm_off(0, 0))));  // rows + 4
  const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
  return (u32x4_t){ua.x, ua.y, ub.x, ub.y};
}
// x_off / y_off: byte offsets of this lane's supplier address inside a wave area (X / dY limb tile, limb 1); onehot: the A operand
// of the bias product
template <int STRIDE_BYTES>
LP_DEV f32x4 dw_quadrant_bf(const char* wave0b, int x_off, int y_off, int v0, int v1, f32x4 acc, f32x4& acc_db, unsigned onehot, bool do_db = true) {
  const u32x4_t oh = {onehot, onehot, onehot, onehot};
#pragma unroll 1
  for (int v = v0; v < v1; ++v) {
    const char* base = wave0b + v * STRIDE_BYTES;
    const u32x4_t b2 = limb_tile_operand(base + y_off + LT_LIMB);
    const u32x4_t a1 = limb_tile_operand(base + x_off);
    if (do_db) acc_db = LP_MFMA16B(oh, b2, acc_db);  // (wave-uniform)
    acc = LP_MFMA16B(a1, b2, acc);
    const u32x4_t b1 = limb_tile_operand(base + y_off);
    if (do_db) acc_db = LP_MFMA16B(oh, b1, acc_db);
    acc = LP_MFMA16B(a1, b1, acc);
    const u32x4_t a2 = limb_tile_operand(base + x_off + LT_LIMB);
    acc = LP_MFMA16B(a2, b1, acc);
  }
  return acc;
}
#if LP_DX_LIMBS == 2 && !defined(LP_DW_FP32)
#define LP_DW_BF16 1
#else
#define LP_DW_BF16 0
#endif

// =======================================================================================
// Backward of the default shape with the recompute and the dX chains as bf16x3 on the bf16 matrix cores (lp_bf3.h).
//
// Same sweep, same lane mapping, same workgroup-shared dW scheme as renderer_bwd_mfma2 above; what changes:
// * 120 of the 232 fp32 MFMA-equivalents per 32 ray-samples (recompute 56 + dX 64, 7.7 k matrix-pipe cycles that no
//   VALU instruction can overlap with) become 90 v_mfma_f32_32x32x16_bf16 (2.9 k cycles that DO overlap with the VALU
//   work of the SIMD's other wave), at the price of ~570 VALU instructions for the operand splits.  The dW quadrants
//   stay fp32 16x16x4 (their operands cross waves through LDS).
// * The weights live in LDS as limb images in BOTH orientations (45 KB) next to the per-wave tiles, so a workgroup is
//   EIGHT waves (one workgroup per CU, still two waves per SIMD); the dW quadrant of a layer is accumulated by two
//   groups of four waves, each over the rays of four source waves.
// * The ray encoding enters the colour hidden layer as a per-ray pre-activation (cb = b_c1 + W_c1^T enc: registers
//   instead of an LDS tile), so the colour layer reuses the limbs of e, and by linearity
//       d enc = W_c1 D ,   dW_c1 = sum e (x) d hc + enc (x) D ,   D = sum over samples of d hc
//   D is accumulated per sample (16 adds, what d enc cost before); the two products are formed ONCE after the sweep.
// =======================================================================================
struct LdsB3 {  // per-wave area behind the images (floats)
  static constexpr int XT = 0;
  static constexpr int YT = 32 * T_LD;
  static constexpr int TS = 2 * 32 * T_LD;
  static constexpr int PER_WAVE = TS + 5 * 32;
};
// NW = 8: eight-wave workgroups, slot images in both orientations (one workgroup per CU).
// NW = 4: four-wave workgroups, ONE row-major limb image per layer read plainly by the dX chains and through
//         ds_read_b64_tr_b16 by the recompute (lp_bf3.h): 81 KB per workgroup, so TWO independent workgroups share a CU
//         and a SIMD hosts waves in unrelated phases again -- the eight waves of NW = 8 are barrier-locked into the same
//         phase, so its two waves per SIMD fight for the same pipe instead of overlapping.
template <int C, int NW>
struct Bf3Lds {
  static constexpr int IMG_END = (NW == 8) ? LdsBf3<C>::BWD_END : LdsBf3Rm<C>::END;  // bytes before the per-wave tiles
  static constexpr int CB = IMG_END + NW * LdsB3::PER_WAVE * 4;                       // lane-private cb records
  // the cb records live in LDS where they fit (C = 16); with C = 32 the image is 3.4 KB larger and two workgroups per CU
  // only fit if cb stays in registers
  static constexpr bool CB_LDS = (NW == 8) ? (C == 16) : (CB + NW * 64 * 16 * 4) * 2 <= 160 * 1024;
  static constexpr int TOTAL = CB + (CB_LDS ? NW * 64 * 16 * 4 : 0);
};

// DUMP (test hook, its own instantiations: the production kernels are unchanged): the ReLU decisions this backward takes are
// written to mp.relu_dump -- the same instruction sequence computes them, only the stores are added.
// F32 (LpRendererArgs.arithmetic == LP_ARITH_FP32): three limbs for the gradient operand of the dX chains and fp32 weight-gradient
// quadrants (v_mfma_f32_16x16x4_f32 on fp32 tiles) -- the reference's arithmetic, selectable per call; the default instantiations
// take two limbs in both (LP_DX_LIMBS, LP_DW_BF16).
template <int C, int GM, bool PLAIN, int NC, int NW, bool SEG = false, bool DUMP = false, bool F32 = false>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) renderer_bwd_bf3(const LpRendererArgs a, const MfmaParams mp) {
  constexpr int DXL = F32 ? 3 : LP_DX_LIMBS;          // limbs of the gradient operand of the dX chains
  constexpr bool DWB = !F32 && (LP_DW_BF16 != 0);     // weight gradients on the bf16 pipe (two-limb tiles)
  using M = Lds;
  using L = LdsBf3<C>;
  using R = LdsBf3Rm<C>;
  using B = LdsB3;
  constexpr int WAVES3 = NW;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  if (NW == 8) stage_weights_bf3<C>(a, mp, lds, true, 64 * NW);
  else stage_weights_rm<C>(a, mp, lds, 64 * NW);
  const float* const sm = lds - M::BIAS;  // small fp32 block: sm[Lds::X]
  const char* const fimg = reinterpret_cast<const char*>(lds) + L::FWD_IMG;   // NW = 8 only
  const char* const bimg = reinterpret_cast<const char*>(lds) + L::BWD_IMG;   // NW = 8 only
  const char* const rimg = reinterpret_cast<const char*>(lds);                // NW = 4: layer offsets R::L_* are absolute
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // scalar: every per-wave base below is an SGPR
  const int h = lane >> 5, r = lane & 31;
  float* const wave0 = reinterpret_cast<float*>(reinterpret_cast<char*>(lds) + Bf3Lds<C, NW>::IMG_END);
  float* const wv = wave0 + wave * B::PER_WAVE;
  float* const xt = wv + B::XT;
  float* const yt = wv + B::YT;
  float* const ts = wv + B::TS;

  // segment-parallel sweep (LpRendererArgs.seg_prefix): workgroup = (128 rays, one block of LP_SEG_LEN samples)
  // (SEG is its own instantiation: the full-batch kernel keeps its register allocation)
  constexpr bool seg_on = SEG;
  const int n_rec = seg_on ? segment_count(a.march) : 1;                        // saved states per ray
  const int seg_len = LP_SEG_LEN * mp.seg_blocks;                               // samples per workgroup
  const int n_seg = seg_on ? (a.march.num_samples + seg_len - 1) / seg_len : 1;  // workgroups per 128 rays
  const int blk = seg_on ? (int)blockIdx.x / n_seg : (int)blockIdx.x;
  const int seg = seg_on ? (int)blockIdx.x - blk * n_seg : 0;
  const int64_t ray_id = ((int64_t)blk * WAVES3 + wave) * RAYS_PER_WAVE + r;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  const Ray ray = load_ray(a.rays, rid);
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
  if (PLAIN) {
    __syncthreads();
  } else {
    if (lane == 0) ts[0] = (float)s_last_w;
    __syncthreads();
    s_begin = 0;
#pragma unroll
    for (int v = 0; v < WAVES3; ++v) {
      const int sv = (int)wave0[v * B::PER_WAVE + B::TS];
      s_begin = sv > s_begin ? sv : s_begin;
    }
    __syncthreads();  // ts[] is reused by the sample loop
  }
  const int s_lo = seg_on ? seg * seg_len : 0;
  if (seg_on) s_begin = (s_lo + seg_len - 1 < s_tot - 1) ? s_lo + seg_len - 1 : s_tot - 1;
  // per-ray pre-activation of the colour hidden layer, cb = b_c1 + W_c1^T enc: read once per sample, so it lives in LDS
  // (lane-private 64-byte records behind the tiles, the four 16-byte quarters rotated by lane >> 2: conflict-free)
  constexpr bool CBL = Bf3Lds<C, NW>::CB_LDS;
  float* const cbt = wave0 + WAVES3 * B::PER_WAVE + (wave * 64 + lane) * 16;
  const int cb_rot = (lane >> 2) & 3;
  float cb[16];  // (dead after this block when the records live in LDS)
  {
    float enc[16];
    load_encoding(a, rid, h, enc);
    if constexpr (NW == 8) color_prebias_bf3(sm, ASlots{fimg, L::CH_C1}, lane, enc, cb);
    else color_prebias_bf3(sm, AColsFwd{rimg + R::L_C1, R::ST_32}, lane, enc, cb);
    if (CBL) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        *reinterpret_cast<float4*>(cbt + ((i + cb_rot) & 3) * 4) = make_float4(cb[4 * i], cb[4 * i + 1], cb[4 * i + 2], cb[4 * i + 3]);
    }
  }
  float dsum[16];  // D = sum over samples of d hc
#pragma unroll
  for (int q = 0; q < 16; ++q) dsum[q] = 0.0f;
  float gfeat[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
    gfeat[c] = (valid && a.grad_feature && c < a.color_chn) ? a.grad_feature[rid * a.color_chn + c] : 0.0f;
  const float g_len = (valid && a.grad_ray_length) ? a.grad_ray_length[rid] : 0.0f;
  const float g_nlt = epilogue_grad_nlt(a, rid, valid, a.neg_log_t[rid],
                                        (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f, gfeat, 4);
  const bool want_params = a.grad_mlp_params != nullptr;
  const float delta0 = (a.march.num_samples > 1) ? (ray.far_t - ray.near_t) / (float)(a.march.num_samples - 1) : 1.0f;

  // dW quadrant of this wave: quadrant (mi, ni) = wave & 3 over the rays of the source waves [v0, v0 + 4)
  // (NW = 8: two groups of four waves, each over four source waves; NW = 4: every wave over all four)
  const int mi = (wave & 3) >> 1, ni = wave & 1;
  const int v0 = 4 * (wave >> 2);
  const int m16 = lane & 15, ka = lane >> 4;
  const int a_off = B::XT + (16 * mi + pi16(m16)) * T_LD + 8 * ka;
  const int b_off = B::YT + (16 * ni + pi16(m16)) * T_LD + 8 * ka;
  // bf16 dW (LP_DW_BF16): the limb tiles [ray][feature] alias the X / dY tile areas; this lane publishes row rho(r) of its wave's
  // tiles, and as MFMA lane (m16, ka) it supplies rows rho(8 ka + (m16 >> 2)) [+ 4], columns f0 + 4 (m16 & 3) .. +3 of a source wave's
  // (rho: see dw_quadrant_bf -- with rows in ray order every limb-tile write was a 2-way bank conflict, 128 LDS cycles per sample)
  auto rho = [](int k) { return (k & 0x15) | ((k & 2) << 2) | ((k & 8) >> 2); };
  char* const xrow = reinterpret_cast<char*>(wv) + B::XT * 4 + rm_off(rho(r), 4 * h);
  char* const yrow = xrow + (B::YT - B::XT) * 4;   // (a constant apart: folds into the store's offset field)
  const int xq_off = B::XT * 4 + rm_off(rho(8 * ka + (m16 >> 2)), 4 * (m16 & 3)) + 32 * mi;   // (16 mi columns = 32 mi bytes)
  const int yq_off = B::YT * 4 + rm_off(rho(8 * ka + (m16 >> 2)), 4 * (m16 & 3)) + 32 * ni;
  f32x4 dq_b = {0, 0, 0, 0};  // bias gradients of the four hidden layers (rows 0 c1, 1 o1, 2 t2, 3 t1; lanes 0..15)
  auto onehot = [&](int li) -> unsigned { return (lane & 15) == li ? 0x3F803F80u : 0u; };  // bf16 (1, 1) in the row of layer li
  // trunk layer 1 has only C input rows: with C == 16 its two quadrants (ni) are split over four ray groups instead
  const int a_off_t1 = (C == 16) ? B::XT + pi16(m16) * T_LD + 8 * ka : a_off;
  // (NW = 8: four ray groups of two source waves; NW = 4: two groups -- wave >> 1 -- of two source waves)
  const int t1_v0 = (C == 16) ? 2 * (wave >> 1) : v0, t1_v1 = (C == 16) ? 2 * (wave >> 1) + 2 : v0 + 4;
  f32x4 dq_t1 = {0, 0, 0, 0}, dq_t2 = {0, 0, 0, 0}, dq_o1 = {0, 0, 0, 0}, dq_c1 = {0, 0, 0, 0};
  float db_t1 = 0.0f, db_t2 = 0.0f, db_o1 = 0.0f, db_c1 = 0.0f;
  float dwo2 = 0.0f, dwc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  float dbo2 = 0.0f, dbc2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool gg = a.grad_grid_list[0] != nullptr;

#ifdef LP_PHASE_TIMING
  unsigned long long ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long t_last = __builtin_readcyclecounter();
  int ph_cur = 9;
#endif
  float nlt = a.neg_log_t[rid];
  // d loss / d (opacity delta)_s = T_s p_s - sum_{i > s} w_i p_i, p_i = g_len depth_i + sum_c g_c colour_ic: the running
  // `suffix` carries the second term; a segment starts it from the sums the forward saved behind its last sample
  float suffix = 0.0f, p_next = 0.0f;
  if (seg_on) {
    const float4* pj = reinterpret_cast<const float4*>(a.seg_prefix + (rid * n_rec + s_begin / LP_SEG_LEN) * 8);
    const float4* pt = reinterpret_cast<const float4*>(a.seg_prefix + (rid * n_rec + n_rec - 1) * 8);
    const float4 j0 = pj[0], j1 = pj[1], t0 = pt[0];
    nlt = j1.y;     // -log T behind the segment's last sample (a segment may end between two checkpoints)
    nlt_lo = j1.z;
    if (seg < n_seg - 1) {
      float rest = g_len * (t0.x - j0.x);
      rest = fmaf(gfeat[0], t0.y - j0.y, rest);
      rest = fmaf(gfeat[1], t0.z - j0.z, rest);
      rest = fmaf(gfeat[2], t0.w - j0.w, rest);
      if (NC == 4) rest = fmaf(gfeat[3], pt[1].x - j1.x, rest);
      suffix = -rest;
    }
  }
  Sample<C> nx;
  fetch_sample<C, GM, false, PLAIN>(a, sm, ray, s_begin, h, nx);
  for (int s = s_begin; s >= s_lo; --s) {
    const bool on = PLAIN || s <= s_last_w;
    const float depth = nx.depth, occ = nx.occ, x = nx.x, y = nx.y, z = nx.z;
    float x0[C / 2];
#pragma unroll
    for (int q = 0; q < C / 2; ++q) x0[q] = nx.x0[q];
    const int zo = opaque_zero();
    const float* ldz = sm + zo;
    // A-operand loaders of the four layers (0 t1, 1 t2, 2 o1, 3 c1), forward (recompute) and backward (dX) form
    auto Af = [&](auto layer) {
      constexpr int l = decltype(layer)::value;
      if constexpr (NW == 8) return ASlots{fimg + zo, l == 0 ? L::CH_T1 : l == 1 ? L::CH_T2 : l == 2 ? L::CH_O1 : L::CH_C1};
      else return AColsFwd{rimg + zo + (l == 0 ? R::L_T1 : l == 1 ? R::L_T2 : l == 2 ? R::L_O1 : R::L_C1), l == 0 ? R::ST_T1 : R::ST_32};
    };
    auto Ab = [&](auto layer) {
      constexpr int l = decltype(layer)::value;
      if constexpr (NW == 8) return ASlots{bimg + zo, l == 0 ? L::CB_T1 : l == 1 ? L::CB_T2 : l == 2 ? L::CB_O1 : L::CB_C1};
      else return ARowsBwd{rimg + zo + (l == 0 ? R::L_T1 : l == 1 ? R::L_T2 : l == 2 ? R::L_O1 : R::L_C1), l == 0 ? R::ST_T1 : R::ST_32,
                           l == 0 ? C - 1 : 31};
    };
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

    // ---------------- forward recompute (bf16x3) ----------------
    LP_MARK("fwd");
    float h1[16];
    float e[16];
    unsigned ho_mask = 0, hc_mask = 0;
    Heads hd;
    {
      f32x16 acc;
      {
        acc = layer_bf3v<C / 16>(Af(I0{}), lane, x0, load_bias(sm, 0, h, zo));
#pragma unroll
        for (int q = 0; q < 16; ++q) h1[q] = relu_f(acc[q]);
      }
      acc = layer_bf3v<2>(Af(I1{}), lane, h1, load_bias(sm, 1, h, zo));
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        e[q] = relu_f(acc[q]);
      }
      float ho[16], hc[16];
      {
        f32x16 acc_o = load_bias(sm, 2, h, zo), acc_c;
        if (CBL) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float4 v = *reinterpret_cast<const float4*>(cbt + zo + ((i + cb_rot) & 3) * 4);
            acc_c[4 * i] = v.x; acc_c[4 * i + 1] = v.y; acc_c[4 * i + 2] = v.z; acc_c[4 * i + 3] = v.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < 16; ++q) acc_c[q] = cb[q];
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
        if (valid && h == 0) {
          uint32_t* d = mp.relu_dump + (rid * s_tot + s) * 5;
          d[0] = m1; d[1] = m2; d[2] = mo; d[3] = mc; d[4] = on ? 1u : 2u;
        }
      }
      LP_SCHED_FENCE();
      // ho / hc go to the (wave-private) tiles: the output layers' dW reads them from there
      if (want_params) {
        tile_store_fm(xt, r, h, ho);
        tile_store_fm(yt, r, h, hc);
      }
      LP_SCHED_FENCE();
    }

    // ---------------- compositing, backward ----------------
    LP_MARK("compositing");
    const float depth_prev =
        PLAIN ? ray.near_t + lin01((s > 0) ? s - 1 : 0, a.march.num_samples) * (ray.far_t - ray.near_t)
              : sample_depth_tab((s > 0) ? s - 1 : 0, a.march, ray.near_t, ray.far_t, sm + M::INF);
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
    LP_MARK("heads_bwd");
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

    // ---------------- colour hidden layer (X tile = e: shared with the opacity layer below) ----------------
    LP_MARK("c1");
#ifndef LP_NO_PRIO
    __builtin_amdgcn_s_setprio(1);
#endif
    f32x16 acc = (f32x16){0};
    {
      if constexpr (DWB) {
        if (want_params) limb_tile_store<2>(xrow, e);
        acc = layer_dxv<2, DXL>(Ab(I3{}), lane, dhc, acc, want_params ? yrow : nullptr);
      } else {
        if (want_params) {
          tile_store_fm(xt, r, h, e);
          tile_store_fm(yt, r, h, dhc);
        }
        acc = layer_dxv<2, DXL>(Ab(I3{}), lane, dhc, acc);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) dsum[q] += dhc[q];
      if (want_params) {
        lds_barrier();
        if constexpr (DWB) dq_c1 = dw_quadrant_bf<B::PER_WAVE * 4>(reinterpret_cast<const char*>(wave0), xq_off, yq_off, v0, v0 + 4, dq_c1, dq_b, onehot(0));
        else dq_c1 = dw_quadrant<B::PER_WAVE>(wave0, a_off, b_off, v0, v0 + 4, dq_c1, db_c1);
        lds_barrier();
      }
    }
    LP_SCHED_FENCE();
    // ---------------- opacity hidden layer ----------------
    LP_MARK("o1");
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
      if constexpr (DWB) {
        acc = layer_dxv<2, DXL>(Ab(I2{}), lane, dho, acc, want_params ? yrow : nullptr);  // the X tile still holds e
      } else {
        if (want_params) tile_store_fm(yt, r, h, dho);  // the X tile still holds e
        acc = layer_dxv<2, DXL>(Ab(I2{}), lane, dho, acc);
      }
      if (want_params) {
        lds_barrier();
        if constexpr (DWB) dq_o1 = dw_quadrant_bf<B::PER_WAVE * 4>(reinterpret_cast<const char*>(wave0), xq_off, yq_off, v0, v0 + 4, dq_o1, dq_b, onehot(1));
        else dq_o1 = dw_quadrant<B::PER_WAVE>(wave0, a_off, b_off, v0, v0 + 4, dq_o1, db_o1);
        lds_barrier();
      }
    }
    float de[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) de[q] = (e[q] > 0.0f) ? acc[q] : 0.0f;
    LP_SCHED_FENCE();
    // ---------------- trunk layer 2 ----------------
    LP_MARK("t2");
    float dh1[16];
    {
      if constexpr (DWB) {
        if (want_params) limb_tile_store<2>(xrow, h1);
        acc = layer_dxv<2, DXL>(Ab(I1{}), lane, de, (f32x16){0}, want_params ? yrow : nullptr);
      } else {
        if (want_params) {
          tile_store_fm(xt, r, h, h1);
          tile_store_fm(yt, r, h, de);
        }
        acc = layer_dxv<2, DXL>(Ab(I1{}), lane, de, (f32x16){0});
      }
      if (want_params) {
        lds_barrier();
        if constexpr (DWB) dq_t2 = dw_quadrant_bf<B::PER_WAVE * 4>(reinterpret_cast<const char*>(wave0), xq_off, yq_off, v0, v0 + 4, dq_t2, dq_b, onehot(2));
        else dq_t2 = dw_quadrant<B::PER_WAVE>(wave0, a_off, b_off, v0, v0 + 4, dq_t2, db_t2);
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) dh1[q] = (h1[q] > 0.0f) ? acc[q] : 0.0f;
      if (want_params) lds_barrier();
    }
    LP_SCHED_FENCE();
    // ---------------- trunk layer 1 ----------------
    LP_MARK("t1");
    {
      if constexpr (DWB) {
        if (want_params) limb_tile_store<C / 16>(xrow, x0);
        if (gg) {
          acc = layer_dxv<2, DXL>(Ab(I0{}), lane, dh1, (f32x16){0}, want_params ? yrow : nullptr);  // rows >= C of the result are unused
        } else if (want_params) {
          limb_tile_store<2>(yrow, dh1);
        }
      } else {
        if (want_params) {
#pragma unroll
          for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * T_LD + r] = x0[q];
          tile_store_fm(yt, r, h, dh1);
        }
        if (gg) {
          acc = layer_dxv<2, DXL>(Ab(I0{}), lane, dh1, (f32x16){0});  // rows >= C of the result are unused
        }
      }
      if (want_params) {
        lds_barrier();
        if constexpr (DWB) dq_t1 = dw_quadrant_bf<B::PER_WAVE * 4>(reinterpret_cast<const char*>(wave0), (C == 16) ? xq_off - 32 * mi : xq_off, yq_off, t1_v0, t1_v1, dq_t1, dq_b, onehot(3));
        else dq_t1 = dw_quadrant<B::PER_WAVE>(wave0, a_off_t1, b_off, t1_v0, t1_v1, dq_t1, db_t1);
        lds_barrier();
      }
    }
    if (gg) {
#pragma unroll
      for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * DX_LD + r] = acc[q];
    }
    LP_SCHED_FENCE();
    // ---------------- next (nearer) sample + grid gradient ----------------
    LP_MARK("fetch");
    __builtin_amdgcn_s_setprio(0);
    const bool live = valid && on && !(a.march.mask_out_of_bounds && !point_in_bounds(x, y, z));
    if (s > s_lo) fetch_sample<C, GM, true, PLAIN>(a, sm, ray, s - 1, h, nx);
    LP_SCHED_FENCE();
    LP_MARK("scatter");
    if (gg && !(mp.dbg & 2)) {
#ifndef LP_SCATTER_V1
      if constexpr (GM == GM_TRIPLANE) {
        scatter_triplane<C>(a.grad_grid_list, a.grid, ray.b, x, y, z, live, lane, xt, yt, mp.dbg);
      } else
#endif
      {
        const int ng = (GM == GM_TRIPLANE) ? 3 : (GM == GM_VOXEL) ? 1 : a.grid.n_grids;
#pragma unroll 1
        for (int g = 0; g < ng; ++g)
          scatter_grid<C, GM>(a.grad_grid_list[g], a.grid.grids[g], ray.b, x, y, z, live, lane, xt, yt, mp.dbg);
      }
    }
  }

  // ---------------- epilogue ----------------
  LP_MARK("epilogue");
#ifdef LP_PHASE_TIMING
  if (lane == 0) {
    for (int i = 0; i < 10; ++i) atomicAdd(&g_phase[i], ph[i]);
  }
#endif
  // d enc = W_c1 D ; dW_c1 += enc (x) D   (one product each, after the sweep)
  {
    f32x16 acc;
    if constexpr (NW == 8) acc = layer_bf3v<2>(ASlots{bimg, L::CB_C1}, lane, dsum, (f32x16){0});
    else acc = layer_bf3v<2>(ARowsBwd{rimg + R::L_C1, R::ST_32, 31}, lane, dsum, (f32x16){0});
    if (valid && a.grad_encoding && !seg_on) {
      float4* dst = reinterpret_cast<float4*>(a.grad_encoding + ray_id * HID + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[2 * j] = make_float4(acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]);
    } else if (valid && a.grad_encoding) {  // the segments of a ray add up
      float* dst = a.grad_encoding + ray_id * HID + 4 * h;
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) atomic_add_f32(dst + 8 * j + i, acc[4 * j + i]);
    }
  }
  if (want_params) {
    {
      float enc[16];
      load_encoding(a, rid, h, enc);
      if (!valid) {
#pragma unroll
        for (int q = 0; q < 16; ++q) enc[q] = 0.0f;
      }
      __syncthreads();  // every wave is done with its tiles (the scatter of the last sample reads them)
      if constexpr (DWB) {
        limb_tile_store<2>(xrow, enc);
        limb_tile_store<2>(yrow, dsum);
        lds_barrier();
        dq_c1 = dw_quadrant_bf<B::PER_WAVE * 4>(reinterpret_cast<const char*>(wave0), xq_off, yq_off, v0, v0 + 4, dq_c1, dq_b, 0u, false);
      } else {
        tile_store_fm(xt, r, h, enc);
        tile_store_fm(yt, r, h, dsum);
        float db_unused = 0.0f;
        lds_barrier();
        dq_c1 = dw_quadrant<B::PER_WAVE>(wave0, a_off, b_off, v0, v0 + 4, dq_c1, db_unused);  // the bias saw d hc already
      }
    }
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
    // (bf16 quadrants: 16x16x32 accumulator, column = lane & 15, rows 4 (lane >> 4) + i; fp32 quadrants: the X rows / dY columns are
    // read through pi16)
    const int col = DWB ? 16 * ni + m16 : 16 * ni + pi16(m16);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int prow = DWB ? 4 * ka + i : pi16(4 * ka + i);
      atomic_add_f32(G + mp.w_t2 + (16 * mi + prow) * HID + col, dq_t2[i]);
      atomic_add_f32(G + mp.w_o1 + (16 * mi + prow) * HID + col, dq_o1[i]);
      atomic_add_f32(G + mp.w_c1 + (16 * mi + prow) * HID + col, dq_c1[i]);
      const int row1 = (C == 16) ? prow : 16 * mi + prow;
      if (row1 < C) atomic_add_f32(G + mp.w_t1 + row1 * HID + col, dq_t1[i]);
    }
    if constexpr (DWB) {
      db_c1 = dq_b[0]; db_o1 = dq_b[1]; db_t2 = dq_b[2]; db_t1 = dq_b[3];  // lanes 0..15 (ka == 0): rows 0..3 of the one-hot products
    } else {
      db_t1 += __shfl_xor(db_t1, 16); db_t1 += __shfl_xor(db_t1, 32);
      db_t2 += __shfl_xor(db_t2, 16); db_t2 += __shfl_xor(db_t2, 32);
      db_o1 += __shfl_xor(db_o1, 16); db_o1 += __shfl_xor(db_o1, 32);
      db_c1 += __shfl_xor(db_c1, 16); db_c1 += __shfl_xor(db_c1, 32);
    }
    if (ka == 0) {
      if (mi == 0) {  // both quadrant rows of a ray group see the same dY columns: count them once per group
        atomic_add_f32(G + mp.b_t2 + col, db_t2);
        atomic_add_f32(G + mp.b_o1 + col, db_o1);
        atomic_add_f32(G + mp.b_c1 + col, db_c1);
      }
      if (C == 16 || mi == 0) atomic_add_f32(G + mp.b_t1 + col, db_t1);
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

template <int C, int GM, bool PLAIN, int NC, int NW, bool SEG = false, bool DUMP = false, bool F32 = false>
static int launch_bwd3w(const LpRendererArgs& a, const MfmaParams& mp_, hipStream_t stream) {
  MfmaParams mp = mp_;
  const unsigned ray_blocks = (unsigned)((a.rays.n_rays + NW * RAYS_PER_WAVE - 1) / (NW * RAYS_PER_WAVE));
  unsigned segs = 1;
  if (SEG) {
    // blocks per segment: seg_blocks_for() (lp_host.h); two workgroups per CU are resident
    const int n_rec = (a.march.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN;
    const int m = seg_blocks_for(ray_blocks, n_rec, 512u);
    mp.seg_blocks = m;
    segs = (unsigned)((n_rec + m - 1) / m);
  }
  constexpr size_t lds = (size_t)Bf3Lds<C, NW>::TOTAL;
  static_assert(lds * (NW == 8 ? 1 : 2) <= 160 * 1024, "the workgroups of one CU must fit the 160 KB LDS");
  static_assert(NW == 8 || 2 * lds <= 160 * 1024, "two 4-wave workgroups per CU");
  const hipError_t e = hipFuncSetAttribute((const void*)renderer_bwd_bf3<C, GM, PLAIN, NC, NW, SEG, DUMP, F32>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));
  const unsigned nb = ray_blocks * segs;
  hipLaunchKernelGGL((renderer_bwd_bf3<C, GM, PLAIN, NC, NW, SEG, DUMP, F32>), dim3(nb), dim3(64 * NW), lds, stream, a, mp);
  return LP_OK;
}
// PLAIN = the common configuration (no opacity noise, no contraction, no scaffold, no beyond-far samples, no early termination): a
// leaner instantiation of the same kernel
static inline bool bwd_is_plain(const LpRendererArgs& a) {
  return !(a.noise_sigma > 0.0f) && !a.march.contract_coords && !a.scaffold && a.march.num_samples_inf == 0 && !(a.stop_neg_log_t > 0.0f);
}

// production instantiations of one (C, GM): PLAIN x NC x {four-wave, eight-wave, segmented}
template <int C, int GM, bool PLAIN, int NC>
static int launch_bwd3(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
  // segment-parallel sweep of a small batch (seg_prefix survives lp_api.hip only where renderer_mfma_segments() > 1)
  if (a.seg_prefix) return launch_bwd3w<C, GM, PLAIN, NC, 4, true>(a, mp, stream);
  // four-wave workgroups (two per CU) unless the beyond-far table does not fit their small block; LP_BF3_NW=8 for A/B
  static const int forced = getenv("LP_BF3_NW") ? atoi(getenv("LP_BF3_NW")) : 0;
  const bool nw4 = forced ? forced == 4 : a.march.num_samples_inf <= LdsBf3Rm<C>::N_INF;
  return nw4 ? launch_bwd3w<C, GM, PLAIN, NC, 4>(a, mp, stream) : launch_bwd3w<C, GM, PLAIN, NC, 8>(a, mp, stream);
}
// recompute + dX chains as bf16x3 on the bf16 matrix cores (renderer_bwd_bf3).  Measured on MI355X against the fp32-MFMA
// kernel of rounds 1-3 (C = 16): cfg 2 backward 2.57 -> 2.31 ms, 1080p x S=128 73.6 -> 65.2 ms; C = 32 (the 3.4 KB larger image
// leaves no room for the cb records, so cb stays in registers; compiled spill-free: build.py FILE_FLAGS): cfg 4 160 -> 133 ms.
template <int C, int GM>
static int launch_bwd2(const LpRendererArgs& a, const MfmaParams& mp, hipStream_t stream) {
  const bool plain = bwd_is_plain(a);
#ifdef LP_DEV_ONE  // development aid: compile ONE instantiation (seconds instead of minutes) for register / ISA studies,
                   // e.g. scripts/kernel_resources.py lp_renderer_mfma_bwd.hip -DLP_DEV_ONE
  return launch_bwd3w<C, GM, true, 3, 4>(a, mp, stream);
#else
  static const bool no_nc3 = getenv("LP_MFMA_NO_NC3") != nullptr;  // A/B knob
  if (a.color_chn <= 3 && !no_nc3)  // RGB: the padding column of the colour path is compiled out
    return plain ? launch_bwd3<C, GM, true, 3>(a, mp, stream) : launch_bwd3<C, GM, false, 3>(a, mp, stream);
  return plain ? launch_bwd3<C, GM, true, 4>(a, mp, stream) : launch_bwd3<C, GM, false, 4>(a, mp, stream);
#endif
}
template <int C>
static int launch_bwd_gm(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream) {
#ifdef LP_DEV_ONE
  (void)gm;
  return launch_bwd2<C, GM_TRIPLANE>(a, mp, stream);
#else
  switch (gm) {
    case GM_TRIPLANE: return launch_bwd2<C, GM_TRIPLANE>(a, mp, stream);
    case GM_VOXEL: return launch_bwd2<C, GM_VOXEL>(a, mp, stream);
    default: return launch_bwd2<C, GM_GENERIC>(a, mp, stream);
  }
#endif
}

// per translation unit (see the head of this file); gm = GM_*
int renderer_bwd_bf3_c16(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream);
int renderer_bwd_bf3_c32(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream);
int renderer_bwd_bf3_f32(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream);   // LP_ARITH_FP32
int renderer_bwd_bf3_dump(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream);  // mp.relu_dump set
// transposed march (LP_MARCH_SAMPLES_PER_WAVE): lp_renderer_mfma_bwd_tm.hip
bool renderer_bwd_tm_supported(const LpRendererArgs& a);
int renderer_bwd_bf3_tm_launch(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream);
int debug_phase_cycles_c32(unsigned long long* out);  // developer builds with -DLP_PHASE_TIMING (the 32-channel unit's g_phase), else -1

}  // namespace lp
