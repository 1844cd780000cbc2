// lp_bf3.h -- the decoder's matrix products on the bf16 matrix cores at fp32 accuracy ("bf16x3").
//
// Why.  v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD) and, measured
// (scripts/mfma_valu_overlap.hip, profiles/r02_mfma_valu_overlap.txt), does not overlap with VALU work at all: 16 of
// them + 128 v_fma cost 1 793 cycles in one wave against 1 118 + 668 alone, and two waves per SIMD do not help
// (M|V 1 659).  v_mfma_f32_32x32x16_bf16 does overlap (16 + 128 v_fma: 755 cycles against 520 + 634 alone) and moves
// 8x the K per instruction in half the time.  So the fp32 MFMA caps the Renderer at (MFMA time + VALU time); the bf16
// MFMA caps it at about the VALU time alone.
//
// How, without giving up fp32 accuracy.  Every fp32 operand is split EXACTLY into three bf16 limbs
// (x = x1 + x2 + x3: 3 x 8 significand bits cover the 24 of an fp32; round-to-nearest splits, residuals formed in fp32
// are exact) and a product is the six limb products of weight >= 2^-24:
//     a w  ~=  a1 w1 + (a1 w2 + a2 w1) + (a1 w3 + a2 w2 + a3 w1)            (dropped terms: <= 3 * 2^-24 |a w|)
// accumulated in the MFMA's fp32 accumulators.  Each bf16 x bf16 product is exact in fp32, so the result carries the
// rounding of an fp32 dot product (measured against fp64 on random 32 x 32 layers: 4e-8 relative, the fp32 FMA chain of
// torch.matmul: 1.5e-7 -- scripts/bf16x3_accuracy.py).  Weights are split once per workgroup while they are staged into
// LDS; activations are split in registers (5.5 VALU instructions per value, the price of the scheme).
//
// Operand layout (v_mfma_f32_32x32x16_bf16: A = 32 x 16, B = 16 x 32, 8 bf16 = 4 VGPRs per lane and operand).
// The lane <-> (ray, feature) mapping of the fp32 kernels is kept (lp_renderer_mfma.hip): lane (h, r) owns ray r and
// the 16 features feat(q, h) of every 32-wide activation, which is also the accumulator layout of every 32 x 32 MFMA.
//   B operand of chunk c (K slots 16c .. 16c+15): lane (h, r) supplies its values q = 8c .. 8c+7, i.e. features
//     feat(8c + j, h), j = 0..7, packed in pairs (j even = low half).
//   A operand of chunk c: lane (h, m) supplies W[feat(8c + j, h)][m], j = 0..7 (forward, Y^T = W^T X^T) or
//     W[m][feat(8c + j, h)] (backward, dX^T = W dY^T).
// A and B pair the same (half, slot j) by construction, so the K order inside the instruction is irrelevant.
// LDS image: one 16-byte slot per (chunk, limb, half, lane m): offset (((chunk * 3 + limb) * 2 + half) * 32 + m) * 16;
// a wave's ds_read_b128 walks 512 contiguous bytes per half: conflict-free.
#pragma once
#include "lp_mfma_common.h"

namespace lp {

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

#define LP_MFMA_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, (a)), __builtin_bit_cast(bf16x8_t, (b)), (c), 0, 0, 0)

// (a, b) -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32): a in the low half
LP_DEV unsigned pk_bf16(float a, float b) {
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
LP_DEV float bf16_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
LP_DEV float bf16_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// one fp32 -> three bf16 limbs (host / staging side, scalar)
LP_DEV void split3_scalar(float x, unsigned short& l1, unsigned short& l2, unsigned short& l3) {
  const unsigned p1 = pk_bf16(x, 0.0f);
  const float r1 = x - bf16_lo(p1);
  const unsigned p2 = pk_bf16(r1, 0.0f);
  const float r2 = r1 - bf16_lo(p2);
  const unsigned p3 = pk_bf16(r2, 0.0f);
  l1 = (unsigned short)(p1 & 0xffffu);
  l2 = (unsigned short)(p2 & 0xffffu);
  l3 = (unsigned short)(p3 & 0xffffu);
}

constexpr int BF3_SLOT = 16;                        // bytes per (lane) slot
constexpr int BF3_CHUNK = 3 * 2 * 32 * BF3_SLOT;    // bytes per chunk: 3 limbs x 2 halves x 32 lanes = 3 KB

// A operand of limb `limb` of chunk `chunk` for this lane; `img` is the LDS image (+ the opaque zero that keeps the
// reads inside the sample loop)
LP_DEV u32x4_t bf3_a(const char* img, int chunk, int limb, int lane) {
  return *reinterpret_cast<const u32x4_t*>(img + chunk * BF3_CHUNK + (limb * 2 + (lane >> 5)) * (32 * BF3_SLOT) + (lane & 31) * BF3_SLOT);
}

// One chunk (8 values) -> its three limbs, 4 packed dwords each (22 VALU instructions per pair of values x 4)
LP_DEV void split3_chunk(const float* v, u32x4_t& l1, u32x4_t& l2, u32x4_t& l3) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned p1 = pk_bf16(a, b);
    const float ra = a - bf16_lo(p1), rb = b - bf16_hi(p1);
    const unsigned p2 = pk_bf16(ra, rb);
    l1[i] = p1;
    l2[i] = p2;
    l3[i] = pk_bf16(ra - bf16_lo(p2), rb - bf16_hi(p2));
  }
}
// ---- A-operand loaders -------------------------------------------------------------------------------------------
// (a) slot image (stage_matrix_bf3): one ds_read_b128 per limb and chunk; one image per orientation
struct ASlots {
  const char* img;
  int chunk0;
  LP_DEV u32x4_t operator()(int c, int limb, int lane) const { return bf3_a(img, chunk0 + c, limb, lane); }
};
// (b) ONE row-major image per layer serving both orientations: limb p of W[row k_in][col m_out] at
//     layer + p * limb_stride + rm_off(k_in, m_out).  Layout (round 5; rounds 2-4: plain 72-byte rows):
//     * rows are 64 bytes (32 bf16); every group of four rows is skewed by a further 16 bytes: row k starts at byte
//       64 k + 16 (k >> 2);
//     * inside every 16-column chunk the four 4-column blocks are stored in the order 0, 2, 1, 3, so that the eight columns
//       a backward lane needs -- 16c + 4h .. +3 and 16c + 8 + 4h .. +3 -- are 16 CONTIGUOUS bytes.
//     forward (two ds_read_b64_tr_b16 per lane): one instruction covers four consecutive rows k0 .. k0 + 3 (k0 a multiple of 4)
//       completely -- 4 x 64 B on four disjoint 16-bank ranges of the 64 four-byte banks: conflict-free (72-byte rows: the four
//       rows span 288 B and wrap onto their own first banks, a 2-way conflict on EVERY forward operand read: 84 LDS cycles per
//       wave-sample of the tuned backward, 336 of the 2/2/2 x 64 forward -- scripts/tr_b16_pmc.sh);
//     backward: ONE ds_read_b128 per lane (was: two ds_read_b64, which the compiler fuses into a ds_read2_b64 -- 32 banks, 16-lane
//       groups: with 64-byte rows those conflict); the four non-contiguous 16-lane groups of a ds_read_b128 each see start banks
//       16 (k & 3) + 4 (k >> 2) that do not overlap (all four groups, both loaders: scripts/lds_bank_model.py; measured: profiles/r05_lds_bank_conflicts.txt).
__host__ __device__ constexpr int rm_off(int k, int m) {
  return k * 64 + (k >> 2) * 16 + ((m >> 4) * 16 + ((((m >> 2) & 1) << 1) | ((m >> 3) & 1)) * 4 + (m & 3)) * 2;
}
__host__ __device__ constexpr int rm_bytes(int rows) { return rows * 64 + ((rows + 3) >> 2) * 16; }
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
//     backward (dX) form: lane (k = l & 31, h) needs W[k][feat(8c + j, h)], j = 0..7 = columns 16c + 4h .. +3 and 16c + 8 + 4h .. +3
struct ARowsBwd {
  const char* layer;
  int limb_stride, row_mask;  // row_mask = rows - 1: lanes beyond a 16-row matrix re-read valid rows (their output rows are unused)
  LP_DEV u32x4_t operator()(int c, int limb, int lane) const {
    // columns 16c + 4h .. +3 and 16c + 8 + 4h .. +3 = blocks h and 2 + h of the chunk = physical blocks 2h, 2h + 1: 16 bytes
    return *reinterpret_cast<const u32x4_t*>(layer + limb * limb_stride + rm_off((lane & 31) & row_mask, 16 * c + 4 * (lane >> 5)));
  }
};
//     forward form: lane (m = l & 31, h) needs W[feat(8c + j, h)][m] = column m of rows 16c + 4h .. +3 and 16c + 8 + 4h .. +3:
//     two ds_read_b64_tr_b16.  Semantics (scripts/tr_b16_probe.hip, profiles/r02_tr_b16_probe.txt): inside a group of 16
//     lanes, lane i receives element (i & 3) of the four b16 supplied by lanes 4j + (i >> 2), j = 0..3.  So supplier lane
//     s = 4j + q addresses row (base + j), columns m0 + 4q .. +3 (m0 = first column of the group), and lane i ends up with
//     rows base .. base + 3 of column m0 + i.
struct AColsFwd {
  const char* layer;
  int limb_stride;
  LP_DEV u32x4_t operator()(int c, int limb, int lane) const {
    const int s = lane & 15, m0 = lane & 16, h = lane >> 5;
    const char* p = layer + limb * limb_stride + rm_off(16 * c + 4 * h + (s >> 2), m0 + 4 * (s & 3));
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + (rm_off(8, 0) - rm_off(0, 0))));  // rows + 8
    const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
    return (u32x4_t){ua.x, ua.y, ub.x, ub.y};
  }
};
// stage W [rows x 32] (row-major in mlp_params at off, leading dimension ld) as the three row-major limb images
LP_DEV void stage_matrix_rm(char* layer, int limb_stride, const float* P, int64_t off, int rows, int ld, int tid, int n_threads) {
  for (int i = tid; i < rows * 32; i += n_threads) {
    const int k = i >> 5, m = i & 31;
    unsigned short l1, l2, l3;
    split3_scalar(P[off + (int64_t)k * ld + m], l1, l2, l3);
    char* base = layer + rm_off(k, m);
    *reinterpret_cast<unsigned short*>(base) = l1;
    *reinterpret_cast<unsigned short*>(base + limb_stride) = l2;
    *reinterpret_cast<unsigned short*>(base + 2 * limb_stride) = l3;
  }
}

// the six limb products of one chunk, one A operand live at a time
template <class A>
LP_DEV f32x16 chunk_bf3(const A& a, int c, int lane, const u32x4_t& l1, const u32x4_t& l2, const u32x4_t& l3, f32x16 acc) {
  {
    const u32x4_t w3 = a(c, 2, lane);
    acc = LP_MFMA_BF16(w3, l1, acc);
  }
  {
    const u32x4_t w2 = a(c, 1, lane);
    acc = LP_MFMA_BF16(w2, l2, acc);
    acc = LP_MFMA_BF16(w2, l1, acc);
  }
  {
    const u32x4_t w1 = a(c, 0, lane);
    acc = LP_MFMA_BF16(w1, l3, acc);
    acc = LP_MFMA_BF16(w1, l2, acc);
    acc = LP_MFMA_BF16(w1, l1, acc);
  }
  return acc;
}
// issue order "1 MFMA, ~7 VALU" x 6: the split of the next chunk (44 VALU) rides in the shadow of this chunk's six
// dependent MFMAs (6 x 32 cycles)
LP_DEV void bf3_interleave_hint() {
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read (A operand)
    __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);  // VALU
  }
}

// The same product chunk by chunk straight from the fp32 values, software-pipelined: the limbs of chunk c + 1 are split
// while the MFMAs of chunk c run; only one chunk's limbs (+ the next one's being formed) are live.
template <int NCH, class A>
LP_DEV f32x16 layer_bf3v(const A& a, int lane, const float (&v)[8 * NCH], f32x16 acc) {
  u32x4_t l1, l2, l3;
  split3_chunk(v, l1, l2, l3);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4_t n1 = l1, n2 = l2, n3 = l3;
    acc = chunk_bf3(a, c, lane, l1, l2, l3, acc);
    if (c + 1 < NCH) {
      split3_chunk(v + 8 * (c + 1), n1, n2, n3);
      bf3_interleave_hint();
    }
    l1 = n1; l2 = n2; l3 = n3;
  }
  return acc;
}
// ---- two-limb gradient operand of the dX chains (default; -DLP_DX_LIMBS=3 restores three limbs) ---------------------------
// dX = W dY back-propagates a GRADIENT: dY is split into two limbs (16 significand bits, relative error 2^-17 per value), the
// weights keep theirs; products kept: w1 y1, w2 y1, w1 y2 (dropped terms <= ~3 * 2^-17 |w y|, random in sign).  Half the MFMAs
// of the dX chains and 24 instead of 44 split instructions per chunk: the tuned backward 2.03 -> 1.88 ms at cfg 2, 125.5 ->
// 117.8 ms at cfg 4 (profiles/r05_dx_limbs_ab.txt).  What it costs in accuracy was measured with the forced-oracle proof
// (tests/test_gpu_config_scale.py::test_flips_are_flips, every gradient entry of the cfg-2 launch against fp64): worst error
// 1.04e-5 with two limbs, 8.7e-6 with three -- the per-product error averages out over the 10^2 .. 10^4 ray-samples every
// gradient entry sums, the fp32 atomics' summation order dominates either way.  The FORWARD products (the outputs, the
// backward's recompute with its ReLU decisions) stay three-limb = fp32-equivalent.  The WEIGHT gradients of the tuned family and of
// the one-wave looped kernels take two limbs per operand as well (dw_quadrant_bf, lp_renderer_mfma_bwd.h: v_mfma_f32_16x16x32_bf16,
// three limb products; measured on the same launch: 1.05e-5, profiles/r05_dw_bf16_ab.txt); the two-waves-per-SIMD shallow looped
// kernels keep fp32 quadrants (v_mfma_f32_16x16x4_f32).  LpRendererArgs.arithmetic = LP_ARITH_FP32 selects, per call, instantiations
// with three limbs in the dX chains and fp32 quadrants (the reference's arithmetic); lp_build_info() reports what was compiled.
#ifndef LP_DX_LIMBS
#define LP_DX_LIMBS 2
#endif
LP_DEV void split2_chunk(const float* v, u32x4_t& l1, u32x4_t& l2) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = v[2 * i], b = v[2 * i + 1];
    const unsigned p1 = pk_bf16(a, b);
    l1[i] = p1;
    l2[i] = pk_bf16(a - bf16_lo(p1), b - bf16_hi(p1));
  }
}
template <class A>
LP_DEV f32x16 chunk_bf2(const A& a, int c, int lane, const u32x4_t& l1, const u32x4_t& l2, f32x16 acc) {
  {
    const u32x4_t w2 = a(c, 1, lane);
    acc = LP_MFMA_BF16(w2, l1, acc);
  }
  {
    const u32x4_t w1 = a(c, 0, lane);
    acc = LP_MFMA_BF16(w1, l2, acc);
    acc = LP_MFMA_BF16(w1, l1, acc);
  }
  return acc;
}
// `trow` (optional): this lane's row of a two-limb bf16 tile [ray][feature] in the rm_off layout (limb 2 at + rm_bytes(32)): the
// limbs the chain forms anyway are published there for the weight-gradient products (limb_tile_store below) -- no extra VALU.
template <int NCH, class A>
LP_DEV f32x16 layer_bf2v(const A& a, int lane, const float (&v)[8 * NCH], f32x16 acc, char* trow = nullptr) {
  u32x4_t l1, l2;
  split2_chunk(v, l1, l2);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4_t n1 = l1, n2 = l2;
    if (trow) {
      *reinterpret_cast<u32x4_t*>(trow + 32 * c) = l1;
      *reinterpret_cast<u32x4_t*>(trow + 32 * c + rm_bytes(32)) = l2;
    }
    acc = chunk_bf2(a, c, lane, l1, l2, acc);
    if (c + 1 < NCH) split2_chunk(v + 8 * (c + 1), n1, n2);
    l1 = n1; l2 = n2;
  }
  return acc;
}
// Two-limb bf16 tile [ray][feature] of one wave (rm_off layout, limb 2 at + rm_bytes(32)): lane (h, r) owns ray r and, per chunk
// c, the features 16c + 4h .. +3 and 16c + 8 + 4h .. +3 -- which the layout stores as 16 contiguous bytes (ONE ds_write_b128 per
// limb and chunk).  trow = tile + rm_off(r, 4 h).
// (The values pass an empty asm: without it LLVM recognises that the forward products split the same activations a few hundred
// instructions earlier, re-uses THOSE limbs and keeps ten 4-register tuples alive across the whole sample -- 49 spilled registers in
// the tuned backward, 2.07 instead of 1.8 ms; splitting again costs 24 VALU instructions per chunk and no register.)
template <int NCH>
LP_DEV void limb_tile_store(char* trow, const float (&v)[8 * NCH]) {
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4_t l1, l2;
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      w[i] = v[8 * c + i];
      asm volatile("" : "+v"(w[i]));
    }
    split2_chunk(w, l1, l2);
    *reinterpret_cast<u32x4_t*>(trow + 32 * c) = l1;
    *reinterpret_cast<u32x4_t*>(trow + 32 * c + rm_bytes(32)) = l2;
  }
}

// the dX chains' operand split / chunk product with DXL = 2 or 3 limbs of the gradient operand (l3 unused with two)
template <int DXL>
LP_DEV void dx_split_chunk(const float* v, u32x4_t& l1, u32x4_t& l2, u32x4_t& l3) {
  if constexpr (DXL == 2) {
    split2_chunk(v, l1, l2);
    l3 = l2;
  } else {
    split3_chunk(v, l1, l2, l3);
  }
}
template <int DXL, class A>
LP_DEV f32x16 dx_chunk(const A& a, int c, int lane, const u32x4_t& l1, const u32x4_t& l2, const u32x4_t& l3, f32x16 acc) {
  if constexpr (DXL == 2) return chunk_bf2(a, c, lane, l1, l2, acc);
  else return chunk_bf3(a, c, lane, l1, l2, l3, acc);
}
template <int NCH, int DXL = LP_DX_LIMBS, class A>
LP_DEV f32x16 layer_dxv(const A& a, int lane, const float (&v)[8 * NCH], f32x16 acc, char* trow = nullptr) {
  if constexpr (DXL == 2) return layer_bf2v<NCH>(a, lane, v, acc, trow);
  else return layer_bf3v<NCH>(a, lane, v, acc);
}

template <int NCH>
LP_DEV f32x16 layer_bf3v(const char* img, int chunk0, int lane, const float (&v)[8 * NCH], f32x16 acc) {
  return layer_bf3v<NCH>(ASlots{img, chunk0}, lane, v, acc);
}

// Two layers that read the SAME input (the opacity and the colour hidden layer both read e): every chunk is split once
// and feeds two independent accumulator chains (which also keeps the matrix pipe busy back to back).
template <int NCH, class A1, class A2>
LP_DEV void layer2_bf3v(const A1& a1, const A2& a2, int lane, const float (&v)[8 * NCH], f32x16& acc_a, f32x16& acc_b) {
  u32x4_t l1, l2, l3;
  split3_chunk(v, l1, l2, l3);
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    u32x4_t n1 = l1, n2 = l2, n3 = l3;
    acc_a = chunk_bf3(a1, c, lane, l1, l2, l3, acc_a);
    acc_b = chunk_bf3(a2, c, lane, l1, l2, l3, acc_b);
    if (c + 1 < NCH) {
      split3_chunk(v + 8 * (c + 1), n1, n2, n3);
      bf3_interleave_hint();
    }
    l1 = n1; l2 = n2; l3 = n3;
  }
}
template <int NCH>
LP_DEV void layer2_bf3v(const char* img, int chunk_a, int chunk_b, int lane, const float (&v)[8 * NCH], f32x16& acc_a,
                        f32x16& acc_b) {
  layer2_bf3v<NCH>(ASlots{img, chunk_a}, ASlots{img, chunk_b}, lane, v, acc_a, acc_b);
}

// Stage one weight matrix W [rows_in x 32] (row-major, leading dimension ld, inside mlp_params at `off`) as limb
// image chunks.  FWD: slot (chunk c, half h, lane m, j) = W[feat(8c + j, h)][m]; otherwise (backward / dX form)
// W[m][feat(8c + j, h)] with rows m >= rows_in zero.  n_chunks = rows_in / 16 (forward) or 2 (backward).
template <bool FWD>
LP_DEV void stage_matrix_bf3(char* img, int chunk0, int n_chunks, const float* P, int64_t off, int rows_in, int ld, int tid,
                             int n_threads) {
  for (int i = tid; i < n_chunks * 2 * 32 * 8; i += n_threads) {
    const int j = i & 7, m = (i >> 3) & 31, h = (i >> 8) & 1, c = i >> 9;
    const int f = featq(8 * c + j, h);
    float w;
    if (FWD) w = (f < rows_in) ? P[off + (int64_t)f * ld + m] : 0.0f;
    else w = (m < rows_in) ? P[off + (int64_t)m * ld + f] : 0.0f;
    unsigned short l1, l2, l3;
    split3_scalar(w, l1, l2, l3);
    char* base = img + (chunk0 + c) * BF3_CHUNK + (h * 32 + m) * BF3_SLOT + j * 2;
    *reinterpret_cast<unsigned short*>(base) = l1;
    *reinterpret_cast<unsigned short*>(base + 2 * 32 * BF3_SLOT) = l2;
    *reinterpret_cast<unsigned short*>(base + 4 * 32 * BF3_SLOT) = l3;
  }
}

// LDS map of the bf16x3 kernels of the default shape (trunk [C,32,32], heads [32,32,.]):
//   [0, F32_END)           the fp32 small stuff of Lds (biases, output layers of the heads, beyond-far table); the four
//                          32 x 33 fp32 matrices of the fp32 kernels are NOT staged
//   forward image          chunks: t1 (C/16), t2 (2), o1 (2), c1 (2)
//   backward image         chunks: c1, o1, t2, t1 (2 each; only the backward kernel stages it)
template <int C>
struct LdsBf3 {
  static constexpr int SMALL = Lds::BIAS;                     // float offset of the reused small block inside Lds
  static constexpr int SMALL_FLOATS = Lds::FWD_END - Lds::BIAS;
  static constexpr int FWD_IMG = SMALL_FLOATS * 4;            // byte offset of the forward image
  static constexpr int CH_T1 = 0, CH_T2 = C / 16, CH_O1 = CH_T2 + 2, CH_C1 = CH_O1 + 2, N_FWD = CH_C1 + 2;
  static constexpr int BWD_IMG = FWD_IMG + N_FWD * BF3_CHUNK;
  static constexpr int CB_C1 = 0, CB_O1 = 2, CB_T2 = 4, CB_T1 = 6, N_BWD = 8;
  static constexpr int FWD_END = BWD_IMG;                     // bytes, forward kernels
  static constexpr int BWD_END = BWD_IMG + N_BWD * BF3_CHUNK; // bytes, backward kernel (before the per-wave tiles)
};

// stage the small fp32 block (at float offset 0 of `lds`: indices are Lds::X - Lds::BIAS; n_inf entries of the beyond-far table)
LP_DEV void stage_small_bf3(const LpRendererArgs& a, const MfmaParams& mp, float* lds, int n_inf, int n_threads) {
  using M = Lds;
  const float* P = a.mlp_params;
  const int tid = threadIdx.x;
  float* sm = lds - M::BIAS;  // so that sm[M::X] addresses the small block
  for (int i = tid; i < 32; i += n_threads) {
    sm[M::BIAS + i] = P[mp.b_t1 + i];
    sm[M::BIAS + 32 + i] = P[mp.b_t2 + i];
    sm[M::BIAS + 64 + i] = P[mp.b_o1 + i];
    sm[M::BIAS + 96 + i] = P[mp.b_c1 + i];
    sm[M::WO2 + i] = P[mp.w_o2 + i];
#pragma unroll
    for (int c = 0; c < 4; ++c) sm[M::WC2 + i * 4 + c] = (c < a.color_chn) ? P[mp.w_c2 + (int64_t)i * mp.ldc2 + c] : 0.0f;
  }
  for (int i = tid; i < n_inf; i += n_threads) sm[M::INF + i] = (i < a.march.num_samples_inf) ? inf_scale(i, a.march) : 0.0f;
  if (tid == 0) {
    sm[M::HB + 0] = P[mp.b_o2];
#pragma unroll
    for (int c = 0; c < 4; ++c) sm[M::HB + 1 + c] = (c < a.color_chn) ? P[mp.b_c2 + c] : 0.0f;
  }
}

// small block + the forward slot image (+ the backward one)
template <int C>
LP_DEV void stage_weights_bf3(const LpRendererArgs& a, const MfmaParams& mp, float* lds, bool with_backward, int n_threads) {
  const float* P = a.mlp_params;
  const int tid = threadIdx.x;
  stage_small_bf3(a, mp, lds, MAX_INF, n_threads);
  using L = LdsBf3<C>;
  char* img = reinterpret_cast<char*>(lds) + L::FWD_IMG;
  stage_matrix_bf3<true>(img, L::CH_T1, C / 16, P, mp.w_t1, C, 32, tid, n_threads);
  stage_matrix_bf3<true>(img, L::CH_T2, 2, P, mp.w_t2, 32, 32, tid, n_threads);
  stage_matrix_bf3<true>(img, L::CH_O1, 2, P, mp.w_o1, 32, 32, tid, n_threads);
  stage_matrix_bf3<true>(img, L::CH_C1, 2, P, mp.w_c1, 32, 32, tid, n_threads);
  if (with_backward) {
    char* bimg = reinterpret_cast<char*>(lds) + L::BWD_IMG;
    stage_matrix_bf3<false>(bimg, L::CB_C1, 2, P, mp.w_c1, 32, 32, tid, n_threads);
    stage_matrix_bf3<false>(bimg, L::CB_O1, 2, P, mp.w_o1, 32, 32, tid, n_threads);
    stage_matrix_bf3<false>(bimg, L::CB_T2, 2, P, mp.w_t2, 32, 32, tid, n_threads);
    stage_matrix_bf3<false>(bimg, L::CB_T1, 2, P, mp.w_t1, C, 32, tid, n_threads);
  }
}

// LDS map of the 4-wave bf16x3 backward: small block with a 64-entry beyond-far table, then ONE row-major limb image per
// layer (both orientations are read from it, see ARowsBwd / AColsFwd): 24 KB with C = 16 instead of 45 KB of slot images
template <int C>
struct LdsBf3Rm {
  static constexpr int N_INF = 64;
  static constexpr int SMALL_BYTES = (Lds::INF - Lds::BIAS + N_INF) * 4;
  static constexpr int ST_T1 = rm_bytes(C), ST_32 = rm_bytes(32);          // limb strides (bytes)
  static constexpr int IMG = SMALL_BYTES;
  static constexpr int L_T1 = IMG, L_T2 = L_T1 + 3 * ST_T1, L_O1 = L_T2 + 3 * ST_32, L_C1 = L_O1 + 3 * ST_32;
  static constexpr int END = L_C1 + 3 * ST_32;
  static_assert(SMALL_BYTES % 16 == 0 && END % 16 == 0, "16-byte alignment of the LDS regions");
};
template <int C>
LP_DEV void stage_weights_rm(const LpRendererArgs& a, const MfmaParams& mp, float* lds, int n_threads) {
  using R = LdsBf3Rm<C>;
  stage_small_bf3(a, mp, lds, R::N_INF, n_threads);
  char* b = reinterpret_cast<char*>(lds);
  const float* P = a.mlp_params;
  stage_matrix_rm(b + R::L_T1, R::ST_T1, P, mp.w_t1, C, 32, threadIdx.x, n_threads);
  stage_matrix_rm(b + R::L_T2, R::ST_32, P, mp.w_t2, 32, 32, threadIdx.x, n_threads);
  stage_matrix_rm(b + R::L_O1, R::ST_32, P, mp.w_o1, 32, 32, threadIdx.x, n_threads);
  stage_matrix_rm(b + R::L_C1, R::ST_32, P, mp.w_c1, 32, 32, threadIdx.x, n_threads);
}

// bias of layer `which` (0 t1, 1 t2, 2 o1, 3 c1) in accumulator order; `sm` = small block base (sm[Lds::X] valid)
LP_DEV f32x16 load_bias_bf3(const float* sm, int which, int h, int zo) { return load_bias(sm, which, h, zo); }

// Per-ray pre-activation of the colour hidden layer: cb = b_c1 + W_c1^T enc.  relu(W^T (e + enc) + b) = relu(W^T e + cb),
// so the colour layer reuses the limbs of e (already split for the opacity layer) instead of splitting e + enc again.
template <class A>
LP_DEV void color_prebias_bf3(const float* sm, const A& a_c1, int lane, const float (&enc)[16], float (&cb)[16]) {
  const f32x16 acc = layer_bf3v<2>(a_c1, lane, enc, load_bias_bf3(sm, 3, lane >> 5, 0));
#pragma unroll
  for (int q = 0; q < 16; ++q) cb[q] = acc[q];
}
template <int C>
LP_DEV void color_prebias_bf3(const float* sm, const char* fimg, int lane, const float (&enc)[16], float (&cb)[16]) {
  color_prebias_bf3(sm, ASlots{fimg, LdsBf3<C>::CH_C1}, lane, enc, cb);
}

// Decoder of one sample, default shape.  t.x0 in; fills t.h1 / t.e / t.ho / t.hc (post-ReLU).
template <int C, int NC>
LP_DEV Heads decode_bf3(const float* sm, const char* fimg_, int lane, const float (&cb)[16], Act<C>& t, int zo) {
  using L = LdsBf3<C>;
  const int h = lane >> 5;
  const char* fimg = fimg_ + zo;
  f32x16 acc = layer_bf3v<C / 16>(fimg, L::CH_T1, lane, t.x0, load_bias_bf3(sm, 0, h, zo));
#pragma unroll
  for (int q = 0; q < 16; ++q) t.h1[q] = relu_f(acc[q]);
  acc = layer_bf3v<2>(fimg, L::CH_T2, lane, t.h1, load_bias_bf3(sm, 1, h, zo));
#pragma unroll
  for (int q = 0; q < 16; ++q) t.e[q] = relu_f(acc[q]);
  f32x16 acc_o = load_bias_bf3(sm, 2, h, zo), acc_c;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc_c[q] = cb[q];
  layer2_bf3v<2>(fimg, L::CH_O1, L::CH_C1, lane, t.e, acc_o, acc_c);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    t.ho[q] = relu_f(acc_o[q]);
    t.hc[q] = relu_f(acc_c[q]);
  }
  return heads_forward<NC>(sm, h, t.ho, t.hc, zo);
}

}  // namespace lp
