// lp_dw_bf16.h -- the weight gradients dW = X^T dY of the decoder's 32-wide layers on the bf16 matrix cores.
//
// Why.  dW contracts over RAYS, and a layer's rays are spread over the four waves of a workgroup, so X and dY cross
// waves through LDS.  As fp32 tiles + v_mfma_f32_16x16x4_f32 that is 32 MFMAs (1 024 cycles) per layer and wave which no
// VALU instruction overlaps with (scripts/mfma_valu_overlap.hip): about a fifth of the backward's SIMD time.  Here the
// tiles hold the two leading bf16 limbs of every value (x ~= x1 + x2, 16 significand bits; the limbs already exist: the
// recompute / dX chains of lp_bf3.h split every activation and every gradient into limbs for their own products), and a
// layer's quadrant is 4 source waves x 3 v_mfma_f32_16x16x32_bf16 (x1 y1 + x1 y2 + x2 y1; 16 cycles each, on the
// pipe that does overlap with VALU work), accumulated in fp32.  Per-term relative error <= 3 * 2^-16 (the dropped
// x2 y2, x1 y3, x3 y1), unbiased -- against the reference's own GPU arithmetic (Triton tl.dot on fp32 inputs = TF32, 10 bits) 64x
// tighter; the CPU oracle comparison in tests/ holds grad_mlp_params to 1e-4 of its largest entry as before.
//
// Tile layout.  Per wave and limb one tile [32 rays][32 features] bf16, 64 B per ray, no padding; the 8-byte chunk
// (4 features) k of ray r lives at chunk position k ^ swz(r), swz(r) = r.bit2 | r.bit4 << 1 | r.bit3 << 2:
//   * producer: lane (h, r) owns features feat(q, h) of ray r -- per limb four 8-byte chunks (k = 4c + 2t + h), written
//     with ds_write_b64 straight from the packed limb registers of lp_bf3.h; 32 lanes of a half write 32 different
//     (bank quad, chunk) pairs: conflict-free;
//   * consumer: the MFMA wants, per lane (m = l & 15, kg = l >> 4), eight consecutive K (= rays 8 kg .. 8 kg + 7) of
//     feature m: two ds_read_b64_tr_b16 (rays 8 kg + j and 8 kg + 4 + j, j < 4; semantics in lp_bf3.h / AColsFwd).
//     Lanes 0..31 touch rays {0..3, 8..11} x 4 chunks, whose swizzles differ in the bit that selects the chunk half:
//     64 different banks.
// A and B use the same ray <-> K mapping, so the K order inside the instruction does not matter.
// Bias gradients ride on the same B operands: A = the indicator of row `layer` (1.0 in every K of row l, else 0) makes
// row l of ONE shared 16 x 16 accumulator the column sums of layer l's dY (2 MFMAs per source wave).
#pragma once
#include "lp_bf3.h"

namespace lp {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

#define LP_MFMA16_BF16(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, (a)), __builtin_bit_cast(bf16x8_t, (b)), (c), 0, 0, 0)

constexpr int LT_ROW = 64;            // bytes per ray: 32 features x bf16
constexpr int LT_TILE = 32 * LT_ROW;  // one limb tile: 2 KB
struct LimbTiles {                    // per-wave area (bytes)
  static constexpr int X_HI = 0, X_MID = LT_TILE, Y_HI = 2 * LT_TILE, Y_MID = 3 * LT_TILE, BYTES = 4 * LT_TILE;
};

LP_DEV int lt_swz(int row) { return ((row >> 2) & 1) | (((row >> 4) & 1) << 1) | (((row >> 3) & 1) << 2); }
LP_DEV int lt_off(int row, int k) { return row * LT_ROW + ((k ^ lt_swz(row)) << 3); }

// producer side of one lane: byte offsets of its row and the swizzle, fixed for the kernel's lifetime
struct LtWriter {
  int row_b, swz8, h8;
  LP_DEV explicit LtWriter(int lane) : row_b((lane & 31) * LT_ROW), swz8(lt_swz(lane & 31) << 3), h8((lane >> 5) << 3) {}
  // chunk c (values 8c .. 8c+7 of the lane = features 16c + 4h .. +3 and 16c + 8 + 4h .. +3), limbs l1 / l2
  LP_DEV void store(char* tile_hi, int c, const u32x4_t& l1, const u32x4_t& l2) const {
    const int o0 = row_b + ((32 * c + h8) ^ swz8), o1 = row_b + ((32 * c + 16 + h8) ^ swz8);
    *reinterpret_cast<u32x2_t*>(tile_hi + o0) = (u32x2_t){l1[0], l1[1]};
    *reinterpret_cast<u32x2_t*>(tile_hi + o1) = (u32x2_t){l1[2], l1[3]};
    *reinterpret_cast<u32x2_t*>(tile_hi + LT_TILE + o0) = (u32x2_t){l2[0], l2[1]};
    *reinterpret_cast<u32x2_t*>(tile_hi + LT_TILE + o1) = (u32x2_t){l2[2], l2[3]};
  }
};

// consumer side of one lane: ONE register -- the offset of its first transposed read for feature block 0; block 1 is
// `^ 32` (bit 2 of the chunk index), the second read (rays + 4: swz bit 0 flips) is `(+ 256) ^ 8`
struct LtReader {
  int base;
  LP_DEV explicit LtReader(int lane) {
    const int s = lane & 15, j = s >> 2, q = s & 3, kg = lane >> 4;
    base = lt_off(8 * kg + j, q);
  }
  // `blk` (0 / 1: features 0..15 / 16..31) is wave-uniform
  LP_DEV u32x4_t load(const char* tile, int blk) const {
    typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
    const int o0 = base ^ (blk << 5), o1 = (o0 + 256) ^ 8;
    const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + o0));
    const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(tile + o1));
    const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
    return (u32x4_t){ua.x, ua.y, ub.x, ub.y};
  }
};

// indicator of row `l` as an A operand: bf16 1.0 in all eight K of the lanes with (lane & 15) == l
// (`zo`: the opaque zero of lp_mfma_common.h -- keeps the four indicators from being hoisted out of the sample loop as
// sixteen live registers)
LP_DEV u32x4_t lt_row_indicator(int lane, int l, int zo = 0) {
  const unsigned one = ((lane & 15) == l + zo) ? 0x3f803f80u : 0u;
  return (u32x4_t){one, one, one, one};
}

// dq += X^T dY over the rays of the source waves [v0, v1) (wave areas `stride` bytes apart, `area0` = wave 0's), feature
// blocks blk_x / blk_y; BIAS: the row of dbq selected by `ind` += column sums of dY.  Register i of lane
// (n = l & 15, g = l >> 4) is entry [16 blk_x + 4 g + i][16 blk_y + n] of the layer's dW.
// Products x1 y1 + x1 y2 + x2 y1: x2 y2 is of the order of the dropped x1 y3 / x3 y1 (2^-16).
template <bool BIAS>
LP_DEV void dw_quadrant_bf16(const char* area0, int stride, const LtReader& rd, int blk_x, int blk_y, int v0, int v1, f32x4_t& dq,
                             f32x4_t& dbq, const u32x4_t& ind) {
#pragma unroll 1  // one source wave's operands live at a time
  for (int v = v0; v < v1; ++v) {
    const char* base = area0 + v * stride;
    const u32x4_t a1 = rd.load(base + LimbTiles::X_HI, blk_x), b2 = rd.load(base + LimbTiles::Y_MID, blk_y);
    dq = LP_MFMA16_BF16(a1, b2, dq);
    if (BIAS) dbq = LP_MFMA16_BF16(ind, b2, dbq);
    const u32x4_t b1 = rd.load(base + LimbTiles::Y_HI, blk_y);
    dq = LP_MFMA16_BF16(a1, b1, dq);
    if (BIAS) dbq = LP_MFMA16_BF16(ind, b1, dbq);
    const u32x4_t a2 = rd.load(base + LimbTiles::X_MID, blk_x);
    dq = LP_MFMA16_BF16(a2, b1, dq);
  }
}

}  // namespace lp
