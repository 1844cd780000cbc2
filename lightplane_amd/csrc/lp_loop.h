// lp_loop.h -- building blocks of the LAYER-LOOPED bf16x3 MFMA families (Renderer: lp_renderer_loop.hip, MLP-Splatter:
// lp_splatter_mlp_loop.hip): one dense layer of any [rows_in x cols] shape up to 64 x 64 as 32 x 32 blocks of row-major limb
// images in LDS, forward (transposed reads), dX (plain reads) and the workgroup-shared dW quadrants.
// Lane mapping and arithmetic: lp_bf3.h / lp_renderer_mfma.hip.
#pragma once
#include "lp_bf3.h"
#include "lp_mfma_common.h"

namespace lp {

typedef float f32x4l __attribute__((ext_vector_type(4)));
#define LP_MFMA16L(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int LOOP_N_INF = 256;     // beyond-far samples tabulated (as many as the tuned family, lp_mfma_common.h MAX_INF)
constexpr int LOOP_ST = rm_bytes(32);     // bytes of one limb of a 32 x 32 block (skewed 64-byte rows, lp_bf3.h)
constexpr int LOOP_BLK = 3 * LOOP_ST;     // bytes of a block image (three limbs)
constexpr int LT_LD = 36;           // row stride of the feature-major fp32 tiles [32 features][32 rays + 4]

struct LoopLayer {
  int64_t w, b;   // float offsets inside mlp_params: W [rows_in x cols] row-major, b [cols]
  int rows_in;    // input width (grid channels or hidden width)
  int cols;       // output width (hidden width)
  int ld;         // row stride of W inside mlp_params (= cols, except for a padded output layer)
  int ob;         // output blocks of 32 features: ceil(cols / 32)
  int img;        // byte offset of the layer's block images in LDS: block (ib, ob) at img + (ib * ob_count + ob) * LOOP_BLK
  int bias;       // float index of the bias (zero-padded to 32 * NB) in the small block
};
// bytes of a layer's block images
inline int loop_layer_bytes(int rows_in, int cols) { return ((rows_in + 31) / 32) * ((cols + 31) / 32) * LOOP_BLK; }

LP_DEV constexpr int pi16l(int m) { return m < 4 ? 2 * m : (m < 12 ? 2 * (m - 4) + 1 : 2 * (m - 8)); }
#ifdef LP_TIMING_NO_BARRIER  // timing experiment (wrong results): what the workgroup barriers of the dW rounds cost
LP_DEV void lds_barrier_l() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#else
LP_DEV void lds_barrier_l() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#endif

// ---------------------------------------------------------------------------------------------------------------
// staging
// ---------------------------------------------------------------------------------------------------------------
// (the staging loops stride by the launch's block size: the eight-wave forwards stage every element once, not twice)
template <int NB>
LP_DEV void loop_stage_layer(char* lds, float* sm, const float* P, const LoopLayer& L, int tid) {
  const int in_blocks = (L.rows_in + 31) >> 5;
  for (int ib = 0; ib < in_blocks; ++ib) {
    for (int ob = 0; ob < L.ob; ++ob) {
      char* blk = lds + L.img + (ib * L.ob + ob) * LOOP_BLK;
      for (int i = tid; i < 32 * 32; i += (int)blockDim.x) {
        const int k = i >> 5, m = i & 31;
        const int row = 32 * ib + k, col = 32 * ob + m;
        const float w = (row < L.rows_in && col < L.cols) ? P[L.w + (int64_t)row * L.ld + col] : 0.0f;
        unsigned short l1, l2, l3;
        split3_scalar(w, l1, l2, l3);
        char* base = blk + rm_off(k, m);
        *reinterpret_cast<unsigned short*>(base) = l1;
        *reinterpret_cast<unsigned short*>(base + LOOP_ST) = l2;
        *reinterpret_cast<unsigned short*>(base + 2 * LOOP_ST) = l3;
      }
    }
  }
  for (int i = tid; i < 32 * NB; i += (int)blockDim.x) sm[L.bias + i] = (i < L.cols) ? P[L.b + i] : 0.0f;
}

// ---------------------------------------------------------------------------------------------------------------
// one layer on the matrix cores
// ---------------------------------------------------------------------------------------------------------------
// issue order "1 MFMA, 1 LDS operand read, a few VALU" for the 6 NB MFMAs of a chunk and the 44 VALU instructions of the next
// chunk's limb split (see bf3_interleave_hint, lp_bf3.h)
template <int NB, int NP = 6>  // NP = limb products per chunk and block
LP_DEV void loop_interleave_hint() {
#pragma unroll
  for (int i = 0; i < NP * NB; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);               // MFMA
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // DS read (A operand)
    __builtin_amdgcn_sched_group_barrier(0x002, NB == 1 ? 8 : 4, 0);  // VALU
  }
}

// out = relu(W^T in + b) (RELU = false: the affine map only): `in` / `out` are NB blocks of this lane's 16 features; lbase =
// LDS base (+ the opaque zero); output blocks beyond the layer's width come out as zeros
template <int NB, bool RELU = true>
LP_DEV void loop_layer_fwd(const char* lbase, const float* sm, const LoopLayer& L, int lane, const float (&in)[NB][16],
                           float (&out)[NB][16]) {
  const int h = lane >> 5;
  const int in_chunks = (L.rows_in + 15) >> 4;
  f32x16 acc[NB];
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
    const float4* bsrc = reinterpret_cast<const float4*>(sm + L.bias + 32 * ob + 4 * h);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 v = bsrc[2 * j];
      acc[ob][4 * j + 0] = v.x; acc[ob][4 * j + 1] = v.y; acc[ob][4 * j + 2] = v.z; acc[ob][4 * j + 3] = v.w;
    }
  }
  if constexpr (NB == 1) {
    // (32-wide layers: the plain form measured faster -- 4/4/4 x 32: 8.90 vs 9.12 ms fwd+bwd -- the compiler's own interleaving
    // of the two chunks is better than the forced one)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < in_chunks) {  // wave-uniform
        u32x4_t l1, l2, l3;
        split3_chunk(&in[0][8 * c], l1, l2, l3);
        acc[0] = chunk_bf3(AColsFwd{lbase + L.img, LOOP_ST}, c, lane, l1, l2, l3, acc[0]);
      }
    }
  } else {
    // 64-wide layers, software pipeline over the K-chunks: the limbs of chunk c + 1 are split (44 VALU instructions) in the
    // shadow of chunk c's 12 dependent MFMAs -- these kernels run ONE wave per SIMD, there is no other wave to fill the matrix
    // pipe's latency (2/2/2 x 64: 13.97 -> 13.11 ms fwd+bwd).  The next chunk is split unconditionally inside chunk c's block, so
    // that block stays one scheduling region; a chunk beyond the layer's input width holds zeros and its block is skipped.
    u32x4_t l1, l2, l3;
    split3_chunk(&in[0][0], l1, l2, l3);
#pragma unroll
    for (int c = 0; c < 2 * NB; ++c) {
      if (c < in_chunks) {  // wave-uniform
        u32x4_t n1 = l1, n2 = l2, n3 = l3;
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) {
          if (ob == 0 || ob < L.ob)  // wave-uniform (a layer has at least one output block)
            acc[ob] = chunk_bf3(AColsFwd{lbase + L.img + ((c >> 1) * L.ob + ob) * LOOP_BLK, LOOP_ST}, c & 1, lane, l1, l2, l3, acc[ob]);
        }
        if (c + 1 < 2 * NB) {
          split3_chunk(&in[(c + 1) >> 1][8 * ((c + 1) & 1)], n1, n2, n3);
          loop_interleave_hint<NB>();
        }
        l1 = n1; l2 = n2; l3 = n3;
      }
    }
  }
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
#pragma unroll
    for (int q = 0; q < 16; ++q) out[ob][q] = RELU ? relu_f(acc[ob][q]) : acc[ob][q];
  }
}

// dX += W dY (blocks of the layer's INPUT): dy = NB blocks of this lane's 16 output features
// DXL: limbs of the gradient operand (lp_bf3.h; 2 only where the caller's chain is short, see renderer_bwd_loop)
template <int NB, int DXL = 3>
LP_DEV void loop_layer_dx(const char* lbase, const LoopLayer& L, int lane, const float (&dy)[NB][16], f32x16 (&dx)[NB]) {
  const int out_chunks = (L.cols + 15) >> 4;
  const int in_blocks = (L.rows_in + 31) >> 5;
  if constexpr (NB == 1) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c < out_chunks) {
        u32x4_t l1, l2, l3;
        dx_split_chunk<DXL>(&dy[0][8 * c], l1, l2, l3);
        dx[0] = dx_chunk<DXL>(ARowsBwd{lbase + L.img, LOOP_ST, 31}, c, lane, l1, l2, l3, dx[0]);
      }
    }
  } else {
    u32x4_t l1, l2, l3;
    dx_split_chunk<DXL>(&dy[0][0], l1, l2, l3);
#pragma unroll
    for (int c = 0; c < 2 * NB; ++c) {
      if (c < out_chunks) {  // wave-uniform; software-pipelined like loop_layer_fwd
        u32x4_t n1 = l1, n2 = l2, n3 = l3;
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
          if (ib == 0 || ib < in_blocks)
            dx[ib] = dx_chunk<DXL>(ARowsBwd{lbase + L.img + (ib * L.ob + (c >> 1)) * LOOP_BLK, LOOP_ST, 31}, c & 1, lane, l1, l2, l3, dx[ib]);
        }
        if (c + 1 < 2 * NB) {
          dx_split_chunk<DXL>(&dy[(c + 1) >> 1][8 * ((c + 1) & 1)], n1, n2, n3);
          loop_interleave_hint<NB, DXL == 2 ? 3 : 6>();
        }
        l1 = n1; l2 = n2; l3 = n3;
      }
    }
  }
}

LP_DEV void loop_tile_store(float* tile, int r, int h, const float (&v)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) tile[featq(q, h) * LT_LD + r] = v[q];
}

// dW quadrant of one 32 x 32 block over the 128 rays of the workgroup: acc += X^T dY (see lp_renderer_mfma_bwd.hip)
LP_DEV f32x4l loop_dw_quadrant(const float* wave0, int stride, int a_off, int b_off, f32x4l acc, float& db) {
  float s = 0.0f;
  for (int v = 0; v < WAVES; ++v) {
    const float* base = wave0 + v * stride;
    const float4 a0 = *reinterpret_cast<const float4*>(base + a_off);
    const float4 a1 = *reinterpret_cast<const float4*>(base + a_off + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(base + b_off);
    const float4 b1 = *reinterpret_cast<const float4*>(base + b_off + 4);
    acc = LP_MFMA16L(a0.x, b0.x, acc);
    acc = LP_MFMA16L(a0.y, b0.y, acc);
    acc = LP_MFMA16L(a0.z, b0.z, acc);
    acc = LP_MFMA16L(a0.w, b0.w, acc);
    acc = LP_MFMA16L(a1.x, b1.x, acc);
    acc = LP_MFMA16L(a1.y, b1.y, acc);
    acc = LP_MFMA16L(a1.z, b1.z, acc);
    acc = LP_MFMA16L(a1.w, b1.w, acc);
    s += ((b0.x + b0.y) + (b0.z + b0.w)) + ((b1.x + b1.y) + (b1.z + b1.w));
  }
  db += s;
  return acc;
}

// Weight gradients on the bf16 pipe (round 5, as in lp_renderer_mfma_bwd.hip: see the comment above dw_quadrant_bf there):
// the producer lanes publish X and dY of a block pair as two-limb bf16 tiles [ray][feature] (rm_off layout, ray k in row
// loop_rho(k), limb 2 at + rm_bytes(32)), the quadrant is three v_mfma_f32_16x16x32_bf16 per source wave; the bias gradient is
// the row 0 of a one-hot product.  -DLP_LOOP_DW_FP32 keeps the fp32 quadrants.
#ifndef LP_LOOP_DW_FP32
#define LP_LOOP_DW_BF16 1
#else
#define LP_LOOP_DW_BF16 0
#endif
typedef __bf16 bf16x8_l __attribute__((ext_vector_type(8)));
#define LP_MFMA16BL(a, b, c) \
  __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_l, (a)), __builtin_bit_cast(bf16x8_l, (b)), (c), 0, 0, 0)
LP_DEV constexpr int loop_rho(int k) { return (k & 0x15) | ((k & 2) << 2) | ((k & 8) >> 2); }
// eight rays (tile rows loop_rho(8 kq + j), j = 0..7) of one feature column: p = the supplier address of the first four rows
LP_DEV u32x4_t loop_limb_operand(const char* p) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_ptr;
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(p + (rm_off(4, 0) - rm_off(0, 0))));  // rays + 4
  const u32x2_t ua = __builtin_bit_cast(u32x2_t, a), ub = __builtin_bit_cast(u32x2_t, b);
  return (u32x4_t){ua.x, ua.y, ub.x, ub.y};
}
// x_off / y_off: byte offsets of this lane's supplier address inside a wave area (X / dY limb tile, limb 1); db: += the column sum of
// dY (valid in lanes 0..15: row 0 of the one-hot product; the other lanes add 0) when with_db
LP_DEV f32x4l loop_dw_quadrant_bf(const char* wave0b, int stride_bytes, int x_off, int y_off, f32x4l acc, float& db, bool with_db, int lane) {
  const unsigned one = (lane & 15) == 0 ? 0x3F803F80u : 0u;  // bf16 (1, 1) in row 0
  const u32x4_t oh = {one, one, one, one};
  f32x4l t = {0, 0, 0, 0};
#pragma unroll 1
  for (int v = 0; v < WAVES; ++v) {
    const char* base = wave0b + v * stride_bytes;
    const u32x4_t b2 = loop_limb_operand(base + y_off + rm_bytes(32));
    const u32x4_t a1 = loop_limb_operand(base + x_off);
    if (with_db) t = LP_MFMA16BL(oh, b2, t);  // (wave-uniform)
    acc = LP_MFMA16BL(a1, b2, acc);
    const u32x4_t b1 = loop_limb_operand(base + y_off);
    if (with_db) t = LP_MFMA16BL(oh, b1, t);
    acc = LP_MFMA16BL(a1, b1, acc);
    const u32x4_t a2 = loop_limb_operand(base + x_off + rm_bytes(32));
    acc = LP_MFMA16BL(a2, b1, acc);
  }
  db += t[0];
  return acc;
}

// accumulators of one layer's weight gradient: this wave's quadrant of every block + its share of the bias gradient
template <int NB>
struct LoopDw {
  f32x4l q[NB][NB];
  float db[NB];
};
template <int NB>
LP_DEV void loop_dw_zero(LoopDw<NB>& d) {
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    d.db[i] = 0.0f;
#pragma unroll
    for (int j = 0; j < NB; ++j) d.q[i][j] = (f32x4l){0, 0, 0, 0};
  }
}

// The two functions below have a body per dW mode (LP_LOOP_DW_FP32 is a per-translation-unit flag of build.py): an inline
// namespace per mode gives them distinct symbols, so the shallow (fp32 quadrants) and the deep (bf16 quadrants) translation units
// never hold two definitions of one name.  (The library is built with -fno-gpu-rdc: device code is per translation unit.)
#if LP_LOOP_DW_BF16
inline namespace dw_bf16 {
#else
inline namespace dw_fp32 {
#endif
// Backward of one layer: dW (workgroup-shared quadrants) and dX.  x = the layer's input activation, dy = upstream gradient
// (already masked by the layer's own ReLU).  Publishes the X / dY tiles of one block pair, runs the dX chain while the LDS
// writes land, then barrier -> quadrant -> barrier per block pair.
// z_delta (two-block layers, bf16 quadrants): byte distance from the dY tile to a THIRD tile of the wave's area, 0 = there is none.
// With it both dY blocks of a layer are published at once and an X block once per input block: a 64 x 64 layer stores 4 tiles behind 4
// barriers instead of 8 behind 8 (every (ib, ob) pair used to re-publish its X and dY block).
template <int NB, int DXL = 3>
LP_DEV void loop_layer_bwd(const char* lbase, const LoopLayer& L, int lane, float* xt, float* yt, const float* wave0, int stride,
                           int a_off, int b_off, bool want_params, bool want_dx, const float (&x)[NB][16], const float (&dy)[NB][16],
                           LoopDw<NB>& dw, f32x16 (&dx)[NB], int z_delta = 0) {
  const int h = lane >> 5, r = lane & 31;
  const int in_blocks = (L.rows_in + 31) >> 5;
#if LP_LOOP_DW_BF16
  // (a_off / b_off are BYTE offsets of the supplier addresses here, see the callers)
  char* const xrow = reinterpret_cast<char*>(xt) + rm_off(loop_rho(r), 4 * h);
  char* const yrow = reinterpret_cast<char*>(yt) + rm_off(loop_rho(r), 4 * h);
  if constexpr (NB == 2) {
    if (z_delta != 0) {  // (workgroup-uniform)
      const bool two_out = L.ob > 1;
      if (want_params) {
        limb_tile_store<2>(xrow, x[0]);
        limb_tile_store<2>(yrow, dy[0]);
        if (two_out) limb_tile_store<2>(yrow + z_delta, dy[1]);
      }
      if (want_dx) loop_layer_dx<NB, DXL>(lbase, L, lane, dy, dx);
      if (want_params) {
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
          if (ib < in_blocks) {  // workgroup-uniform
            if (ib > 0) limb_tile_store<2>(xrow, x[ib]);  // (behind the closing barrier of the previous input block)
            lds_barrier_l();
            float db_unused = 0.0f;
            dw.q[ib][0] = loop_dw_quadrant_bf(reinterpret_cast<const char*>(wave0), stride * 4, a_off, b_off, dw.q[ib][0],
                                              ib == 0 ? dw.db[0] : db_unused, ib == 0, lane);
            if (two_out)
              dw.q[ib][1] = loop_dw_quadrant_bf(reinterpret_cast<const char*>(wave0), stride * 4, a_off, b_off + z_delta, dw.q[ib][1],
                                                ib == 0 ? dw.db[1] : db_unused, ib == 0, lane);
            lds_barrier_l();
          }
        }
      }
      return;
    }
  }
  if (want_params) {
    limb_tile_store<2>(xrow, x[0]);
    limb_tile_store<2>(yrow, dy[0]);
  }
#else
  if (want_params) {
    loop_tile_store(xt, r, h, x[0]);
    loop_tile_store(yt, r, h, dy[0]);
  }
#endif
  if (want_dx) loop_layer_dx<NB, DXL>(lbase, L, lane, dy, dx);
  if (want_params) {
#pragma unroll
    for (int ib = 0; ib < NB; ++ib) {
#pragma unroll
      for (int ob = 0; ob < NB; ++ob) {
        if (ib < in_blocks && ob < L.ob) {  // workgroup-uniform
          if (ib + ob > 0) {
#if LP_LOOP_DW_BF16
            limb_tile_store<2>(xrow, x[ib]);
            limb_tile_store<2>(yrow, dy[ob]);
#else
            loop_tile_store(xt, r, h, x[ib]);
            loop_tile_store(yt, r, h, dy[ob]);
#endif
          }
          lds_barrier_l();
          float db_unused = 0.0f;
#if LP_LOOP_DW_BF16
          dw.q[ib][ob] = loop_dw_quadrant_bf(reinterpret_cast<const char*>(wave0), stride * 4, a_off, b_off, dw.q[ib][ob],
                                             ib == 0 ? dw.db[ob] : db_unused, ib == 0, lane);
#else
          dw.q[ib][ob] = loop_dw_quadrant(wave0, stride, a_off, b_off, dw.q[ib][ob], ib == 0 ? dw.db[ob] : db_unused);
#endif
          lds_barrier_l();
        }
      }
    }
  }
}

// flush this wave's quadrants of one layer
template <int NB>
LP_DEV void loop_dw_flush(float* G, const LoopLayer& L, const LoopDw<NB>& dw, int wave, int lane) {
  const int mi = wave >> 1, ni = wave & 1;
  const int m16 = lane & 15, ka = lane >> 4;
#pragma unroll
  for (int ib = 0; ib < NB; ++ib) {
#pragma unroll
    for (int ob = 0; ob < NB; ++ob) {
#if LP_LOOP_DW_BF16
      const int col = 32 * ob + 16 * ni + m16;  // 16x16x32 accumulator: column = lane & 15, rows 4 (lane >> 4) + i
#else
      const int col = 32 * ob + 16 * ni + pi16l(m16);
#endif
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#if LP_LOOP_DW_BF16
        const int row = 32 * ib + 16 * mi + 4 * ka + i;
#else
        const int row = 32 * ib + 16 * mi + pi16l(4 * ka + i);
#endif
        if (row < L.rows_in && col < L.cols) atomic_add_f32(G + L.w + (int64_t)row * L.ld + col, dw.q[ib][ob][i]);
      }
    }
  }
#pragma unroll
  for (int ob = 0; ob < NB; ++ob) {
    float d = dw.db[ob];
    d += __shfl_xor(d, 16);
    d += __shfl_xor(d, 32);
#if LP_LOOP_DW_BF16
    const int col = 32 * ob + 16 * ni + m16;
#else
    const int col = 32 * ob + 16 * ni + pi16l(m16);
#endif
    if (ka == 0 && mi == 0 && col < L.cols) atomic_add_f32(G + L.b + col, d);
  }
}

}  // inline namespace dw_bf16 / dw_fp32

template <int NB>
LP_DEV void loop_copy(const float (&src)[NB][16], float (&dst)[NB][16]) {
#pragma unroll
  for (int b = 0; b < NB; ++b) {
#pragma unroll
    for (int q = 0; q < 16; ++q) dst[b][q] = src[b][q];
  }
}

}  // namespace lp
