// lp_generic_mlp.h -- shape-generic MLP / grid-list device routines (one lane = one ray, private
// activation arrays; wide layers for the whole wave on the fp32 matrix cores, narrow ones per lane with
// wave-uniform weight loads) shared by the generic Renderer kernels (lp_renderer_generic.hip) and the
// MLP-Splatter kernels (lp_splatter_mlp.hip).
#pragma once
#include "lp_device.h"

namespace lp {

// y[o] = b[o] + sum_i x[i] * W[i*ldw + o], o < n_out.  x, y: private arrays.
// Blocks of eight outputs; every output is the chain b + x[0] w[0] + x[1] w[1] + ... in this order.  Full blocks run without a
// per-output test: the eight weight loads of a row are issued together and waited for once (round 6, late: the guarded form below --
// kept for the last, partial block -- compiled to eight load / s_waitcnt vmcnt(0) / v_fma triples behind eight scalar branches per
// row, i.e. eight serial memory round trips per eight FMAs: ~1 000 cycles per row of a 64-wide layer, 150x the MFMA families' time per
// MAC).  Same operations in the same order: results are bit-identical.
LP_DEV void dense(const float* __restrict__ W, const float* __restrict__ b, int d_in, int ldw,
                  int n_out, const float* x, float* y, bool relu) {
  int o0 = 0;
  for (; o0 + 8 <= n_out; o0 += 8) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = b[o0 + k];
    const float* w = W + o0;
#pragma unroll 8
    for (int i = 0; i < d_in; ++i) {  // (eight rows = 16 dwordx4 loads in flight: a wave of these kernels is alone on its SIMD, ILP is all it has)
      const float xi = x[i];
      float wv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) wv[k] = w[k];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(xi, wv[k], acc[k]);
      w += ldw;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) y[o0 + k] = relu ? fmaxf(acc[k], 0.0f) : acc[k];
  }
  if (o0 < n_out) {
    // the last, partial block (the heads' output layers: 1 opacity, 3 colour columns): the same batched loop; the slots beyond the
    // last output re-read that output's column (valid memory) and are not stored
    const int rem = n_out - o0;
    int kk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) kk[k] = k < rem ? k : rem - 1;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = b[o0 + kk[k]];
    const float* w = W + o0;
#pragma unroll 8
    for (int i = 0; i < d_in; ++i) {
      const float xi = x[i];
      float wv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) wv[k] = w[kk[k]];
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(xi, wv[k], acc[k]);
      w += ldw;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < rem) y[o0 + k] = relu ? fmaxf(acc[k], 0.0f) : acc[k];
  }
}

// dx[i] = sum_o dy[o] * W[i*ldw + o]  (o < n_out): every dx[i] is the chain dy[0] w[0] + dy[1] w[1] + ... in this order.  Four rows at
// a time over four outputs at a time: 4 + 16 loads in flight before 16 FMAs on four independent chains (the plain double loop was one
// load pair and one dependent FMA per round trip); the per-row order is unchanged -- bit-identical results.
LP_DEV void dense_bwd_input(const float* __restrict__ W, int d_in, int ldw, int n_out,
                            const float* dy, float* dx) {
  int i = 0;
  if (n_out < 4) {
    // the heads' output layers (1 opacity, 3 colour columns): eight rows of up to three weights in flight (the general loop below
    // takes a round trip per output here); every dx[i] is still dy[0] w[0] + dy[1] w[1] + dy[2] w[2] in this order
    float d[3];
    int kk[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      kk[k] = k < n_out ? k : n_out - 1;
      d[k] = dy[kk[k]];
    }
    for (; i + 8 <= d_in; i += 8) {
      float w[8][3];
#pragma unroll
      for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) w[r][k] = W[(int64_t)(i + r) * ldw + kk[k]];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float sum = 0.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k)
          if (k < n_out) sum = fmaf(d[k], w[r][k], sum);
        dx[i + r] = sum;
      }
    }
  }
  for (; i + 4 <= d_in; i += 4) {
    const float* w0 = W + (int64_t)i * ldw;
    const float* w1 = w0 + ldw;
    const float* w2 = w1 + ldw;
    const float* w3 = w2 + ldw;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int o = 0;
#pragma unroll 4
    for (; o + 4 <= n_out; o += 4) {
      float d[4], a0[4], a1[4], a2[4], a3[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        d[k] = dy[o + k];
        a0[k] = w0[o + k];
        a1[k] = w1[o + k];
        a2[k] = w2[o + k];
        a3[k] = w3[o + k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s0 = fmaf(d[k], a0[k], s0);
        s1 = fmaf(d[k], a1[k], s1);
        s2 = fmaf(d[k], a2[k], s2);
        s3 = fmaf(d[k], a3[k], s3);
      }
    }
    for (; o < n_out; ++o) {
      const float d = dy[o];
      s0 = fmaf(d, w0[o], s0);
      s1 = fmaf(d, w1[o], s1);
      s2 = fmaf(d, w2[o], s2);
      s3 = fmaf(d, w3[o], s3);
    }
    dx[i] = s0;
    dx[i + 1] = s1;
    dx[i + 2] = s2;
    dx[i + 3] = s3;
  }
  for (; i < d_in; ++i) {
    const float* w = W + (int64_t)i * ldw;
    float s = 0.0f;
    for (int o = 0; o < n_out; ++o) s = fmaf(dy[o], w[o], s);
    dx[i] = s;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The same two products for the WHOLE WAVE on the fp32 matrix cores (round 6, last): Y^T = W^T X^T as v_mfma_f32_32x32x2_f32 with
// M = 32 output features, N = 32 rays, K = two input features per instruction -- fp32 products, fp32 accumulation, the chain of an
// output still b + x[0] w[0] + x[1] w[1] + ... in ascending order.  The rays' inputs go through one LDS tile Xs[64][ld] (lane = ray
// writes its row; MFMA lane (n, k) reads ray n's and ray 32 + n's input i + k: conflict-free with ld odd), the weights come straight
// from global memory (lane (m, k) reads W[i + k][o0 + m]: two 128-byte rows per instruction instead of one 16-byte broadcast per four
// FMAs), the accumulators return to the lane = ray layout with one cross-half shuffle per register.  A 64 x 64 layer: 128 matrix
// instructions (~8 k cycles) against 4 096 FMAs + 1 024 broadcast loads per lane (~30 k measured).  Wave-uniform control flow only.
typedef float lp_acc16_t __attribute__((ext_vector_type(16)));

// one lane = one ray writes its row of an LDS staging tile [64][ld]
LP_DEV void stage(float* s, int ld, int lane, const float* v, int n, bool zero) {
  float* row = s + lane * ld;
  int i = 0;
  for (; i + 8 <= n; i += 8) {  // eight private-array reads in flight (one read / wait / write per element before)
    float t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = v[i + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) row[i + u] = zero ? 0.0f : t[u];
  }
  for (; i < n; ++i) row[i] = zero ? 0.0f : v[i];
}


// After two MFMAs lane (n, h) holds rows (j & 3) + 8 (j >> 2) + 4 h of column n for rays n (acc0) and 32 + n (acc1).  One exchange
// with lane ^ 32: every lane then holds its OWN ray -- acc0[j] = row (j & 3) + 8 (j >> 2), acc1[j] = that row + 4.
LP_DEV void mfma_rows_to_rays(lp_acc16_t& acc0, lp_acc16_t& acc1, int h) {
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const float send = h ? acc0[j] : acc1[j];
    const float recv = __shfl_xor(send, 32, 64);
    if (h) acc0[j] = recv; else acc1[j] = recv;
  }
}

LP_DEV bool dense_on_mfma(int d_in, int n_out) { return n_out >= 24 && d_in >= 8; }

LP_DEV void dense_wave(const float* __restrict__ W, const float* __restrict__ b, int d_in, int ldw, int n_out, const float* x,
                       float* y, bool relu, float* Xs, int ld, int lane) {
  stage(Xs, ld, lane, x, d_in, false);
  __syncthreads();
  const int m = lane & 31, k = lane >> 5;
  const float* x0 = Xs + m * ld;
  const float* x1 = Xs + (32 + m) * ld;
  for (int o0 = 0; o0 < n_out; o0 += 32) {
    const float* wp = W + ((o0 + m < n_out) ? o0 + m : n_out - 1);
    lp_acc16_t acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = o0 + (j & 3) + 8 * (j >> 2) + 4 * k;
      const float bv = b[o < n_out ? o : n_out - 1];
      acc0[j] = bv;
      acc1[j] = bv;
    }
    // batches of eight instruction pairs (16 inputs), the operands of batch n + 1 in flight under the matrix instructions of batch n
    // (the plain loop compiled to load / s_waitcnt vmcnt(0) / 2 MFMAs: one memory round trip per pair)
    int i2 = 0;
    if (d_in >= 16) {
      float aw[8], b0[8], b1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = 2 * u + k;
        aw[u] = wp[(int64_t)i * ldw];
        b0[u] = x0[i];
        b1[u] = x1[i];
      }
      for (; i2 + 16 <= d_in; i2 += 16) {
        float awn[8], b0n[8], b1n[8];
        const bool more = i2 + 32 <= d_in;  // wave-uniform
        if (more) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i2 + 16 + 2 * u + k;
            awn[u] = wp[(int64_t)i * ldw];
            b0n[u] = x0[i];
            b1n[u] = x1[i];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[u], b0[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[u], b1[u], acc1, 0, 0, 0);
        }
        if (more) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            aw[u] = awn[u];
            b0[u] = b0n[u];
            b1[u] = b1n[u];
          }
        }
      }
    }
    for (; i2 + 2 <= d_in; i2 += 2) {
      const int i = i2 + k;
      const float aw1 = wp[(int64_t)i * ldw];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw1, x0[i], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw1, x1[i], acc1, 0, 0, 0);
    }
    if (i2 < d_in) {  // odd width: the k = 1 half of the last instruction carries a zero weight
      const float aw = k == 0 ? wp[(int64_t)i2 * ldw] : 0.0f;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, x0[i2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, x1[i2], acc1, 0, 0, 0);
    }
    mfma_rows_to_rays(acc0, acc1, k);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int o = o0 + (j & 3) + 8 * (j >> 2);
      if (o < n_out) y[o] = relu ? fmaxf(acc0[j], 0.0f) : acc0[j];
      if (o + 4 < n_out) y[o + 4] = relu ? fmaxf(acc1[j], 0.0f) : acc1[j];
    }
  }
  __syncthreads();
}

// dx[ray][i] = sum_o dy[ray][o] W[i][o] for the whole wave: M = 32 inputs i, N = 32 rays, K = two outputs o per instruction; dy from
// its staging tile Ys[64][ld] (zeros for lanes without a contribution), lane (m, k) reads W[i0 + m][o + k].
LP_DEV bool dense_bwd_on_mfma(int d_in, int n_out) { return d_in >= 24 && n_out >= 8; }

LP_DEV void dense_bwd_input_wave(const float* __restrict__ W, int d_in, int ldw, int n_out, const float* Ys, int ld, float* dx,
                                 int lane) {
  const int m = lane & 31, k = lane >> 5;
  const float* y0 = Ys + m * ld;
  const float* y1 = Ys + (32 + m) * ld;
  for (int i0 = 0; i0 < d_in; i0 += 32) {
    const float* wp = W + (int64_t)((i0 + m < d_in) ? i0 + m : d_in - 1) * ldw;
    lp_acc16_t acc0, acc1;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      acc0[j] = 0.0f;
      acc1[j] = 0.0f;
    }
    int o2 = 0;
    if (n_out >= 16) {  // (batched as in dense_wave)
      float aw[8], b0[8], b1[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int o = 2 * u + k;
        aw[u] = wp[o];
        b0[u] = y0[o];
        b1[u] = y1[o];
      }
      for (; o2 + 16 <= n_out; o2 += 16) {
        float awn[8], b0n[8], b1n[8];
        const bool more = o2 + 32 <= n_out;
        if (more) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int o = o2 + 16 + 2 * u + k;
            awn[u] = wp[o];
            b0n[u] = y0[o];
            b1n[u] = y1[o];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[u], b0[u], acc0, 0, 0, 0);
          acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[u], b1[u], acc1, 0, 0, 0);
        }
        if (more) {
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            aw[u] = awn[u];
            b0[u] = b0n[u];
            b1[u] = b1n[u];
          }
        }
      }
    }
    for (; o2 + 2 <= n_out; o2 += 2) {
      const int o = o2 + k;
      const float aw1 = wp[o];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw1, y0[o], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw1, y1[o], acc1, 0, 0, 0);
    }
    if (o2 < n_out) {
      const float aw = k == 0 ? wp[o2] : 0.0f;
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, y0[o2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, y1[o2], acc1, 0, 0, 0);
    }
    mfma_rows_to_rays(acc0, acc1, k);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int i = i0 + (j & 3) + 8 * (j >> 2);
      if (i < d_in) dx[i] = acc0[j];
      if (i + 4 < d_in) dx[i + 4] = acc1[j];
    }
  }
  __syncthreads();
}

LP_DEV const float* mlp_w(const float* params, const LpMlp& m, int layer) {
  int64_t off = m.offset;
  for (int l = 0; l < layer; ++l) off += (int64_t)m.dims[l] * m.dims[l + 1];
  return params + off;
}
LP_DEV const float* mlp_b(const float* params, const LpMlp& m, int layer) {
  int64_t off = m.offset;
  for (int l = 0; l < m.n_layers; ++l) off += (int64_t)m.dims[l] * m.dims[l + 1];
  for (int l = 0; l < layer; ++l) off += m.dims[l + 1];
  return params + off;
}

// Sample every grid of the list at (x,y,z) and sum into out[C].
LP_DEV void sample_list(const LpGridList& gl, int b, float x, float y, float z, bool mask_oob,
                        float* out) {
  const int C = gl.channels;
  for (int c = 0; c < C; ++c) out[c] = 0.0f;
  if (mask_oob && !point_in_bounds(x, y, z)) return;
  for (int g = 0; g < gl.n_grids; ++g) {
    const Corners cs = grid_corners<false>(gl.grids[g], b, x, y, z);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < cs.n && cs.row[k] >= 0) {
        const float w = cs.w[k];
        const float* src = gl.grids[g].data + cs.row[k] * C;
        if ((C & 3) == 0) {
          int c = 0;
          for (; c + 16 <= C; c += 16) {  // four row loads + 16 reads of the private sums in flight (was: a round trip per four channels)
            float4 v[4];
            float o16[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const float4*>(src + c + 4 * q);
#pragma unroll
            for (int q = 0; q < 16; ++q) o16[q] = out[c + q];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              out[c + 4 * q + 0] = fmaf(w, v[q].x, o16[4 * q + 0]);
              out[c + 4 * q + 1] = fmaf(w, v[q].y, o16[4 * q + 1]);
              out[c + 4 * q + 2] = fmaf(w, v[q].z, o16[4 * q + 2]);
              out[c + 4 * q + 3] = fmaf(w, v[q].w, o16[4 * q + 3]);
            }
          }
          for (; c < C; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + c);
            out[c + 0] = fmaf(w, v.x, out[c + 0]);
            out[c + 1] = fmaf(w, v.y, out[c + 1]);
            out[c + 2] = fmaf(w, v.z, out[c + 2]);
            out[c + 3] = fmaf(w, v.w, out[c + 3]);
          }
        } else {
          for (int c = 0; c < C; ++c) out[c] = fmaf(w, src[c], out[c]);
        }
      }
    }
  }
}

// grad[g]: gradient buffer of grid g (same row indexing as gl.grids[g].data)
LP_DEV void splat_list(const LpGridList& gl, float* const* grad, int b, float x, float y, float z,
                       bool mask_oob, const float* d) {
  const int C = gl.channels;
  if (mask_oob && !point_in_bounds(x, y, z)) return;
  for (int g = 0; g < gl.n_grids; ++g) {
    const Corners cs = grid_corners<false>(gl.grids[g], b, x, y, z);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < cs.n && cs.row[k] >= 0) {
        const float w = cs.w[k];
        float* dst = grad[g] + cs.row[k] * C;
        int c = 0;
        for (; c + 16 <= C; c += 16) {  // 16 reads of the private gradient in flight, then 16 atomics (was: a round trip per atomic)
          float d16[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) d16[q] = d[c + q];
#pragma unroll
          for (int q = 0; q < 16; ++q) atomic_add_f32(dst + c + q, w * d16[q]);
        }
        for (; c < C; ++c) atomic_add_f32(dst + c, w * d[c]);
      }
    }
  }
}

// The same scatter for the whole wave with lanes = CHANNELS: in the lane = ray form above one atomic instruction carries ONE channel of
// 64 different rows -- 64 four-byte memory transactions --, i.e. C transactions per (ray, corner) where the row is C * 4 contiguous bytes:
// 805 M transactions for 16 384 rays x 128 samples of a 32-channel triplane, 38 ms at the chip's ~21 G atomic segments per second and the
// whole of the large-batch backward (7.2 G for 147 456 rays).  Here the rays' gradients go through the LDS tile Xs[64][ld], their corner
// rows and weights through Ys (three words per corner), and an instruction writes 64 / CW whole corner rows (CW = 16 / 32 / 64 lanes
// per row): C * 4 / 64 segments per (ray, corner).  Needs ld >= 24 and wave-uniform control flow; `live`: the lane contributes.
LP_DEV bool splat_wave_ok(int ld) { return ld >= 24; }

LP_DEV void splat_list_wave(const LpGridList& gl, float* const* grad, int b, float x, float y, float z, bool mask_oob, const float* d,
                            bool live, float* Xs, float* Ys, int ld, int lane) {
  const int C = gl.channels;
  const bool on = live && !(mask_oob && !point_in_bounds(x, y, z));
  stage(Xs, ld, lane, d, C, false);
  const int cw_log = C <= 16 ? 4 : (C <= 32 ? 5 : 6);
  const int per = 64 >> cw_log;  // corner rows per instruction
  const int sub = lane >> cw_log, c = lane & ((1 << cw_log) - 1);
  float* yr = Ys + lane * ld;
  for (int g = 0; g < gl.n_grids; ++g) {
    const Corners cs = grid_corners<false>(gl.grids[g], b, x, y, z);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const bool ok = on && k < cs.n && cs.row[k] >= 0;
      const int64_t row = ok ? cs.row[k] : (int64_t)-1;
      yr[3 * k + 0] = __int_as_float((int)(row & 0xffffffff));
      yr[3 * k + 1] = __int_as_float((int)(row >> 32));
      yr[3 * k + 2] = ok ? cs.w[k] : 0.0f;
    }
    __syncthreads();
    const int nk_log = __builtin_amdgcn_readfirstlane(cs.n) == 8 ? 3 : 2;  // (8 corners of a voxel grid, 4 of a plane: a property of the grid)
    const int items = 64 << nk_log;
    float* gbase = grad[g];
    for (int t0 = 0; t0 < items; t0 += 4 * per) {  // four instructions' operands in flight
      int64_t rows[4];
      float wv[4], dv[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int t = t0 + q * per + sub;
        const int r = t >> nk_log, k = t & ((1 << nk_log) - 1);
        const float* e = Ys + r * ld + 3 * k;
        rows[q] = ((int64_t)__float_as_int(e[1]) << 32) | (int64_t)(unsigned)__float_as_int(e[0]);
        wv[q] = e[2];
        dv[q] = c < C ? Xs[r * ld + c] : 0.0f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (rows[q] >= 0 && c < C) atomic_add_f32(gbase + rows[q] * C + c, wv[q] * dv[q]);
      if (C > 64) {  // (the second 64 channels of a 128-channel grid)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int t = t0 + q * per + sub;
          const int r = t >> nk_log;
          if (rows[q] >= 0 && 64 + c < C) atomic_add_f32(gbase + rows[q] * C + 64 + c, wv[q] * Xs[r * ld + 64 + c]);
        }
      }
    }
    __syncthreads();
  }
}

// Wave-level reduction of dW += X^T dY and db += sum dY over the 64 rays of the block.
// Xs/Ys: LDS staging [64][ld] (stage(), above).  gW/gb: accumulation targets (LDS or global).
template <bool LDS_ACC>
LP_DEV void accum(float* target, float v) {
  if (LDS_ACC)
    *target += v;  // each (i,o) entry is owned by exactly one lane
  else
#ifdef LP_TIMING_GEN_NO_DW_ATOMICS  // timing experiment (wrong gradients): what the weight-gradient atomics to global memory cost
    { if (v == 12345.678f) atomic_add_f32(target, v); }
#else
    atomic_add_f32(target, v);
#endif
}

// dW += X^T dY over the wave's 64 rays on the fp32 matrix cores: one v_mfma_f32_32x32x2_f32 per pair of rays and 32 x 32 block of
// (i, o) -- fp32 products, fp32 accumulation, rays in ascending order --, operands straight from the staging tiles (lane (m, k) reads
// X[2t + k][i0 + m] and dY[2t + k][o0 + m]: one ds_read_b32 each, conflict-free rows).  A 64 x 64 layer is 128 matrix instructions
// (~8 k cycles) where the per-entry chains of rounds 1-6 took 4 096 FMAs per lane behind an LDS round trip each (~170 of the ~200 ms of
// a 3/2/2 x 64 backward on 16 384 rays; batching the reads: 139 ms; this form: profiles/r06_generic_kernels.txt).  Rows / columns beyond
// the layer re-read its last row / column and are not stored.  Accumulator layout: lane (n, h) holds column o0 + n, rows i0 + (j & 3)
// + 8 (j >> 2) + 4 h of register j.
template <bool LDS_ACC>
LP_DEV void wave_outer(const float* Xs, const float* Ys, int ld, int d_in, int ldw, int n_out,
                       float* gW, float* gb, int lane) {
  typedef float acc16_t __attribute__((ext_vector_type(16)));
  __syncthreads();
  const int m = lane & 31, h = lane >> 5;
  for (int i0 = 0; i0 < d_in; i0 += 32) {
    const float* xa = Xs + h * ld + ((i0 + m < d_in) ? i0 + m : d_in - 1);
    for (int o0 = 0; o0 < n_out; o0 += 32) {
      const float* yb = Ys + h * ld + ((o0 + m < n_out) ? o0 + m : n_out - 1);
      acc16_t acc;
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
#pragma unroll 8
      for (int t = 0; t < 32; ++t) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * t * ld], yb[2 * t * ld], acc, 0, 0, 0);
      const int o = o0 + m;
      if (o < n_out) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const int i = i0 + (j & 3) + 8 * (j >> 2) + 4 * h;
          if (i < d_in) accum<LDS_ACC>(gW + (int64_t)i * ldw + o, acc[j]);
        }
      }
    }
  }
  for (int o = lane; o < n_out; o += 64) {
    float s = 0.0f;
    for (int r = 0; r < 64; r += 8) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = Ys[(r + q) * ld + o];
#pragma unroll
      for (int q = 0; q < 8; ++q) s += v[q];
    }
    accum<LDS_ACC>(gb + o, s);
  }
  __syncthreads();
}

// Backward through one MLP (layers last..first).  On entry dy[] holds the gradient w.r.t.
// the MLP's (raw) output; on exit dx[] holds the gradient w.r.t. its input (pre input-ReLU).
// in_slot / out_slots index act[].  Uses tmp as ping-pong.  `live`: lane contributes.
template <bool LDS_ACC>
LP_DEV void mlp_backward(const float* params, int ld, const LpMlp& m, int n_out_last, int in_slot,
                         const int* out_slots, const float* act, float* dy, float* dx,
                         float* gparams, float* Xs, float* Ys, int lane, bool live) {
  for (int l = m.n_layers - 1; l >= 0; --l) {
    const int d_in = m.dims[l], ldw = m.dims[l + 1];
    const int n_out = (l == m.n_layers - 1) ? n_out_last : ldw;
    const float* x = act + (l == 0 ? in_slot : out_slots[l - 1]);
    // dy currently w.r.t. this layer's output post-activation; hidden layers: apply ReLU mask
    if (l != m.n_layers - 1) {
      const float* yv = act + out_slots[l];
      int o = 0;
      for (; o + 8 <= n_out; o += 8) {  // (eight reads of each private array in flight)
        float a8[8], d8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          a8[u] = yv[o + u];
          d8[u] = dy[o + u];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) dy[o + u] = (a8[u] > 0.0f) ? d8[u] : 0.0f;
      }
      for (; o < n_out; ++o) dy[o] = (yv[o] > 0.0f) ? dy[o] : 0.0f;
    }
    const bool mm = dense_bwd_on_mfma(d_in, n_out);  // wave-uniform
    if (gparams || mm) stage(Ys, ld, lane, dy, n_out, !live);
    if (gparams) {
      stage(Xs, ld, lane, x, d_in, !live);
      const int64_t w_off = mlp_w(params, m, l) - params;
      const int64_t b_off = mlp_b(params, m, l) - params;
      wave_outer<LDS_ACC>(Xs, Ys, ld, d_in, ldw, n_out, gparams + w_off, gparams + b_off, lane);
    } else if (mm) {
      __syncthreads();
    }
    // the input gradient becomes the next (earlier) layer's output gradient
    if (mm) {
      // (dY is read from its tile, staged above: the private dy[] is free and takes the result directly; the first layer's goes to dx[])
      dense_bwd_input_wave(mlp_w(params, m, l), d_in, ldw, n_out, Ys, ld, l == 0 ? dx : dy, lane);
    } else {
      dense_bwd_input(mlp_w(params, m, l), d_in, ldw, n_out, dy, dx);
      if (l > 0)
        for (int i = 0; i < d_in; ++i) dy[i] = dx[i];
    }
  }
}

}  // namespace lp
