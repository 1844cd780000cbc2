// lp_generic_mlp.h -- shape-generic MLP / grid-list device routines (one lane = one ray, private
// activation arrays, wave-uniform weight loads) shared by the generic Renderer kernels
// (lp_renderer_generic.hip) and the MLP-Splatter kernels (lp_splatter_mlp.hip).
#pragma once
#include "lp_device.h"

namespace lp {

// y[o] = b[o] + sum_i x[i] * W[i*ldw + o], o < n_out.  x, y: private arrays.
LP_DEV void dense(const float* __restrict__ W, const float* __restrict__ b, int d_in, int ldw,
                  int n_out, const float* x, float* y, bool relu) {
  for (int o0 = 0; o0 < n_out; o0 += 8) {
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = (o0 + k < n_out) ? b[o0 + k] : 0.0f;
    for (int i = 0; i < d_in; ++i) {
      const float xi = x[i];
      const float* w = W + (int64_t)i * ldw + o0;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (o0 + k < n_out) acc[k] = fmaf(xi, w[k], acc[k]);
    }
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (o0 + k < n_out) y[o0 + k] = relu ? fmaxf(acc[k], 0.0f) : acc[k];
  }
}

// dx[i] = sum_o dy[o] * W[i*ldw + o]  (o < n_out)
LP_DEV void dense_bwd_input(const float* __restrict__ W, int d_in, int ldw, int n_out,
                            const float* dy, float* dx) {
  for (int i = 0; i < d_in; ++i) {
    const float* w = W + (int64_t)i * ldw;
    float s = 0.0f;
    for (int o = 0; o < n_out; ++o) s = fmaf(dy[o], w[o], s);
    dx[i] = s;
  }
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
          for (int c = 0; c < C; c += 4) {
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
        for (int c = 0; c < C; ++c) atomic_add_f32(dst + c, w * d[c]);
      }
    }
  }
}

// Wave-level reduction of dW += X^T dY and db += sum dY over the 64 rays of the block.
// Xs/Ys: LDS staging [64][ld].  gW/gb: accumulation targets (LDS or global).
LP_DEV void stage(float* s, int ld, int lane, const float* v, int n, bool zero) {
  for (int i = 0; i < n; ++i) s[lane * ld + i] = zero ? 0.0f : v[i];
}

template <bool LDS_ACC>
LP_DEV void accum(float* target, float v) {
  if (LDS_ACC)
    *target += v;  // each (i,o) entry is owned by exactly one lane
  else
    atomic_add_f32(target, v);
}

template <bool LDS_ACC>
LP_DEV void wave_outer(const float* Xs, const float* Ys, int ld, int d_in, int ldw, int n_out,
                       float* gW, float* gb, int lane) {
  __syncthreads();
  const int n = d_in * n_out;
  for (int e = lane; e < n; e += 64) {
    const int i = e / n_out, o = e - i * n_out;
    float s = 0.0f;
    for (int r = 0; r < 64; ++r) s = fmaf(Xs[r * ld + i], Ys[r * ld + o], s);
    accum<LDS_ACC>(gW + (int64_t)i * ldw + o, s);
  }
  for (int o = lane; o < n_out; o += 64) {
    float s = 0.0f;
    for (int r = 0; r < 64; ++r) s += Ys[r * ld + o];
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
      for (int o = 0; o < n_out; ++o) dy[o] = (yv[o] > 0.0f) ? dy[o] : 0.0f;
    }
    if (gparams) {
      stage(Xs, ld, lane, x, d_in, !live);
      stage(Ys, ld, lane, dy, n_out, !live);
      const int64_t w_off = mlp_w(params, m, l) - params;
      const int64_t b_off = mlp_b(params, m, l) - params;
      wave_outer<LDS_ACC>(Xs, Ys, ld, d_in, ldw, n_out, gparams + w_off, gparams + b_off, lane);
    }
    dense_bwd_input(mlp_w(params, m, l), d_in, ldw, n_out, dy, dx);
    // the input gradient becomes the next (earlier) layer's output gradient
    for (int i = 0; i < d_in; ++i) dy[i] = dx[i];
  }
}

}  // namespace lp
