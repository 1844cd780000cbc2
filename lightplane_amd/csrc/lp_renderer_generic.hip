// lp_renderer_generic.hip -- shape-generic Renderer kernels (any layer count / width).
//
// One lane = one ray, one wave (64 lanes) = one workgroup.  Activations of the current
// sample live in a per-lane private array; MLP weights are read through wave-uniform
// (scalar) loads.  The backward kernel recomputes the forward per sample (far -> near),
// reconstructs the transmittance from the saved final -log T and reduces the weight
// gradients across the 64 rays of the wave with an LDS-staged outer product.
//
// This is the correctness anchor and the fallback for shapes the MFMA kernels
// (lp_renderer_mfma.hip) are not specialised for.  Replaces the reference's Triton
// fw_kernel / bw_kernel (templates/renderer_fw.py:85-375, renderer_bw.py:89-627) for
// arbitrary (n_layers, width) without code generation.
#include "lp_generic_mlp.h"
#include "lp_host.h"

// Waves per SIMD the kernels are compiled for (HIP: the second argument of __launch_bounds__ is the minimum number of waves per
// execution unit; 0 here = no bound: the forward takes 134 registers = three waves per SIMD, the backward all 512 = one).  A bound on
// the backward costs more in spills than its extra waves return, for small AND large batches -- 3/2/2 x 64, fwd+bwd ms at 16 384 / 147 456
// rays: no bound 155 / 672 | 2 waves: 269 / 1 369 | 3: 340 / 1 751 | 4: 350 / 1 896 | 8: 382 / 1 979 (profiles/r06_generic_kernels.txt).
#ifndef LP_GEN_BWD_OCC
#define LP_GEN_BWD_OCC 0
#endif
#ifndef LP_GEN_FWD_OCC
#define LP_GEN_FWD_OCC 0
#endif
#if LP_GEN_BWD_OCC > 0
#define LP_GEN_BWD_BOUNDS __launch_bounds__(64, LP_GEN_BWD_OCC)
#else
#define LP_GEN_BWD_BOUNDS __launch_bounds__(64)
#endif
#if LP_GEN_FWD_OCC > 0
#define LP_GEN_FWD_BOUNDS __launch_bounds__(64, LP_GEN_FWD_OCC)
#else
#define LP_GEN_FWD_BOUNDS __launch_bounds__(64)
#endif

namespace lp {

// Offsets (in floats) of every activation of one sample inside the private array.
struct GenPlan {
  int x0;                   // [C] summed grid sample (raw)
  int cx0;                  // [C] colour-grid sample (raw) or -1
  int trunk[LP_MAX_LAYERS]; // trunk layer outputs (post ReLU)
  int op_in;                // opacity head input (post ReLU)
  int col_in;               // colour head input (+ encoding)
  int op[LP_MAX_LAYERS];    // opacity layer outputs (hidden: post ReLU, last: raw)
  int col[LP_MAX_LAYERS];   // colour layer outputs (hidden: post ReLU, last: raw, color_chn used)
  int total;
  int head_w;               // width of the head inputs
};

struct GenArgs {
  LpRendererArgs a;
  GenPlan p;
  int stage_ld;        // LDS staging row stride (floats), bwd only
  int lds_param_accum; // 1: accumulate weight grads in LDS, flush once per block
  // test hook (lp_renderer_backward_relu_dump, DUMP twin only): [ray][sample][dump_words] words -- dump_wps words per ReLU site in the
  // reference's evaluation order, then the visited flag
  uint32_t* relu_dump;
  int dump_words, dump_wps;
};

// Full decoder of one sample.  Fills act[] per plan; returns the raw opacity (pre noise).
// The raw colours are left in act[p.col[nC-1] .. +color_chn).
// Layers of 24 and more outputs run for the whole wave on the fp32 matrix cores (dense_wave, lp_generic_mlp.h; Xs: the wave's LDS tile
// [64][ga.stage_ld]); the narrow ones (the heads' output layers) stay per lane.  Wave-uniform control flow.
LP_DEV void dense_any(const float* W, const float* b, int d_in, int ldw, int n_out, const float* x, float* y, bool relu, float* Xs,
                      int ld, int lane) {
  if (dense_on_mfma(d_in, n_out))
    dense_wave(W, b, d_in, ldw, n_out, x, y, relu, Xs, ld, lane);
  else
    dense(W, b, d_in, ldw, n_out, x, y, relu);
}

LP_DEV float decode(const GenArgs& ga, const Ray& ray, float x, float y, float z,
                    const float* enc, float* act, float* Xs, int lane) {
  const LpRendererArgs& a = ga.a;
  const GenPlan& p = ga.p;
  const bool mask = a.march.mask_out_of_bounds != 0;
  const bool two_grids = a.color_grid.n_grids > 0;
  const int C = a.grid.channels;
  sample_list(a.grid, ray.b, x, y, z, mask, act + p.x0);
  if (two_grids) {
    sample_list(a.color_grid, ray.b, x, y, z, mask, act + p.cx0);
    for (int c = 0; c < C; ++c) {
      act[p.op_in + c] = fmaxf(act[p.x0 + c], 0.0f);
      act[p.col_in + c] = fmaxf(act[p.cx0 + c], 0.0f) + enc[c];
    }
  } else {
    const float* cur = act + p.x0;
    int w = C;
    for (int l = 0; l < a.trunk.n_layers; ++l) {
      dense_any(mlp_w(a.mlp_params, a.trunk, l), mlp_b(a.mlp_params, a.trunk, l), a.trunk.dims[l],
                a.trunk.dims[l + 1], a.trunk.dims[l + 1], cur, act + p.trunk[l], true, Xs, ga.stage_ld, lane);
      cur = act + p.trunk[l];
      w = a.trunk.dims[l + 1];
    }
    if (a.trunk.n_layers == 0) {
      for (int c = 0; c < C; ++c) act[p.op_in + c] = fmaxf(act[p.x0 + c], 0.0f);
      cur = act + p.op_in;
    }
    {
      int c = 0;
      for (; c + 8 <= w; c += 8) {  // (eight reads of each private array in flight)
        float u8[8], e8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          u8[q] = cur[c + q];
          e8[q] = enc[c + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) act[p.col_in + c + q] = u8[q] + e8[q];
      }
      for (; c < w; ++c) act[p.col_in + c] = cur[c] + enc[c];
    }
  }
  // opacity head
  {
    const float* cur = act + p.op_in;
    const LpMlp& m = a.opacity;
    for (int l = 0; l < m.n_layers; ++l) {
      const bool last = (l == m.n_layers - 1);
      dense_any(mlp_w(a.mlp_params, m, l), mlp_b(a.mlp_params, m, l), m.dims[l], m.dims[l + 1],
                last ? 1 : m.dims[l + 1], cur, act + p.op[l], !last, Xs, ga.stage_ld, lane);
      cur = act + p.op[l];
    }
  }
  // colour head
  {
    const float* cur = act + p.col_in;
    const LpMlp& m = a.color;
    for (int l = 0; l < m.n_layers; ++l) {
      const bool last = (l == m.n_layers - 1);
      dense_any(mlp_w(a.mlp_params, m, l), mlp_b(a.mlp_params, m, l), m.dims[l], m.dims[l + 1],
                last ? a.color_chn : m.dims[l + 1], cur, act + p.col[l], !last, Xs, ga.stage_ld, lane);
      cur = act + p.col[l];
    }
  }
  return act[p.op[a.opacity.n_layers - 1]];
}

template <int ACT_CAP>
__global__ void LP_GEN_FWD_BOUNDS renderer_fwd_generic(const GenArgs ga) {
  extern __shared__ __attribute__((aligned(16))) float lds[];  // the wave's input tile [64][ga.stage_ld] (dense_wave)
  const LpRendererArgs& a = ga.a;
  const GenPlan& p = ga.p;
  const int64_t ray_id = (int64_t)blockIdx.x * 64 + threadIdx.x;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;
  float act[ACT_CAP];
  float enc[LP_MAX_WIDTH];
  float facc[LP_MAX_WIDTH];
  const Ray ray = load_ray(a.rays, rid);
  const int E = a.rays.encoding_dim;
  for (int c = 0; c < E; ++c) enc[c] = a.rays.encoding[rid * E + c];
  for (int c = 0; c < a.color_chn; ++c) facc[c] = 0.0f;
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  float nlt = 0.0f, nlt_lo = 0.0f, t_prev = 1.0f, len = 0.0f, depth_prev = 0.0f;
  const int n_ckpt = ckpt_count(a.march);
  const int craw = p.col[a.color.n_layers - 1];
  int s_last = s_tot - 1;  // last sample marched
  for (int s = 0; s < s_tot; ++s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float delta;
    if (s == 0)
      delta = sample_delta(0, a.march, ray.near_t, ray.far_t, depth);
    else
      delta = depth - depth_prev;
    depth_prev = depth;
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    float occ = 1.0f;
    if (a.scaffold) occ = scaffold_lookup(a.scaffold, a.scaffold_shape, ray.b, x, y, z);
    float raw = decode(ga, ray, x, y, z, enc, act, lds, (int)threadIdx.x);
    if (a.noise_sigma > 0.0f)
      raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float opacity = a.gain * softplus_f(raw) * occ;
    nlt_add(nlt, nlt_lo, opacity * delta);
    if (a.neg_log_t_ckpt && valid) {
      const int ck = ckpt_index(s, a.march);
      if (ck >= 0) *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + ck) * 2) = make_float2(nlt, nlt_lo);
    }
    const float t = __expf(-nlt);
    const float w = t_prev - t;
    t_prev = t;
    len = fmaf(w, depth, len);
    for (int c = 0; c < a.color_chn; ++c)
      facc[c] = fmaf(w, sigmoid_f(act[craw + c]) * occ, facc[c]);
    if (a.stop_neg_log_t > 0.0f && __ballot(valid && nlt < a.stop_neg_log_t) == 0) {  // early termination
      s_last = s;
      break;
    }
  }
  if (valid) {
    write_ray_outputs(a, ray_id, len, nlt, facc);
    if (a.neg_log_t_ckpt)
      *reinterpret_cast<float2*>(a.neg_log_t_ckpt + (ray_id * n_ckpt + n_ckpt - 1) * 2) = make_float2((float)s_last, nlt_lo);
  }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------

// DUMP (test hook, instantiated under -DLP_TEST_HOOKS): the ReLU decisions of the recompute are also written to ga.relu_dump -- same
// instruction sequence, stores added.
template <int ACT_CAP, bool LDS_ACC, bool DUMP = false>
__global__ void LP_GEN_BWD_BOUNDS renderer_bwd_generic(const GenArgs ga) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const LpRendererArgs& a = ga.a;
  const GenPlan& p = ga.p;
  const int lane = threadIdx.x;
  const int64_t ray_id = (int64_t)blockIdx.x * 64 + lane;
  const bool valid = ray_id < a.rays.n_rays;
  const int64_t rid = valid ? ray_id : 0;

  float* Xs = lds;
  float* Ys = lds + 64 * ga.stage_ld;
  float* gparams_lds = lds + 128 * ga.stage_ld;
  float* gparams = nullptr;
  if (a.grad_mlp_params) {
    if (LDS_ACC) {
      for (int64_t i = lane; i < a.n_mlp_params; i += 64) gparams_lds[i] = 0.0f;
      gparams = gparams_lds;
    } else {
      gparams = a.grad_mlp_params;
    }
  }
  __syncthreads();

  float act[ACT_CAP];
  float enc[LP_MAX_WIDTH], denc[LP_MAX_WIDTH];
  float gfeat[LP_MAX_WIDTH];
  float dy[LP_MAX_WIDTH], dx[LP_MAX_WIDTH], dhead[LP_MAX_WIDTH];
  const Ray ray = load_ray(a.rays, rid);
  const int E = a.rays.encoding_dim;
  for (int c = 0; c < E; ++c) {
    enc[c] = a.rays.encoding[rid * E + c];
    denc[c] = 0.0f;
  }
  const int Cc = a.color_chn;
  for (int c = 0; c < Cc; ++c)
    gfeat[c] = (valid && a.grad_feature) ? a.grad_feature[rid * Cc + c] : 0.0f;
  const float g_len = (valid && a.grad_ray_length) ? a.grad_ray_length[rid] : 0.0f;
  const float g_nlt = epilogue_grad_nlt(a, rid, valid, a.neg_log_t[rid],
                                        (valid && a.grad_neg_log_t) ? a.grad_neg_log_t[rid] : 0.0f, gfeat, Cc);

  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  const bool contract = a.march.contract_coords != 0;
  const bool mask = a.march.mask_out_of_bounds != 0;
  const bool two_grids = a.color_grid.n_grids > 0;
  const int C = a.grid.channels;
  const int craw = p.col[a.color.n_layers - 1];

  float nlt = a.neg_log_t[rid], nlt_lo = 0.0f;  // -log T after the last marched sample
  const int n_ckpt = ckpt_count(a.march);
  // closing pair of the checkpoint list: last sample the forward marched for this wave (early termination,
  // wave-uniform) and the low word of the final -log T
  int s_begin = s_tot - 1;
  if (a.neg_log_t_ckpt) {
    const float2 e2 = *reinterpret_cast<const float2*>(a.neg_log_t_ckpt + (rid * n_ckpt + n_ckpt - 1) * 2);
    s_begin = __builtin_amdgcn_readfirstlane((int)e2.x);
    s_begin = s_begin < 0 ? 0 : (s_begin > s_tot - 1 ? s_tot - 1 : s_begin);
    nlt_lo = e2.y;
  }
  float suffix = 0.0f;           // sum_{i >= k} T_i (p_i - p_{i+1})
  float p_next = 0.0f;
  for (int s = s_begin; s >= 0; --s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    const float delta = sample_delta(s, a.march, ray.near_t, ray.far_t, depth);
    float x, y, z;
    sample_point(ray, depth, contract, x, y, z);
    float occ = 1.0f;
    if (a.scaffold) occ = scaffold_lookup(a.scaffold, a.scaffold_shape, ray.b, x, y, z);
    float raw = decode(ga, ray, x, y, z, enc, act, Xs, lane);
    if constexpr (DUMP) {
      // sites in the reference's evaluation order (naive_renderer.py:328-501): single grid-list: trunk layers, opacity hidden layers,
      // colour hidden layers; two-grid decoder: relu(feature), opacity hidden layers, relu(colour feature), colour hidden layers.
      // An activation (two-grid: the raw sample) is > 0 exactly where the unit is active -- the test every mask of this backward applies.
      uint32_t* const dsite = ga.relu_dump + (rid * (int64_t)s_tot + s) * ga.dump_words;
      int k = 0;
      auto put = [&](int off, int width) {
        for (int w0 = 0; w0 < ga.dump_wps; ++w0) {
          unsigned m = 0;
          for (int b = 0; b < 32; ++b) {
            const int c = 32 * w0 + b;
            if (c < width && act[off + c] > 0.0f) m |= 1u << b;
          }
          if (valid) dsite[k * ga.dump_wps + w0] = m;
        }
        ++k;
      };
      if (two_grids) put(p.x0, C);
      else
        for (int l = 0; l < a.trunk.n_layers; ++l) put(p.trunk[l], a.trunk.dims[l + 1]);
      for (int l = 0; l + 1 < a.opacity.n_layers; ++l) put(p.op[l], a.opacity.dims[l + 1]);
      if (two_grids) put(p.cx0, C);
      for (int l = 0; l + 1 < a.color.n_layers; ++l) put(p.col[l], a.color.dims[l + 1]);
      if (valid) dsite[k * ga.dump_wps] = 1u;  // (this kernel marches every sample up to the forward's last one: visited = contributed)
    }
    if (a.noise_sigma > 0.0f)
      raw = raw + sample_noise(rid, s, a.rays.n_rays, s_tot, a.noise_seed) * a.noise_sigma;
    const float sp = softplus_f(raw);
    const float opacity = a.gain * sp * occ;
    // transmittance after (t_i) and before (t_im1) this sample; re-anchor on the exact
    // forward value wherever a checkpoint exists (bounds the subtractive drift)
    if (a.neg_log_t_ckpt) {
      const int ck = ckpt_index(s, a.march);
      if (ck >= 0) {
        const float2 c2 = *reinterpret_cast<const float2*>(a.neg_log_t_ckpt + (rid * n_ckpt + ck) * 2);
        nlt = c2.x;
        nlt_lo = c2.y;
      }
    }
    const float t_i = __expf(-nlt);
    nlt_add(nlt, nlt_lo, -(opacity * delta));
    if (!(nlt > 0.0f)) { nlt = 0.0f; nlt_lo = 0.0f; }
    const float t_im1 = __expf(-nlt);
    const float w = t_im1 - t_i;
    // p_i = g_len * depth + sum_c g_feat_c * colour_c
    float p_i = g_len * depth;
    for (int c = 0; c < Cc; ++c) p_i = fmaf(gfeat[c], sigmoid_f(act[craw + c]) * occ, p_i);
    suffix = fmaf(t_i, p_i - p_next, suffix);
    p_next = p_i;
    const float d_a = suffix + g_nlt;  // d loss / d (delta * opacity)
    const bool live = valid;
    // grid gradients: whole rows per instruction through the staging tiles (splat_list_wave), per lane only when the tiles are too narrow
    auto scatter = [&](const LpGridList& gl, float* const* grad, const float* d) {
      if (splat_wave_ok(ga.stage_ld))
        splat_list_wave(gl, grad, ray.b, x, y, z, mask, d, live, Xs, Ys, ga.stage_ld, lane);
      else if (live)
        splat_list(gl, grad, ray.b, x, y, z, mask, d);
    };
    const float d_raw_op = live ? d_a * delta * a.gain * occ * d_softplus_f(raw) : 0.0f;

    // ---- colour head ----
    for (int c = 0; c < Cc; ++c) {
      const float sg = sigmoid_f(act[craw + c]);
      dy[c] = live ? w * gfeat[c] * occ * sg * (1.0f - sg) : 0.0f;
    }
    if (LDS_ACC)
      mlp_backward<true>(a.mlp_params, ga.stage_ld, a.color, Cc, p.col_in, p.col, act, dy, dx, gparams, Xs, Ys, lane, live);
    else
      mlp_backward<false>(a.mlp_params, ga.stage_ld, a.color, Cc, p.col_in, p.col, act, dy, dx, gparams, Xs, Ys, lane, live);
    const int hw = p.head_w;
    {
      int c = 0;
      for (; c + 8 <= hw; c += 8) {  // (eight reads of each private array in flight)
        float x8[8], e8[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          x8[q] = dx[c + q];
          e8[q] = denc[c + q];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          dhead[c + q] = x8[q];
          denc[c + q] = e8[q] + x8[q];
        }
      }
      for (; c < hw; ++c) {
        dhead[c] = dx[c];
        denc[c] += dx[c];
      }
    }
    // ---- opacity head ----
    dy[0] = d_raw_op;
    if (LDS_ACC)
      mlp_backward<true>(a.mlp_params, ga.stage_ld, a.opacity, 1, p.op_in, p.op, act, dy, dx, gparams, Xs, Ys, lane, live);
    else
      mlp_backward<false>(a.mlp_params, ga.stage_ld, a.opacity, 1, p.op_in, p.op, act, dy, dx, gparams, Xs, Ys, lane, live);

    if (two_grids) {
      // opacity input = relu(x0), colour input = relu(cx0) + enc
      for (int c = 0; c < C; ++c) {
        dx[c] = (act[p.x0 + c] > 0.0f) ? dx[c] : 0.0f;
        dhead[c] = (act[p.cx0 + c] > 0.0f) ? dhead[c] : 0.0f;
      }
      if (a.grad_grid_list[0]) scatter(a.grid, a.grad_grid_list, dx);
      if (a.grad_color_grid_list[0]) scatter(a.color_grid, a.grad_color_grid_list, dhead);
    } else {
      // trunk output gradient = colour-input grad + opacity-input grad, through the ReLU
      {
        int c = 0;
        for (; c + 8 <= hw; c += 8) {
          float h8[8], x8[8], a8[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            h8[q] = dhead[c + q];
            x8[q] = dx[c + q];
            a8[q] = act[p.op_in + c + q];
          }
#pragma unroll
          for (int q = 0; q < 8; ++q) dy[c + q] = (a8[q] > 0.0f) ? h8[q] + x8[q] : 0.0f;
        }
        for (; c < hw; ++c) {
          const float g = dhead[c] + dx[c];
          dy[c] = (act[p.op_in + c] > 0.0f) ? g : 0.0f;
        }
      }
      if (a.trunk.n_layers > 0) {
        // mlp_backward applies ReLU masks only to hidden outputs; the trunk's last layer
        // is ReLU'd too and was handled just above (op_in aliases trunk[n-1]).
        const LpMlp& m = a.trunk;
        if (LDS_ACC)
          mlp_backward<true>(a.mlp_params, ga.stage_ld, m, m.dims[m.n_layers], p.x0, p.trunk, act, dy, dx, gparams, Xs, Ys, lane, live);
        else
          mlp_backward<false>(a.mlp_params, ga.stage_ld, m, m.dims[m.n_layers], p.x0, p.trunk, act, dy, dx, gparams, Xs, Ys, lane, live);
      } else {
        for (int c = 0; c < C; ++c) dx[c] = dy[c];
      }
      if (a.grad_grid_list[0]) scatter(a.grid, a.grad_grid_list, dx);
    }
  }
  if (valid && a.grad_encoding)
    for (int c = 0; c < E; ++c) a.grad_encoding[ray_id * E + c] = denc[c];
  if (LDS_ACC && a.grad_mlp_params) {
    __syncthreads();
    for (int64_t i = lane; i < a.n_mlp_params; i += 64) {
      const float v = gparams_lds[i];
      if (v != 0.0f) atomic_add_f32(a.grad_mlp_params + i, v);
    }
  }
}

// ---------------------------------------------------------------------------------------
// debug hook: integer corner rows
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) renderer_corner_rows(const LpRendererArgs a, int64_t* rows, int k_tot) {
  const int64_t ray_id = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (ray_id >= a.rays.n_rays) return;
  const Ray ray = load_ray(a.rays, ray_id);
  const int s_tot = a.march.num_samples + a.march.num_samples_inf;
  for (int s = 0; s < s_tot; ++s) {
    const float depth = sample_depth(s, a.march, ray.near_t, ray.far_t);
    float x, y, z;
    sample_point(ray, depth, a.march.contract_coords != 0, x, y, z);
    int64_t* dst = rows + (ray_id * s_tot + s) * k_tot;
    int pos = 0;
    for (int g = 0; g < a.grid.n_grids; ++g) {
      const Corners cs = grid_corners<false>(a.grid.grids[g], ray.b, x, y, z);
      for (int k = 0; k < cs.n; ++k) {
        // report rows relative to the start of grid g (the oracle indexes each grid separately)
        dst[pos++] = cs.row[k] < 0 ? (int64_t)-1 : cs.row[k] - a.grid.grids[g].row_offset;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------

static int make_plan(const LpRendererArgs& a, GenPlan& p) {
  int pos = 0;
  const int C = a.grid.channels;
  const bool two = a.color_grid.n_grids > 0;
  p.x0 = pos; pos += C;
  p.cx0 = -1;
  if (two) { p.cx0 = pos; pos += C; }
  int w = C;
  for (int l = 0; l < a.trunk.n_layers; ++l) { p.trunk[l] = pos; pos += a.trunk.dims[l + 1]; w = a.trunk.dims[l + 1]; }
  if (!two && a.trunk.n_layers > 0) {
    p.op_in = p.trunk[a.trunk.n_layers - 1];
  } else {
    p.op_in = pos; pos += C; w = C;
  }
  p.head_w = w;
  p.col_in = pos; pos += w;
  for (int l = 0; l < a.opacity.n_layers; ++l) { p.op[l] = pos; pos += a.opacity.dims[l + 1]; }
  for (int l = 0; l < a.color.n_layers; ++l) { p.col[l] = pos; pos += a.color.dims[l + 1]; }
  p.total = pos;
  return pos;
}

// row stride (floats) of the waves' LDS staging tiles: the widest layer input / output + 1 (odd for the usual even widths:
// conflict-free rows)
static int generic_stage_ld(const LpRendererArgs& a) {
  int maxw = a.grid.channels;
  const LpMlp* ms[3] = {&a.trunk, &a.opacity, &a.color};
  for (const LpMlp* m : ms)
    for (int l = 0; l <= m->n_layers && m->n_layers > 0; ++l) maxw = m->dims[l] > maxw ? m->dims[l] : maxw;
  return maxw + 1;
}

int renderer_forward_generic(const LpRendererArgs& a, hipStream_t stream) {
  GenArgs ga;
  ga.a = a;
  const int total = make_plan(a, ga.p);
  ga.stage_ld = generic_stage_ld(a);
  ga.lds_param_accum = 0;
  ga.relu_dump = nullptr;
  ga.dump_words = ga.dump_wps = 0;
  const size_t lds = (size_t)64 * ga.stage_ld * sizeof(float);  // <= 33 KB (LP_MAX_WIDTH 128): below the 64 KB default limit
  const unsigned blocks = (unsigned)((a.rays.n_rays + 63) / 64);
  if (blocks == 0) return LP_OK;
  if (total <= 256)
    hipLaunchKernelGGL(renderer_fwd_generic<256>, dim3(blocks), dim3(64), lds, stream, ga);
  else if (total <= 1024)
    hipLaunchKernelGGL(renderer_fwd_generic<1024>, dim3(blocks), dim3(64), lds, stream, ga);
  else
    return set_error(LP_EUNSUPPORTED, "generic renderer: sum of layer widths %d exceeds 1024", total);
  return check_launch("renderer_fwd_generic");
}

// ReLU dump of the shape-generic backward: sites and words per site (include/lightplane_hip.h, lp_renderer_relu_dump_words)
static void generic_dump_shape(const LpRendererArgs& a, int& n_sites, int& wps) {
  const bool two = a.color_grid.n_grids > 0;
  int maxw = 1;
  n_sites = 0;
  auto site = [&](int w) { ++n_sites; maxw = w > maxw ? w : maxw; };
  if (two) { site(a.grid.channels); site(a.grid.channels); }
  else for (int l = 0; l < a.trunk.n_layers; ++l) site(a.trunk.dims[l + 1]);
  for (int l = 0; l + 1 < a.opacity.n_layers; ++l) site(a.opacity.dims[l + 1]);
  for (int l = 0; l + 1 < a.color.n_layers; ++l) site(a.color.dims[l + 1]);
  wps = (maxw + 31) / 32;
}
int renderer_generic_dump_words(const LpRendererArgs& a) {
#ifdef LP_TEST_HOOKS
  int n_sites, wps;
  generic_dump_shape(a, n_sites, wps);
  return n_sites * wps + 1;
#else
  (void)a;
  return set_error(LP_EUNSUPPORTED, "relu dump: the library was built without -DLP_TEST_HOOKS");
#endif
}

int renderer_backward_generic(const LpRendererArgs& a, hipStream_t stream) {
  GenArgs ga;
  ga.a = a;
  ga.relu_dump = g_relu_dump;  // test hook (NULL in every product call)
  {
    int n_sites;
    generic_dump_shape(a, n_sites, ga.dump_wps);
    ga.dump_words = n_sites * ga.dump_wps + 1;
  }
  const int total = make_plan(a, ga.p);
  ga.stage_ld = generic_stage_ld(a);
  const size_t stage_bytes = (size_t)128 * ga.stage_ld * sizeof(float);
  const size_t param_bytes = (size_t)a.n_mlp_params * sizeof(float);
  const bool lds_acc = a.grad_mlp_params && (stage_bytes + param_bytes <= 96 * 1024);
  ga.lds_param_accum = lds_acc ? 1 : 0;
  const size_t lds = stage_bytes + (lds_acc ? param_bytes : 0);
  const unsigned blocks = (unsigned)((a.rays.n_rays + 63) / 64);
  if (blocks == 0) return LP_OK;
  if (total > 1024)
    return set_error(LP_EUNSUPPORTED, "generic renderer: sum of layer widths %d exceeds 1024", total);
#define LP_LAUNCH_BWD(CAP, ACC)                                                                   \
  do {                                                                                            \
    hipError_t e = hipFuncSetAttribute((const void*)renderer_bwd_generic<CAP, ACC>,               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
    if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e)); \
    hipLaunchKernelGGL((renderer_bwd_generic<CAP, ACC>), dim3(blocks), dim3(64), lds, stream, ga); \
  } while (0)
#ifdef LP_TEST_HOOKS
  if (ga.relu_dump) {  // the DUMP twins
#define LP_LAUNCH_BWD_DUMP(CAP, ACC)                                                                    \
  do {                                                                                                  \
    hipError_t e = hipFuncSetAttribute((const void*)renderer_bwd_generic<CAP, ACC, true>,               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);           \
    if (e != hipSuccess) return set_error((int)e, "hipFuncSetAttribute: %s", hipGetErrorString(e));     \
    hipLaunchKernelGGL((renderer_bwd_generic<CAP, ACC, true>), dim3(blocks), dim3(64), lds, stream, ga); \
  } while (0)
    if (total <= 256) {
      if (lds_acc) LP_LAUNCH_BWD_DUMP(256, true); else LP_LAUNCH_BWD_DUMP(256, false);
    } else {
      if (lds_acc) LP_LAUNCH_BWD_DUMP(1024, true); else LP_LAUNCH_BWD_DUMP(1024, false);
    }
#undef LP_LAUNCH_BWD_DUMP
    return check_launch("renderer_bwd_generic (dump)");
  }
#endif
  if (total <= 256) {
    if (lds_acc) LP_LAUNCH_BWD(256, true); else LP_LAUNCH_BWD(256, false);
  } else {
    if (lds_acc) LP_LAUNCH_BWD(1024, true); else LP_LAUNCH_BWD(1024, false);
  }
#undef LP_LAUNCH_BWD
  return check_launch("renderer_bwd_generic");
}

int renderer_corner_rows_launch(const LpRendererArgs& a, int64_t* rows, hipStream_t stream) {
  int k_tot = 0;
  for (int g = 0; g < a.grid.n_grids; ++g) {
    const LpGrid& gd = a.grid.grids[g];
    k_tot += (gd.D > 1 && gd.H > 1 && gd.W > 1) ? 8 : 4;
  }
  const unsigned blocks = (unsigned)((a.rays.n_rays + 63) / 64);
  if (blocks == 0) return LP_OK;
  hipLaunchKernelGGL(renderer_corner_rows, dim3(blocks), dim3(64), 0, stream, a, rows, k_tot);
  return check_launch("renderer_corner_rows");
}

}  // namespace lp
