// lp_mfma_common.h -- building blocks shared by the MFMA Renderer kernels (forward:
// lp_renderer_mfma.hip, backward: lp_renderer_mfma_bwd.hip).  See the header comment of
// lp_renderer_mfma.hip for the lane <-> (ray, feature) mapping.
#pragma once
#include <stdlib.h>

#include "lp_device.h"
#include "lp_splat_walk.h"
#include "lp_host.h"

namespace lp {

typedef float f32x16 __attribute__((ext_vector_type(16)));

LP_DEV constexpr int featq(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

#define LP_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// An integer the optimiser must treat as unknown (it is always 0).  Added to LDS offsets inside
// the sample loop it stops LICM from hoisting the ~200 loop-invariant weight / bias reads out of
// the loop (which costs >200 VGPRs and spills); the reads stay ds_read (LDS address space kept).
LP_DEV int opaque_zero() {
  int z = 0;
  asm volatile("" : "+s"(z));
  return z;
}

constexpr int HID = 32;        // hidden width of the shape family
constexpr int W_LD = 33;       // padded row stride of the weight matrices in LDS
constexpr int WAVES = 4;       // waves per workgroup
constexpr int RAYS_PER_WAVE = 32;
constexpr int MAX_INF = 256;   // beyond-far samples tabulated in LDS

// grid-list shape the kernel is specialised for
constexpr int GM_GENERIC = 0;   // run-time loop over the grid-list
constexpr int GM_TRIPLANE = 1;  // a canonical triplane (is_canonical_triplane): xy, xz, yz planes, one size per axis
constexpr int GM_VOXEL = 2;     // exactly one voxel grid

// float offsets of the parameter blocks inside mlp_params (computed on the host)
struct MfmaParams {
  int64_t w_t1, w_t2, b_t1, b_t2;  // trunk
  int64_t w_o1, w_o2, b_o1, b_o2;  // opacity
  int64_t w_c1, w_c2, b_c1, b_c2;  // colour
  int ldc2;                        // row stride of w_c2 (padded colour width)
  int dbg;                         // LP_MFMA_DEBUG bits (timing experiments only)
  // shape flexibility of the width-32 family: actual hidden width (16 or 32; staged zero-padded to 32),
  // second trunk layer present, hidden layer of the opacity / colour head present
  int hid, t2, oh, ch;
  // two-grid decoder (separate colour grid, no trunk): t1 = trunk layer 1 present, tg = colour grid-list present,
  // hin = input width of the heads (grid channels with tg, hid otherwise) = width of the ray encoding
  int t1, tg, hin;
  // segment-parallel backward: LP_SEG_LEN-sample blocks per workgroup (chosen at launch, see launch_bwd3)
  int seg_blocks;
  int seg_fwd;  // forward of the flex / two-grid shapes: 1 = this launch marches segments (segment-local state records)
  // test hook (lp_renderer_backward_relu_dump): the ReLU decisions of the backward's recompute, [ray][sample][5] words -- t1, t2, o1,
  // c1 (bit f = unit f active) and 1 = "this sample was visited".  Only the DUMP instantiations read it.
  uint32_t* relu_dump;
  int tm_rpw;  // transposed march (lp_renderer_mfma_bwd_tm.hip): rays per wave (1 .. 32, chosen at launch: small batches spread over the chip)
};

// LDS map (floats).  Weight matrices are kept ONCE, row-major [in][W_LD] with a padded row
// stride of 33: the forward operand W[feat(kk,h)][l&31] walks a row (conflict-free), the
// backward operand W[l&31][feat(kk,h)] walks a column with stride 33 (conflict-free as well).
struct Lds {
  static constexpr int WT1 = 0;                  // [32][33] (rows >= C are zero)
  static constexpr int WT2 = WT1 + 32 * W_LD;
  static constexpr int WO1 = WT2 + 32 * W_LD;
  static constexpr int WC1 = WO1 + 32 * W_LD;
  static constexpr int BIAS = WC1 + 32 * W_LD;   // b_t1, b_t2, b_o1, b_c1 : 4 x 32
  static constexpr int WO2 = BIAS + 4 * 32;      // [32]
  static constexpr int WC2 = WO2 + 32;           // [32][4]
  static constexpr int HB = WC2 + 32 * 4;        // bo2, bc2[0..3], pad -> 8
  static constexpr int INF = HB + 8;             // [MAX_INF] depth scale of the beyond-far samples
  static constexpr int FWD_END = INF + MAX_INF;
};

// (stage_weights, the fp32 [32][33] images of the first-generation Renderer kernels, went with them in round 4; the layout
// constants above stay: the bf16x3 kernels address their small fp32 block -- biases, output layers, beyond-far table --
// through them.)
// bias of layer `which` (0 t1, 1 t2, 2 o1, 3 c1) in accumulator-register order for half h
LP_DEV f32x16 load_bias(const float* lds, int which, int h, int zo) {
  const float4* b = reinterpret_cast<const float4*>(lds + Lds::BIAS + which * 32 + 4 * h + zo);
  f32x16 acc;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = b[2 * j];  // floats 8j + 4h .. +3
    acc[4 * j + 0] = v.x; acc[4 * j + 1] = v.y; acc[4 * j + 2] = v.z; acc[4 * j + 3] = v.w;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------
// grid-list gather: interpolated feature of this lane's ray, channels feat(q,h), q < C/2
// ---------------------------------------------------------------------------------------
template <int C>
LP_DEV void gather_tap(const float* data, int row, float w, int h, float (&x0)[C / 2]) {
  // out-of-range taps carry weight 0 and read row 0: no branch
  const float4* src = reinterpret_cast<const float4*>(data + (int64_t)(row < 0 ? 0 : row) * C + 4 * h);
#pragma unroll
  for (int j = 0; j < C / 8; ++j) {
    const float4 v = src[2 * j];  // channels 8j + 4h .. +3
    x0[4 * j + 0] = fmaf(w, v.x, x0[4 * j + 0]);
    x0[4 * j + 1] = fmaf(w, v.y, x0[4 * j + 1]);
    x0[4 * j + 2] = fmaf(w, v.z, x0[4 * j + 2]);
    x0[4 * j + 3] = fmaf(w, v.w, x0[4 * j + 3]);
  }
}

// Four taps at once: all their loads are issued before the first one is consumed (the scheduling fence keeps
// the compiler from recycling one register quad for every load, which serialises 4 * C/8 memory round trips);
// accumulation order is that of four gather_tap() calls.
template <int C>
LP_DEV void gather_taps4(const float* data, const int* row, const float* w, float keep, int h, float (&x0)[C / 2]) {
  float4 v[4][C / 8];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float4* src = reinterpret_cast<const float4*>(data + (int64_t)(row[k] < 0 ? 0 : row[k]) * C + 4 * h);
#pragma unroll
    for (int j = 0; j < C / 8; ++j) v[k][j] = src[2 * j];
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float wk = w[k] * keep;
#pragma unroll
    for (int j = 0; j < C / 8; ++j) {
      x0[4 * j + 0] = fmaf(wk, v[k][j].x, x0[4 * j + 0]);
      x0[4 * j + 1] = fmaf(wk, v[k][j].y, x0[4 * j + 1]);
      x0[4 * j + 2] = fmaf(wk, v[k][j].z, x0[4 * j + 2]);
      x0[4 * j + 3] = fmaf(wk, v[k][j].w, x0[4 * j + 3]);
    }
  }
}

// gather from an explicit grid-list (run-time loop over its grids): the colour grid-list of the two-grid decoder
template <int C, bool FENCED>
LP_DEV void gather_list(const LpGridList& gl, bool mask_oob, const Ray& ray, float x, float y, float z, int h,
                        float (&x0)[C / 2]) {
#pragma unroll
  for (int q = 0; q < C / 2; ++q) x0[q] = 0.0f;
  const float keep = (mask_oob && !point_in_bounds(x, y, z)) ? 0.0f : 1.0f;
  for (int g = 0; g < gl.n_grids; ++g) {
    Taps t;
    const float* data = gl.grids[g].data;  // per-grid base (the host normalises NULL to the flat tensor)
    grid_taps<false>(gl.grids[g], ray.b, x, y, z, t);
    if (FENCED) __builtin_amdgcn_sched_barrier(0);
    gather_taps4<C>(data, t.row, t.w, keep, h, x0);
    if (FENCED) __builtin_amdgcn_sched_barrier(0);
    if (t.n == 8) {
      gather_taps4<C>(data, t.row + 4, t.w + 4, keep, h, x0);
      if (FENCED) __builtin_amdgcn_sched_barrier(0);
    }
  }
}

template <int C, int GM, bool FENCED = false>
LP_DEV void gather_features(const LpRendererArgs& a, const Ray& ray, float x, float y, float z, int h,
                            float (&x0)[C / 2]) {
#pragma unroll
  for (int q = 0; q < C / 2; ++q) x0[q] = 0.0f;
  const float keep = (a.march.mask_out_of_bounds && !point_in_bounds(x, y, z)) ? 0.0f : 1.0f;
  if (GM == GM_TRIPLANE) {
    if (FENCED) {
      // taps of the three planes first, then one plane's loads in flight at a time
      Taps t[3];
      triplane_taps<false>(a.grid.grids, ray.b, x, y, z, t);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        __builtin_amdgcn_sched_barrier(0);
        gather_taps4<C>(a.grid.grids[g].data, t[g].row, t[g].w, keep, h, x0);
      }
      __builtin_amdgcn_sched_barrier(0);
    } else {
      Taps t[3];
      triplane_taps<false>(a.grid.grids, ray.b, x, y, z, t);
#pragma unroll
      for (int g = 0; g < 3; ++g) {
#pragma unroll
        for (int k = 0; k < 4; ++k) gather_tap<C>(a.grid.grids[g].data, t[g].row[k], t[g].w[k] * keep, h, x0);
      }
    }
  } else if (GM == GM_VOXEL) {
    Taps t;
    voxel_taps<false>(a.grid.grids[0], ray.b, x, y, z, t);
    if (FENCED) {
      __builtin_amdgcn_sched_barrier(0);
      gather_taps4<C>(a.grid.grids[0].data, t.row, t.w, keep, h, x0);
      __builtin_amdgcn_sched_barrier(0);
      gather_taps4<C>(a.grid.grids[0].data, t.row + 4, t.w + 4, keep, h, x0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) gather_tap<C>(a.grid.grids[0].data, t.row[k], t.w[k] * keep, h, x0);
    }
  } else {
    for (int g = 0; g < a.grid.n_grids; ++g) {
      Taps t;
      const float* data = a.grid.grids[g].data;
      grid_taps<false>(a.grid.grids[g], ray.b, x, y, z, t);
      if (FENCED) {
        __builtin_amdgcn_sched_barrier(0);
        gather_taps4<C>(data, t.row, t.w, keep, h, x0);
        __builtin_amdgcn_sched_barrier(0);
        if (t.n == 8) {
          gather_taps4<C>(data, t.row + 4, t.w + 4, keep, h, x0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) gather_tap<C>(data, t.row[k], t.w[k] * keep, h, x0);
        if (t.n == 8) {
#pragma unroll
          for (int k = 4; k < 8; ++k) gather_tap<C>(data, t.row[k], t.w[k] * keep, h, x0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// decoder
// ---------------------------------------------------------------------------------------

struct Heads {
  float raw_o;
  float raw_c[4];
};

// opacity / colour output layers on the VALU (N = 1 and N <= 4): each lane covers its 16
// features, the partner lane (l ^ 32) the other 16.
// NC = colour channels evaluated (3: the fourth column of the output layer is padding and skipped)
template <int NC = 4>
LP_DEV Heads heads_forward(const float* lds_, int h, const float (&ho)[16], const float (&hc)[16], int zo) {
  using M = Lds;
  const float* lds = lds_ + zo;
  float po = 0.0f, pc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 wo = *reinterpret_cast<const float4*>(lds + M::WO2 + 8 * j + 4 * h);
    const float wov[4] = {wo.x, wo.y, wo.z, wo.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int q = 4 * j + i;
      po = fmaf(ho[q], wov[i], po);
      const float4 wc = *reinterpret_cast<const float4*>(lds + M::WC2 + (8 * j + 4 * h + i) * 4);
      pc[0] = fmaf(hc[q], wc.x, pc[0]);
      pc[1] = fmaf(hc[q], wc.y, pc[1]);
      pc[2] = fmaf(hc[q], wc.z, pc[2]);
      if (NC > 3) pc[3] = fmaf(hc[q], wc.w, pc[3]);
    }
  }
  Heads o;
  o.raw_o = (po + __shfl_xor(po, 32)) + lds[M::HB];
#pragma unroll
  for (int c = 0; c < 4; ++c) o.raw_c[c] = (c < NC) ? (pc[c] + __shfl_xor(pc[c], 32)) + lds[M::HB + 1 + c] : 0.0f;
  return o;
}

// Activations of one sample (accumulator-register order).
template <int C>
struct Act {
  float x0[C / 2];
  float h1[16], e[16], ho[16], hc[16];
};

// Nothing may be scheduled across this point.  Used to cut the sample loop body into groups of
// "one layer's MFMAs + one plane's gather": inside a group the scheduler interleaves freely, but it
// can no longer hoist all 24 dwordx4 loads of a sample to the top (96 live VGPRs).
#ifdef LP_NO_FENCE
#define LP_SCHED_FENCE()
#else
#define LP_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// this lane's 16 encoding features feat(q,h); `hid` = actual width (features >= hid read as 0)
LP_DEV void load_encoding(const LpRendererArgs& a, int64_t rid, int h, float (&enc)[16], int hid = HID) {
  const float4* src = reinterpret_cast<const float4*>(a.rays.encoding + rid * hid + 4 * h);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 v = (8 * j + 4 * h < hid) ? src[2 * j] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    enc[4 * j + 0] = v.x; enc[4 * j + 1] = v.y; enc[4 * j + 2] = v.z; enc[4 * j + 3] = v.w;
  }
}

// geometry of one sample + its (prefetched) grid feature
template <int C>
struct Sample {
  float depth, occ, x, y, z;
  float x0[C / 2];
};

// PLAIN: no beyond-far samples, no contraction, no scaffold (the kernels are specialised on it)
template <int C, bool PLAIN = false>
LP_DEV void sample_geometry(const LpRendererArgs& a, const float* lds, const Ray& ray, int s, Sample<C>& o) {
  if (PLAIN) {
    o.depth = ray.near_t + lin01(s, a.march.num_samples) * (ray.far_t - ray.near_t);
    sample_point(ray, o.depth, false, o.x, o.y, o.z);
    o.occ = 1.0f;
    return;
  }
  o.depth = sample_depth_tab(s, a.march, ray.near_t, ray.far_t, lds + Lds::INF);
  sample_point(ray, o.depth, a.march.contract_coords != 0, o.x, o.y, o.z);
  o.occ = 1.0f;
  if (a.scaffold) o.occ = scaffold_lookup(a.scaffold, a.scaffold_shape, ray.b, o.x, o.y, o.z);
}

template <int C, int GM, bool FENCED = false, bool PLAIN = false>
LP_DEV void fetch_sample(const LpRendererArgs& a, const float* lds, const Ray& ray, int s, int h, Sample<C>& o) {
  sample_geometry<C, PLAIN>(a, lds, ray, s, o);
  gather_features<C, GM, FENCED>(a, ray, o.x, o.y, o.z, h, o.x0);
}

// Gradient scatter, row-contiguous and run-length merged.
// dx0 of the wave's 32 rays has been transposed through LDS ([channel][ray], row stride DX_LD):
// every lane reads its channel(s) of all rays, lane group `grp` (16 lanes) works on tap slot k.  Rays are walked
// in order; consecutive rays that fall into the same cell (the common case for image-coherent rays)
// are summed in a register and leave as ONE atomic per row whose C lanes cover the C contiguous
// floats of the row.  All slots of a grid change cell together, so the run boundaries are a
// wave-uniform bit mask (ballot in the lane = ray layout): the walk is scalar-branched, the tap rows
// come from v_readlane, and the only LDS traffic is the bulk (b128) load of the weights.
constexpr int DX_LD = 36;  // row stride of the transposed dx0 tile [channel][ray]

template <int C>
LP_DEV void flush_run(float* gg, int s_row, unsigned s_ok, int koff, unsigned kbit, int sub, const float (&run)[C / 16],
                      int dbg) {
  // the run's row goes into the scalar base (64-bit: grid-lists of any size below 2^31 rows), the lane part -- this slot's corner
  // offset inside the cell, this lane's channel -- stays a 32-bit byte offset (a corner is at most one z-slice + one line away)
  if ((s_ok & kbit) && !(dbg & 1)) {
    char* const rb = reinterpret_cast<char*>(gg) + (int64_t)s_row * (C * 4);
    const unsigned lane_off = (unsigned)koff * (unsigned)(C * 4) + (unsigned)(sub * 4);
#pragma unroll
    for (int j = 0; j < C / 16; ++j) atomic_add_f32(reinterpret_cast<float*>(rb + lane_off + 64 * j), run[j]);
  }
}

// Lane layout of the walk: 16 lanes per tap slot (four slots per pass), lane `sub` owns channels
// sub, sub + 16, ... (C/16 of them): every atomic instruction covers four rows x 64 contiguous bytes.
// dxT: the transposed dx0 tile [channel][ray] (row stride DX_LD).
// GMS: what is known about the grid at compile time (GM_TRIPLANE: a plane, GM_VOXEL: a voxel grid, GM_GENERIC: either)
// Plane grids, compile-time known (triplane kernels): the per-slot walk without per-corner validity bits.  Border cells
// are re-expressed so that all four corners are in range (cell -1 becomes cell 0 with the weights moved to the near
// slots, the last cell likewise) and a ray that misses the plane altogether -- or is not live -- gets the sentinel row
// -1, which the flush tests on the scalar unit.  The row part of the address goes into the scalar base of the atomic,
// the lane part is computed once per call: a flush costs one v_readlane, one v_mov and the atomic instead of seven
// VALU instructions.  (Measured: -1 % kernel time only -- the walk is bound by its scalar branches, not by its vector
// instructions; starting a run with a multiply instead of zero + fma was 6 % SLOWER.)
template <int C>
LP_DEV void scatter_plane_ax(float* gg, int base, int U, AxisTap u, AxisTap v, bool live, int lane, const float* dxT,
                             float* wT, int dbg) {
  constexpr int CPL = C / 16;
  const int h = lane >> 5, r = lane & 31, sub = lane & 15, grp = lane >> 4;
  int iu = u.i0, iv = v.i0;
  float wu[2] = {u.w[0], u.w[1]}, wv[2] = {v.w[0], v.w[1]};
  const bool oku[2] = {u.ok[0], u.ok[1]}, okv[2] = {v.ok[0], v.ok[1]};
  const bool dead = !live || !(oku[0] || oku[1]) || !(okv[0] || okv[1]);
  if (oku[0] && !oku[1]) { iu -= 1; wu[1] = wu[0]; wu[0] = 0.0f; }
  else if (!oku[0] && oku[1]) { iu += 1; wu[0] = wu[1]; wu[1] = 0.0f; }
  if (okv[0] && !okv[1]) { iv -= 1; wv[1] = wv[0]; wv[0] = 0.0f; }
  else if (!okv[0] && okv[1]) { iv += 1; wv[0] = wv[1]; wv[1] = 0.0f; }
  const int row0 = dead ? -1 : base + iv * U + iu;
  // weights -> wT[slot][ray]; the two lanes of a ray write two slots each (slot k = u-bit + 2 v-bit)
  wT[(2 * h) * 32 + r] = dead ? 0.0f : wu[0] * wv[h];
  wT[(2 * h + 1) * 32 + r] = dead ? 0.0f : wu[1] * wv[h];
  const int prow = lane_prev(row0);  // all lanes enabled: see run_head()
  const bool head = run_head(r, row0, prow);
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(head));
  const int koff = (grp & 1) + (grp >> 1) * U;
  const unsigned lane_off = (unsigned)koff * (unsigned)(C * 4) + (unsigned)(sub * 4);  // bytes inside a 2 x 2 cell
  const float4* wsrc = reinterpret_cast<const float4*>(wT + grp * 32);
  const float4* dsrc[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) dsrc[j] = reinterpret_cast<const float4*>(dxT + (sub + 16 * j) * DX_LD);
  float run[CPL];
  int s_row = __builtin_amdgcn_readlane(row0, 0);
  auto flush = [&](int row) {
    if (row >= 0 && !(dbg & 1)) {
      char* base = reinterpret_cast<char*>(gg) + (int64_t)row * (C * 4);  // scalar
#pragma unroll
      for (int j = 0; j < CPL; ++j) atomic_add_f32(reinterpret_cast<float*>(base + lane_off + 64 * j), run[j]);
    }
  };
  if (mask == 1u) {  // all 32 rays in one cell: no run logic at all
#pragma unroll
    for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
#pragma unroll
    for (int c4 = 0; c4 < 8; ++c4) {
      const float4 w = wsrc[c4];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float4 d = dsrc[j][c4];
        run[j] = fmaf(w.x, d.x, run[j]);
        run[j] = fmaf(w.y, d.y, run[j]);
        run[j] = fmaf(w.z, d.z, run[j]);
        run[j] = fmaf(w.w, d.w, run[j]);
      }
    }
    flush(s_row);
    return;
  }
#pragma unroll
  for (int c8 = 0; c8 < 4; ++c8) {  // 8 rays at a time
    const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
    const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
    float dx[CPL][8];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
      const float4 d0 = dsrc[j][2 * c8], d1 = dsrc[j][2 * c8 + 1];
      dx[j][0] = d0.x; dx[j][1] = d0.y; dx[j][2] = d0.z; dx[j][3] = d0.w;
      dx[j][4] = d1.x; dx[j][5] = d1.y; dx[j][6] = d1.z; dx[j][7] = d1.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 8 * c8 + i;
      if (rr == 0) {
#pragma unroll
        for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
      } else if ((mask >> rr) & 1u) {
        flush(s_row);
#pragma unroll
        for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
        s_row = __builtin_amdgcn_readlane(row0, rr);
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) run[j] = fmaf(w[i], dx[j][i], run[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  flush(s_row);
}

// one plane (orientation decided at run time).  Sharing the axis computations of a canonical triplane between the
// three planes here as well means three inlined copies of the walk instead of a loop: measured 2.5 % slower.
template <int C>
LP_DEV void scatter_plane(float* gg, const LpGrid& g, int b, float x, float y, float z, bool live, int lane,
                          const float* dxT, float* wT, int dbg) {
  const bool xy = g.D == 1, xz = g.H == 1;
  const float cu = (xy || xz) ? x : y;
  const float cv = xy ? y : z;
  const int U = (xy || xz) ? g.W : g.H;
  const int V = xy ? g.H : g.D;
  AxisTap u, v;
  axis_taps<false>(cu, U, u.i0, u.w, u.ok);
  axis_taps<false>(cv, V, v.i0, v.w, v.ok);
  scatter_plane_ax<C>(gg, (int)g.row_offset + b * (U * V), U, u, v, live, lane, dxT, wT, dbg);
}

// Canonical triplane (GM_TRIPLANE), all three planes in one call.  What the per-plane walk above spent per plane BEFORE its ray
// loop -- two axis computations with their border re-expression, the run-time choice of the plane's orientation, the
// previous lane's row through ds_bpermute, the grid descriptor re-read from the kernel arguments (scalar loads + a
// wait that also drains the LDS queue) -- was ~100 of its ~190 vector instructions (profiles/r05_scatter_*).  Here the three
// axes are evaluated ONCE per sample (three computations instead of six) and normalised (cell index, two weights, dead
// flag); a plane picks its (u, v) pair with eight v_cndmask on the scalar plane index; the previous lane's row comes from a
// DPP wave shift (one VALU instruction, no LDS round trip).  The ray walk itself (scalar-branched run merge, one atomic per
// run and row) is the one of scatter_plane_ax.
struct AxisNorm {
  int i;        // first cell of the pair, both cells in range unless dead
  float w0, w1;
  bool dead;
};
LP_DEV AxisNorm axis_norm(float c, int size) {
  AxisTap t;
  axis_taps<false>(c, size, t.i0, t.w, t.ok);
  const bool only0 = t.ok[0] & !t.ok[1], only1 = !t.ok[0] & t.ok[1];  // a border cell: one tap in range (selects, no branches)
  AxisNorm n;
  n.i = t.i0 + (only1 ? 1 : 0) - (only0 ? 1 : 0);
  n.w0 = only0 ? 0.0f : (only1 ? t.w[1] : t.w[0]);
  n.w1 = only1 ? 0.0f : (only0 ? t.w[0] : t.w[1]);
  n.dead = !(t.ok[0] | t.ok[1]);
  return n;
}
template <int C>
LP_DEV void scatter_triplane(float* const* gg_list, const LpGridList& gl, int b, float x, float y, float z, bool live, int lane,
                             const float* dxT, float* wT, int dbg) {
  constexpr int CPL = C / 16;
  const int h = lane >> 5, r = lane & 31, sub = lane & 15, grp = lane >> 4;
  const int W = gl.grids[0].W, H = gl.grids[0].H, D = gl.grids[1].D;
  const AxisNorm ax = axis_norm(x, W), ay = axis_norm(y, H), az = axis_norm(z, D);
  // ---- all three planes' rows, weights (wT[plane][slot][ray], slot = u-bit + 2 v-bit) and run masks, straight-line ----
  const bool d0 = !live | ax.dead | ay.dead, d1 = !live | ax.dead | az.dead, d2 = !live | ay.dead | az.dead;
  const int row_xy = d0 ? -1 : (int)gl.grids[0].row_offset + (b * H + ay.i) * W + ax.i;
  const int row_xz = d1 ? -1 : (int)gl.grids[1].row_offset + (b * D + az.i) * W + ax.i;
  const int row_yz = d2 ? -1 : (int)gl.grids[2].row_offset + (b * D + az.i) * H + ay.i;
  const float wy = h ? ay.w1 : ay.w0, wz = h ? az.w1 : az.w0;  // the v-weight of this lane's two slots
  float* const wl = wT + (2 * h) * 32 + r;
  wl[0] = d0 ? 0.0f : ax.w0 * wy;
  wl[32] = d0 ? 0.0f : ax.w1 * wy;
  wl[128] = d1 ? 0.0f : ax.w0 * wz;
  wl[160] = d1 ? 0.0f : ax.w1 * wz;
  wl[256] = d2 ? 0.0f : ay.w0 * wz;
  wl[288] = d2 ? 0.0f : ay.w1 * wz;
  const bool first = r == 0;
  const unsigned m_xy = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(first | (row_xy != lane_prev(row_xy))));
  const unsigned m_xz = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(first | (row_xz != lane_prev(row_xz))));
  const unsigned m_yz = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(first | (row_yz != lane_prev(row_yz))));
  const float4* dsrc[CPL];
#pragma unroll
  for (int j = 0; j < CPL; ++j) dsrc[j] = reinterpret_cast<const float4*>(dxT + (sub + 16 * j) * DX_LD);
  // ---- the walk, one plane at a time (the code of the walk exists once) ----
#pragma unroll 1
  for (int g = 0; g < 3; ++g) {
    const int row0 = g == 0 ? row_xy : (g == 1 ? row_xz : row_yz);
    const unsigned mask = g == 0 ? m_xy : (g == 1 ? m_xz : m_yz);
    const int U = g == 2 ? H : W;
    const float4* wsrc = reinterpret_cast<const float4*>(wT + g * 128 + grp * 32);
    const int koff = (grp & 1) + (grp >> 1) * U;
    const unsigned lane_off = (unsigned)koff * (unsigned)(C * 4) + (unsigned)(sub * 4);  // bytes inside a 2 x 2 cell
    float* const gg = gg_list[g];
    float run[CPL];
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    auto flush = [&](int row) {
      if (row >= 0 && !(dbg & 1)) {
        char* rb = reinterpret_cast<char*>(gg) + (int64_t)row * (C * 4);  // scalar
#pragma unroll
        for (int j = 0; j < CPL; ++j) atomic_add_f32(reinterpret_cast<float*>(rb + lane_off + 64 * j), run[j]);
      }
    };
    if (mask == 1u) {  // all 32 rays in one cell: no run logic at all
#pragma unroll
      for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 w = wsrc[c4];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float4 d = dsrc[j][c4];
          run[j] = fmaf(w.x, d.x, run[j]);
          run[j] = fmaf(w.y, d.y, run[j]);
          run[j] = fmaf(w.z, d.z, run[j]);
          run[j] = fmaf(w.w, d.w, run[j]);
        }
      }
      flush(s_row);
      continue;
    }
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {  // 8 rays at a time
      const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      float dx[CPL][8];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float4 d0v = dsrc[j][2 * c8], d1v = dsrc[j][2 * c8 + 1];
        dx[j][0] = d0v.x; dx[j][1] = d0v.y; dx[j][2] = d0v.z; dx[j][3] = d0v.w;
        dx[j][4] = d1v.x; dx[j][5] = d1v.y; dx[j][6] = d1v.z; dx[j][7] = d1v.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr == 0) {
#pragma unroll
          for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
        } else if ((mask >> rr) & 1u) {
          flush(s_row);
#pragma unroll
          for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
          s_row = __builtin_amdgcn_readlane(row0, rr);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) run[j] = fmaf(w[i], dx[j][i], run[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    flush(s_row);
  }
}

// COLS = false: always the per-slot walk (the MLP-Splatter backward, with a coarse input grid and a large register
// footprint of its own, is 7 % faster with it)
template <int C, int GMS = GM_GENERIC, bool COLS = true>
LP_DEV void scatter_grid(float* gg, const LpGrid& g, int b, float x, float y, float z, bool live, int lane,
                         const float* dxT, float* wT, int dbg) {
  constexpr int CPL = C / 16;  // channels per lane
  if (GMS == GM_TRIPLANE && !(dbg & 8)) {
    scatter_plane<C>(gg, g, b, x, y, z, live, lane, dxT, wT, dbg);
    return;
  }
  if (COLS && GMS != GM_TRIPLANE && (GMS == GM_VOXEL || (g.D > 1 && g.H > 1 && g.W > 1)) && !(dbg & 4)) {
    // voxel grids: the column walk of lp_splat_walk.h (two columns per corner pair, one pass, half the atomics)
    splat_walk_vox<C, 32, SplatSrcLds, false>(gg, nullptr, g, b, x, y, z, live, lane, SplatSrcLds{dxT, DX_LD, lane & 15},
                                              wT, dbg);
    return;
  }
  const int h = lane >> 5, r = lane & 31, sub = lane & 15, grp = lane >> 4;
  TapSet tp;
  grid_tapset<false>(g, b, x, y, z, tp);
  if (!live) {
    tp.ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tp.w[k] = 0.0f;
  }
  // weights -> wT[slot][ray]; the two lanes of a ray write four slots each
#pragma unroll
  for (int i = 0; i < 4; ++i) wT[(4 * h + i) * 32 + r] = h ? tp.w[4 + i] : tp.w[i];
  const int row0 = tp.row0;
  const int ok = (int)tp.ok;
  const int prow = lane_prev(row0);
  const int pok = lane_prev(ok);
  const bool head = run_head(r, row0, prow, ok, pok);
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(head));
  const bool voxel = g.D > 1 && g.H > 1 && g.W > 1;
  const int n_pass = voxel ? 2 : 1;
  for (int p = 0; p < n_pass; ++p) {
    const int k = p * 4 + grp;
    const int koff = (k & 1) * tp.su + ((k >> 1) & 1) * tp.sv + (k >> 2) * tp.st;
    const unsigned kbit = 1u << k;
    const float4* wsrc = reinterpret_cast<const float4*>(wT + k * 32);
    const float4* dsrc[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) dsrc[j] = reinterpret_cast<const float4*>(dxT + (sub + 16 * j) * DX_LD);
    float run[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
    if (mask == 1u) {
      // all 32 rays in one cell (e.g. the plane an image row projects onto as a line): no run logic at all
#pragma unroll
      for (int c4 = 0; c4 < 8; ++c4) {
        const float4 w = wsrc[c4];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          const float4 d = dsrc[j][c4];
          run[j] = fmaf(w.x, d.x, run[j]);
          run[j] = fmaf(w.y, d.y, run[j]);
          run[j] = fmaf(w.z, d.z, run[j]);
          run[j] = fmaf(w.w, d.w, run[j]);
        }
      }
      flush_run<C>(gg, s_row, s_ok, koff, kbit, sub, run, dbg);
      continue;
    }
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {  // 8 rays at a time
      const float4 w0 = wsrc[2 * c8], w1 = wsrc[2 * c8 + 1];
      const float w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
      float dx[CPL][8];
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float4 d0 = dsrc[j][2 * c8], d1 = dsrc[j][2 * c8 + 1];
        dx[j][0] = d0.x; dx[j][1] = d0.y; dx[j][2] = d0.z; dx[j][3] = d0.w;
        dx[j][4] = d1.x; dx[j][5] = d1.y; dx[j][6] = d1.z; dx[j][7] = d1.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask >> rr) & 1u)) {
          flush_run<C>(gg, s_row, s_ok, koff, kbit, sub, run, dbg);
#pragma unroll
          for (int j = 0; j < CPL; ++j) run[j] = 0.0f;
          s_row = __builtin_amdgcn_readlane(row0, rr);
          s_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) run[j] = fmaf(w[i], dx[j][i], run[j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    flush_run<C>(gg, s_row, s_ok, koff, kbit, sub, run, dbg);
  }
}

// host: the grid-list is a canonical triplane (see triplane_taps in lp_device.h)
inline bool is_canonical_triplane(const LpGridList& gl) {
  if (gl.n_grids != 3) return false;
  const LpGrid &xy = gl.grids[0], &xz = gl.grids[1], &yz = gl.grids[2];
  if (xy.D != 1 || xz.H != 1 || yz.W != 1) return false;
  if (xy.H < 2 || xy.W < 2 || xz.D < 2 || xz.W < 2 || yz.D < 2 || yz.H < 2) return false;
  return xy.W == xz.W && xy.H == yz.H && xz.D == yz.D && xy.B == xz.B && xy.B == yz.B;
}

// second-generation backward (lp_renderer_mfma_bwd.hip); gm = GM_* grid-list shape
int renderer_backward_mfma2(const LpRendererArgs& a, const MfmaParams& mp, int gm, hipStream_t stream);
int debug_phase_cycles(unsigned long long* out);  // developer builds with -DLP_PHASE_TIMING, else -1

}  // namespace lp
