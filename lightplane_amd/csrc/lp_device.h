// lp_device.h -- device-side building blocks shared by all kernels (gfx950 only).
//
// Everything here follows the ORACLE semantics (reference naive_renderer.py /
// naive_splatter.py), see SURVEY.md 8(a) "Semantics the HIP kernels must reproduce".
// The translation units are compiled with -ffp-contract=off: every float operation in
// the coordinate / index path is individually rounded (bit-exact integer indexing vs the
// oracle); fused multiply-adds are written explicitly (fmaf) where we want them.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lightplane_hip.h"

#define LP_DEV __device__ __forceinline__

namespace lp {

// ---------------------------------------------------------------------------------------
// ray march schedule
// ---------------------------------------------------------------------------------------

struct Ray {
  float ox, oy, oz, dx, dy, dz, near_t, far_t;
  int b;  // grid batch index
};

LP_DEV Ray load_ray(const LpRays& r, int64_t i) {
  Ray q;
  q.ox = r.origins[3 * i + 0];
  q.oy = r.origins[3 * i + 1];
  q.oz = r.origins[3 * i + 2];
  q.dx = r.directions[3 * i + 0];
  q.dy = r.directions[3 * i + 1];
  q.dz = r.directions[3 * i + 2];
  q.near_t = r.near_t[i];
  q.far_t = r.far_t[i];
  q.b = r.grid_idx[i];
  return q;
}

// linspace(0,1,S)[i], scalar torch formula: i < S/2 ? i*step : 1 - (S-1-i)*step
// LpRays.row_length: which ray a wave's item `idx` (block * rays-per-block + lane order) works on.  The batch is cut into bands of
// four image rows; a band's 4 W rays are walked column by column, alternating direction (a snake): 8 consecutive items = a 2 x 4
// pixel patch whose neighbours in the walk are pixel neighbours (used by the Splatter's backward walk).  A bijection of [0, n_rays): the tail that is not a whole band
// keeps its scanline order.  Measured on reordered inputs before it moved in here: scripts/bench_ray_order.py,
// profiles/r06_ray_order.txt.
LP_DEV int64_t patch_ray_index(int64_t idx, int64_t n_rays, int W) {
  if (W <= 0) return idx;
  const int64_t band_len = 4 * (int64_t)W;
  const int64_t band = idx / band_len;
  if ((band + 1) * band_len > n_rays) return idx;
  const int rem = (int)(idx - band * band_len);
  const int x = rem >> 2, sub = rem & 3;
  const int y = (x & 1) ? 3 - sub : sub;
  return band * band_len + (int64_t)y * W + x;
}

LP_DEV float lin01(int i, int S) {
  if (S <= 1) return 0.0f;
  const float step = 1.0f / (float)(S - 1);
  return (i < S / 2) ? step * (float)i : 1.0f - step * (float)(S - 1 - i);
}

// depth of sample `i` in [0, S + S_inf)  (naive_renderer.py:218-219, 239-247, 810-813)
LP_DEV float sample_depth(int i, const LpMarch& m, float near_t, float far_t) {
  if (i < m.num_samples) {
    return near_t + lin01(i, m.num_samples) * (far_t - near_t);
  }
  // python-double arithmetic of the oracle, then one cast to f32
  const int k = i - m.num_samples;
  const double frac = (double)(k + 1) / (double)m.num_samples_inf;
  const double n_disp = (m.disparity_at_inf - 1.0) * frac + 1.0;
  return far_t * (float)(1.0 / n_disp);
}

// Scale factors of the beyond-far samples, float(1 / n_disp_k) in the oracle's double arithmetic;
// kernels tabulate them once (LDS) so that the per-sample depth is branch-free.
LP_DEV float inf_scale(int k, const LpMarch& m) {
  const double frac = (double)(k + 1) / (double)m.num_samples_inf;
  const double n_disp = (m.disparity_at_inf - 1.0) * frac + 1.0;
  return (float)(1.0 / n_disp);
}
LP_DEV float sample_depth_tab(int i, const LpMarch& m, float near_t, float far_t, const float* inf_tab) {
  const int k = i - m.num_samples;
  const float lin = near_t + lin01(i, m.num_samples) * (far_t - near_t);
  const float inf = far_t * inf_tab[k > 0 ? k : 0];
  return (k < 0) ? lin : inf;
}

// interval length of sample i (naive_renderer.py:252-257); depth_i must be sample_depth(i)
LP_DEV float sample_delta(int i, const LpMarch& m, float near_t, float far_t, float depth_i) {
  if (i == 0) {
    return (m.num_samples > 1) ? (far_t - near_t) / (float)(m.num_samples - 1) : 1.0f;
  }
  return depth_i - sample_depth(i - 1, m, near_t, far_t);
}

// -log T checkpoints: index of the checkpoint taken right after sample i, or -1.
LP_DEV int ckpt_index(int i, const LpMarch& m) {
  if (i < m.num_samples) {
    if ((i + 1) % LP_NLT_CKPT == 0 || i == m.num_samples - 1) return i / LP_NLT_CKPT;
    return -1;
  }
  // beyond-far samples: interval lengths grow up to far/disparity_at_inf, so -log T can jump by
  // orders of magnitude per sample -> checkpoint every one of them
  return (m.num_samples + LP_NLT_CKPT - 1) / LP_NLT_CKPT + (i - m.num_samples);
}
// -log T is accumulated as an unevaluated float pair hi + lo (TwoSum + renormalisation).  The backward
// sweep subtracts the very same products again (far -> near) and so recovers every intermediate value to
// ~2^-45 of the largest one; with a plain float the near samples of a ray whose -log T grows from 0.05 to
// 50 within one checkpoint interval would be reconstructed to ulp(50) only (observed: 3e-4 relative error
// in the encoding gradient of such rays).  Checkpoints store the pair.
LP_DEV void nlt_add(float& hi, float& lo, float x) {
  const float s = hi + x;
  const float bb = s - hi;
  const float err = (hi - (s - bb)) + (x - bb);
  const float l = lo + err;
  const float t = s + l;
  lo = l - (t - s);
  hi = t;
}
// pairs per ray in the checkpoint buffer: the checkpoints + the closing pair (last marched sample, low word)
LP_DEV int ckpt_end_index(const LpMarch& m) {
  return (m.num_samples + LP_NLT_CKPT - 1) / LP_NLT_CKPT + m.num_samples_inf;
}
LP_DEV int ckpt_count(const LpMarch& m) { return ckpt_end_index(m) + 1; }
// blocks of LP_SEG_LEN regular samples = ray segments of the segment-parallel backward (LpRendererArgs.seg_prefix)
LP_DEV int segment_count(const LpMarch& m) { return (m.num_samples + LP_SEG_LEN - 1) / LP_SEG_LEN; }

LP_DEV float contract_one(float p, float n) {
  const float a = fabsf(p);
  if (fabsf(a - n) <= 1e-7f) {
    return (2.0f - 1.0f / a) * (p / a);
  }
  return p / n;
}

// point on the ray, optionally MeRF-contracted (naive_renderer.py:249-250, 796-807)
LP_DEV void sample_point(const Ray& r, float depth, bool contract, float& x, float& y, float& z) {
  x = depth * r.dx + r.ox;
  y = depth * r.dy + r.oy;
  z = depth * r.dz + r.oz;
  if (contract) {
    const float n = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
    if (!(n <= 1.0f)) {
      x = contract_one(x, n);
      y = contract_one(y, n);
      z = contract_one(z, n);
    }
    x = x / 2.0f;
    y = y / 2.0f;
    z = z / 2.0f;
  }
}

LP_DEV bool point_in_bounds(float x, float y, float z) {
  return fabsf(x) <= 1.0f && fabsf(y) <= 1.0f && fabsf(z) <= 1.0f;
}

// ---------------------------------------------------------------------------------------
// grid addressing
// ---------------------------------------------------------------------------------------

// un-normalisation: Renderer = torch grid_sample ((x+1)*size-1)/2 ; Splatter = naive
// splatter (x+1)/2*size - 0.5  
template <bool SPLAT>
LP_DEV float unnormalize(float c, int size) {
  if (SPLAT) return (c + 1.0f) / 2.0f * (float)size - 0.5f;
  return ((c + 1.0f) * (float)size - 1.0f) / 2.0f;
}

struct Axis {
  int lo;      // floor index (may be -1 .. size-1 or further out)
  float w_lo;  // weight of `lo`
  float w_hi;  // weight of `lo + 1`
};

template <bool SPLAT>
LP_DEV Axis axis_setup(float c, int size) {
  const float t = unnormalize<SPLAT>(c, size);
  const float f = floorf(t);
  Axis a;
  // clamp before the int conversion so that far-away / non-finite coordinates stay legal
  a.lo = (int)fminf(fmaxf(f, -2.0f), (float)size);
  a.w_hi = t - f;
  a.w_lo = (f + 1.0f) - t;
  if (!(f >= -2.0f && f <= (float)size)) {  // also catches NaN
    a.w_hi = 0.0f;
    a.w_lo = 0.0f;
  }
  return a;
}

// Corner set of one grid for one point.  K = 8 (voxel) or 4 (plane).  Corner k: bit j of k
// selects the upper neighbour along the j-th *sampled* axis, axes ordered x, y, z.
// row[k] < 0 marks an out-of-range corner (weight irrelevant).
struct Corners {
  int64_t row[8];
  float w[8];
  int n;
};

template <bool SPLAT>
LP_DEV Corners grid_corners(const LpGrid& g, int b, float x, float y, float z) {
  Corners c;
  const bool sx = g.W > 1, sy = g.H > 1, sz = g.D > 1;
  const bool voxel = sx && sy && sz;
  Axis ax{0, 1.0f, 0.0f}, ay{0, 1.0f, 0.0f}, az{0, 1.0f, 0.0f};
  if (sx) ax = axis_setup<SPLAT>(x, g.W);
  if (sy) ay = axis_setup<SPLAT>(y, g.H);
  if (sz) az = axis_setup<SPLAT>(z, g.D);
  const int64_t base = g.row_offset + (int64_t)b * g.D * g.H * g.W;
  c.n = voxel ? 8 : 4;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // map corner bits onto the sampled axes
    int ux, uy, uz;
    if (voxel) {
      ux = k & 1; uy = (k >> 1) & 1; uz = (k >> 2) & 1;
    } else if (!sz) {  // xy plane
      ux = k & 1; uy = (k >> 1) & 1; uz = 0;
    } else if (!sy) {  // xz plane
      ux = k & 1; uy = 0; uz = (k >> 1) & 1;
    } else {           // yz plane
      ux = 0; uy = k & 1; uz = (k >> 1) & 1;
    }
    const int ix = ax.lo + ux, iy = ay.lo + uy, iz = az.lo + uz;
    const bool ok = (k < c.n) && ix >= 0 && ix < g.W && iy >= 0 && iy < g.H && iz >= 0 && iz < g.D;
    float w = (sx ? (ux ? ax.w_hi : ax.w_lo) : 1.0f);
    w = w * (sy ? (uy ? ay.w_hi : ay.w_lo) : 1.0f);
    w = w * (sz ? (uz ? az.w_hi : az.w_lo) : 1.0f);
    c.row[k] = ok ? base + ((int64_t)iz * g.H + iy) * g.W + ix : (int64_t)-1;
    c.w[k] = ok ? w : 0.0f;
  }
  return c;
}

// Slim variant used by the hot kernels: int32 rows (the host guarantees < 2^31 rows), planes
// evaluate 4 taps on their two live axes only.  Arithmetic (and therefore every integer index
// and weight) is identical to grid_corners().
struct Taps {
  int row[8];   // row inside the flat grid-list tensor, -1 = out of range
  float w[8];   // interpolation weight, 0 where out of range
  int n;        // 8 (voxel) or 4 (plane)
};

template <bool SPLAT>
LP_DEV void axis_taps(float c, int size, int& i0, float (&w)[2], bool (&ok)[2]) {
  const float t = unnormalize<SPLAT>(c, size);
  const float f = floorf(t);
  i0 = (int)fminf(fmaxf(f, -2.0f), (float)size);  // NaN -> -2: both taps invalid
  w[1] = t - f;
  w[0] = (f + 1.0f) - t;
  ok[0] = (unsigned)i0 < (unsigned)size;
  ok[1] = (unsigned)(i0 + 1) < (unsigned)size;
}

template <bool SPLAT>
LP_DEV void voxel_taps(const LpGrid& g, int b, float x, float y, float z, Taps& t) {
  const int base = (int)g.row_offset + b * (g.D * g.H * g.W);
  int ix, iy, iz;
  float wx[2], wy[2], wz[2];
  bool okx[2], oky[2], okz[2];
  axis_taps<SPLAT>(x, g.W, ix, wx, okx);
  axis_taps<SPLAT>(y, g.H, iy, wy, oky);
  axis_taps<SPLAT>(z, g.D, iz, wz, okz);
  t.n = 8;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int ux = k & 1, uy = (k >> 1) & 1, uz = (k >> 2) & 1;
    const bool ok = okx[ux] && oky[uy] && okz[uz];
    t.row[k] = ok ? base + ((iz + uz) * g.H + (iy + uy)) * g.W + (ix + ux) : -1;
    t.w[k] = ok ? (wx[ux] * wy[uy]) * wz[uz] : 0.0f;
  }
}

// plane: (u, v) = (x, y) | (x, z) | (y, z); row = iv * U + iu in all three cases (branch-free)
template <bool SPLAT>
LP_DEV void plane_taps(const LpGrid& g, int b, float x, float y, float z, Taps& t) {
  const int base = (int)g.row_offset + b * (g.D * g.H * g.W);
  const bool xy = g.D == 1, xz = g.H == 1;
  const float cu = (xy || xz) ? x : y;
  const float cv = xy ? y : z;
  const int U = (xy || xz) ? g.W : g.H;
  const int V = xy ? g.H : g.D;
  int iu, iv;
  float wu[2], wv[2];
  bool oku[2], okv[2];
  axis_taps<SPLAT>(cu, U, iu, wu, oku);
  axis_taps<SPLAT>(cv, V, iv, wv, okv);
  t.n = 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int uu = k & 1, uv = (k >> 1) & 1;
    const bool ok = oku[uu] && okv[uv];
    t.row[k] = ok ? base + (iv + uv) * U + (iu + uu) : -1;
    t.w[k] = ok ? wu[uu] * wv[uv] : 0.0f;
  }
#pragma unroll
  for (int k = 4; k < 8; ++k) { t.row[k] = -1; t.w[k] = 0.0f; }
}

// Canonical triplane (grids[0] = xy plane [B,1,H,W], grids[1] = xz plane [B,D,1,W], grids[2] = yz plane [B,D,H,1] with
// ONE size per axis -- what `is_canonical_triplane` checks on the host): the three planes share three axis
// computations instead of doing six, and nothing about the plane orientation is decided at run time.
struct AxisTap {
  int i0;
  float w[2];
  bool ok[2];
};
template <bool SPLAT>
LP_DEV void triplane_axes(const LpGrid* g, float x, float y, float z, AxisTap& ax, AxisTap& ay, AxisTap& az) {
  axis_taps<SPLAT>(x, g[0].W, ax.i0, ax.w, ax.ok);
  axis_taps<SPLAT>(y, g[0].H, ay.i0, ay.w, ay.ok);
  axis_taps<SPLAT>(z, g[1].D, az.i0, az.w, az.ok);
}
LP_DEV void plane_taps_from_axes(int base, int U, const AxisTap& u, const AxisTap& v, Taps& t) {
  t.n = 4;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int uu = k & 1, uv = (k >> 1) & 1;
    const bool ok = u.ok[uu] && v.ok[uv];
    t.row[k] = ok ? base + (v.i0 + uv) * U + (u.i0 + uu) : -1;
    t.w[k] = ok ? u.w[uu] * v.w[uv] : 0.0f;
  }
#pragma unroll
  for (int k = 4; k < 8; ++k) { t.row[k] = -1; t.w[k] = 0.0f; }
}
template <bool SPLAT>
LP_DEV void triplane_taps(const LpGrid* g, int b, float x, float y, float z, Taps (&t)[3]) {
  AxisTap ax, ay, az;
  triplane_axes<SPLAT>(g, x, y, z, ax, ay, az);
  const int W = g[0].W, H = g[0].H, D = g[1].D;
  plane_taps_from_axes((int)g[0].row_offset + b * (H * W), W, ax, ay, t[0]);  // xy: u = x, v = y
  plane_taps_from_axes((int)g[1].row_offset + b * (D * W), W, ax, az, t[1]);  // xz: u = x, v = z
  plane_taps_from_axes((int)g[2].row_offset + b * (D * H), H, ay, az, t[2]);  // yz: u = y, v = z
}

template <bool SPLAT>
LP_DEV void grid_taps(const LpGrid& g, int b, float x, float y, float z, Taps& t) {
  if (g.D > 1 && g.H > 1 && g.W > 1) voxel_taps<SPLAT>(g, b, x, y, z, t);
  else plane_taps<SPLAT>(g, b, x, y, z, t);
}

// Tap set in "origin + strides" form (used by the run-merging gradient scatter): slot k adds
// (k&1)*su + ((k>>1)&1)*sv + (k>>2)*st rows to the VIRTUAL row of slot 0 (which may lie outside
// the grid; `ok` bit k says whether slot k is in range).  Weights / indices are computed by the
// same arithmetic as voxel_taps() / plane_taps(), i.e. they are bit-identical to them.
struct TapSet {
  int row0;
  int iu;       // cell index along the fastest axis (x / u), -1 .. size-1
  int cell;     // voxel grids: (ix+1) | (iy+1) << 10 | (iz+1) << 20 (axis sizes <= 1022), else 0
  int su, sv, st;
  float w[8];   // 0 where out of range
  unsigned ok;  // bit k set <=> slot k in range
  int n;        // 4 (plane) or 8 (voxel)
};

template <bool SPLAT>
LP_DEV void grid_tapset(const LpGrid& g, int b, float x, float y, float z, TapSet& t) {
  const int base = (int)g.row_offset + b * (g.D * g.H * g.W);
  if (g.D > 1 && g.H > 1 && g.W > 1) {
    int ix, iy, iz;
    float wx[2], wy[2], wz[2];
    bool okx[2], oky[2], okz[2];
    axis_taps<SPLAT>(x, g.W, ix, wx, okx);
    axis_taps<SPLAT>(y, g.H, iy, wy, oky);
    axis_taps<SPLAT>(z, g.D, iz, wz, okz);
    t.n = 8;
    t.su = 1; t.sv = g.W; t.st = g.H * g.W;
    t.row0 = base + (iz * g.H + iy) * g.W + ix;
    t.iu = ix;
    t.cell = (ix + 1) | ((iy + 1) << 10) | ((iz + 1) << 20);
    t.ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int ux = k & 1, uy = (k >> 1) & 1, uz = (k >> 2) & 1;
      const bool ok = okx[ux] && oky[uy] && okz[uz];
      t.ok |= ok ? (1u << k) : 0u;
      t.w[k] = ok ? (wx[ux] * wy[uy]) * wz[uz] : 0.0f;
    }
  } else {
    const bool xy = g.D == 1, xz = g.H == 1;
    const float cu = (xy || xz) ? x : y;
    const float cv = xy ? y : z;
    const int U = (xy || xz) ? g.W : g.H;
    const int V = xy ? g.H : g.D;
    int iu, iv;
    float wu[2], wv[2];
    bool oku[2], okv[2];
    axis_taps<SPLAT>(cu, U, iu, wu, oku);
    axis_taps<SPLAT>(cv, V, iv, wv, okv);
    t.n = 4;
    t.su = 1; t.sv = U; t.st = 0;
    t.row0 = base + iv * U + iu;
    t.iu = iu;
    t.cell = 0;
    t.ok = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int uu = k & 1, uv = (k >> 1) & 1;
      const bool ok = oku[uu] && okv[uv];
      t.ok |= ok ? (1u << k) : 0u;
      t.w[k] = ok ? wu[uu] * wv[uv] : 0.0f;
    }
#pragma unroll
    for (int k = 4; k < 8; ++k) t.w[k] = 0.0f;
  }
}

// nearest-neighbour scaffold lookup (round-half-even like F.grid_sample(mode="nearest")),
// zero outside the grid, times the in-bounds mask (naive_renderer.py:484-499).
LP_DEV float scaffold_lookup(const float* scaffold, const LpGrid& s, int b, float x, float y, float z) {
  if (!point_in_bounds(x, y, z)) return 0.0f;
  const float fx = rintf(unnormalize<false>(x, s.W));
  const float fy = rintf(unnormalize<false>(y, s.H));
  const float fz = rintf(unnormalize<false>(z, s.D));
  if (!(fx >= 0.0f && fx <= (float)(s.W - 1) && fy >= 0.0f && fy <= (float)(s.H - 1) && fz >= 0.0f &&
        fz <= (float)(s.D - 1)))
    return 0.0f;
  const int64_t row = (((int64_t)b * s.D + (int)fz) * s.H + (int)fy) * s.W + (int)fx;
  return scaffold[row];
}

// ---------------------------------------------------------------------------------------
// activations
// ---------------------------------------------------------------------------------------

// softplus (torch: beta 1, linear above 20).  log1p(e) through the hardware log for e >= 2^-6 and
// through its series below (|error| < 3e-8 there): a handful of VALU ops instead of libm's log1pf.
#if defined(LP_X_RELU_F) || defined(LP_X_MASK_BIT_INT) || defined(LP_X_MASK_APPLY)
#ifndef LP_EXPERIMENTS
#error "LP_X_* are A/B timing switches: build them with -DLP_EXPERIMENTS (LP_BUILD_FLAGS), which makes lp_version() negative so that the binding refuses the library unless LIGHTPLANE_AMD_ALLOW_EXPERIMENTAL=1"
#endif
#endif
LP_DEV float softplus_f(float x) {
  const float e = __expf(x);
  const float series = e * fmaf(e, fmaf(e, 0.333333343f, -0.5f), 1.0f);
  const float lg = __logf(1.0f + e);
  const float sp = (e < 0.015625f) ? series : lg;
  return (x > 20.0f) ? x : sp;
}
// ReLU on the integer unit.  fmaxf(x, 0) costs TWO instructions here: in IEEE mode v_max_f32 has to be preceded by a
// canonicalising v_max_f32 x, x, x whenever the compiler cannot prove x is not a signalling NaN (MFMA results never qualify).
// The sign-magnitude encoding makes max over the bit patterns the same function: negative floats are negative integers,
// -0 -> +0, positive values and +NaN pass through (one v_max_i32).
// NaN contract of every MFMA family (all their ReLU sites use this helper): a NaN with the sign bit clear propagates (as in
// torch.relu), a NaN with the sign bit set becomes 0 (torch.relu would propagate it); the mask bit of a NaN activation is
// clear (`relu_value > 0` is false), so no gradient flows through it.  Non-finite activations are outside the parity contract
// -- the reference asserts isfinite on its gradients after every backward (lightplane_renderer.py:713-722) -- the kernels only
// promise not to fault on them.  The shape-generic kernels use fmaxf (both NaNs -> 0).
LP_DEV float relu_f(float x) {
#ifdef LP_X_RELU_F
  return fmaxf(x, 0.0f);
#else
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
#endif
}

// ReLU masks as bits of one register (the backward keeps 16 activations' masks in 16 bits instead of 16 registers).
// mask_bit: post-ReLU value -> bit q.  mask_apply: v if bit q is set else 0: v_bfe_i32 (sign-extends the bit to all ones) +
// v_and_b32 instead of and + compare + select.
LP_DEV unsigned mask_bit(unsigned m, float relu_value, int q) {
#ifdef LP_X_MASK_BIT_INT  // two integer ops (v_min_i32, v_lshl_or_b32) -- but with it the allocator of the dominant backward
  // kernel spills 19 values inside the sample loop instead of 2 (scripts/isa_loop_scratch.py): off
  const int i = __builtin_bit_cast(int, relu_value);
  return m | ((unsigned)(i < 1 ? i : 1) << q);
#else
  return m | ((relu_value > 0.0f) ? (1u << q) : 0u);
#endif
}
LP_DEV float mask_apply(unsigned m, int q, float v) {
#ifdef LP_X_MASK_APPLY
  return (m & (1u << q)) ? v : 0.0f;
#else
  const int all = ((int)(m << (31 - q))) >> 31;
  return __builtin_bit_cast(float, __builtin_bit_cast(int, v) & all);
#endif
}

// 1 / (1 + e^-x) with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division (ten instructions):
// the forward evaluates it 3-4 times per ray and sample, the backward once more for d softplus
LP_DEV float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// d softplus / dx (torch: threshold 20 -> 1)
LP_DEV float d_softplus_f(float x) { return (x > 20.0f) ? 1.0f : sigmoid_f(x); }

// ---------------------------------------------------------------------------------------
// hash RNG (reference triton_src/shared/rand_util.py:39-80, 110-145): int32 wrap-around
// ---------------------------------------------------------------------------------------

LP_DEV int32_t hash32(int32_t x) {
  x = (int32_t)((uint32_t)((x >> 16) ^ x) * 0x45D9F3Bu);
  x = (int32_t)((uint32_t)((x >> 16) ^ x) * 0x45D9F3Bu);
  return (x >> 16) ^ x;
}
LP_DEV int32_t pair_hash(int32_t x, int32_t h) {
  const uint32_t u = (uint32_t)(h ^ x);
  return (int32_t)((u << 24) + u * 0x193u);
}
LP_DEV float int32_to_u01(int32_t h) {
  return (((float)h + 2147483648.0f) + 3.0f) / 4294967296.0f;
}
LP_DEV float hash_randn(int32_t x1, int32_t x2, int32_t seed) {
  const int32_t prime = 105097564;
  const int32_t h1 = pair_hash(pair_hash(prime, seed), hash32(x1));
  const int32_t h2 = pair_hash(pair_hash(prime, seed + 1), hash32(x2));
  const float u1 = int32_to_u01(h1), u2 = int32_to_u01(h2);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530718f * u2);
}
// noise of (ray, step): i1 = ray*S_tot + step + 1, i2 = i1 + max(N,16)*S_tot (naive_renderer.py:779-793)
LP_DEV float sample_noise(int64_t ray, int step, int64_t n_rays, int s_tot, int32_t seed) {
  const int64_t i1 = ray * s_tot + step + 1;
  const int64_t pad = n_rays > 16 ? n_rays : 16;
  const int64_t i2 = i1 + pad * s_tot;
  return hash_randn((int32_t)(uint32_t)(uint64_t)i1, (int32_t)(uint32_t)(uint64_t)i2, seed);
}

// ---------------------------------------------------------------------------------------
// fused module epilogue (LpRendererArgs.bg_color / alpha; reference renderer_module.py:552-561)
// ---------------------------------------------------------------------------------------
// forward: per-ray outputs.  feature + T * bg and 1 - T are formed like the reference's PyTorch ops (one rounding per
// operation, accurate expf).
LP_DEV void write_ray_outputs(const LpRendererArgs& a, int64_t ray_id, float len, float nlt, const float* facc) {
  a.ray_length[ray_id] = len;
  a.neg_log_t[ray_id] = nlt;
  const bool epi = a.bg_color != nullptr || a.alpha != nullptr;
  const float T = epi ? expf(-nlt) : 0.0f;
  for (int c = 0; c < a.color_chn; ++c) {
    float f = facc[c];
    if (a.bg_color) f = f + T * a.bg_color[c];
    a.feature[ray_id * a.color_chn + c] = f;
  }
  if (a.alpha) a.alpha[ray_id] = (a.alpha_mode == 2) ? -nlt : 1.0f - T;
}
// backward: gradient of the loss w.r.t. the final -log T, with the epilogue's terms folded in.  gfeat = upstream
// gradient of the (composited) feature, which is also the gradient of the rendered feature.
LP_DEV float epilogue_grad_nlt(const LpRendererArgs& a, int64_t rid, bool valid, float nlt_final, float g_nlt,
                               const float* gfeat, int n_chn) {
  if (a.bg_color == nullptr && a.grad_alpha == nullptr) return g_nlt;
  const float T = expf(-nlt_final);
  if (a.grad_alpha && valid) {
    const float ga = a.grad_alpha[rid];
    g_nlt += (a.alpha_mode == 2) ? -ga : ga * T;
  }
  if (a.bg_color) {
    float s = 0.0f;
    // bg_color holds color_chn floats; the MFMA kernels call this with their 4-wide colour path (gfeat[c >= color_chn] = 0,
    // but 0 * NaN from a float behind the tensor would poison the ray)
    for (int c = 0; c < n_chn && c < a.color_chn; ++c) s = fmaf(a.bg_color[c], gfeat[c], s);
    g_nlt -= T * s;
  }
  return g_nlt;
}

// ---------------------------------------------------------------------------------------
// run heads of the scatter walks
// ---------------------------------------------------------------------------------------
// Value of lane - 1 (lane 0: its own).  Cross-lane reads must NOT sit behind a short-circuit (`(r == 0) || x !=
// __shfl_up(x, 1)`): the compiler turns the `||` into a branch, the shuffle then runs with lane 0 (and every other lane
// that took the first alternative) disabled, and ds_bpermute returns 0 for a disabled SOURCE lane -- lane 1 compared its
// row with 0 instead of lane 0's row, so a run starting at row 0 right behind a dead first ray (row -1) was merged into
// the dead run and never flushed (found by tests/test_gpu_coherent.py, reproduced by scripts/scatter_plane_test.hip).
// These helpers read first, with all lanes enabled, and combine afterwards.
// (DPP wave_shr:1 -- one VALU instruction -- instead of __shfl_up's ds_bpermute round trip through the LDS crossbar; lane 0 and
// any lane whose source is disabled get 0, which run_head() never looks at for r == 0.)
LP_DEV int lane_prev(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, false); }
LP_DEV bool run_head(int r, int row, int prev_row) { return (r == 0) | (row != prev_row); }
LP_DEV bool run_head(int r, int row, int prev_row, int ok, int prev_ok) {
  return (r == 0) | (row != prev_row) | (ok != prev_ok);
}

// ---------------------------------------------------------------------------------------
// atomics: hardware fp32 add, result unused (global_atomic_add_f32, no CAS loop)
// ---------------------------------------------------------------------------------------
LP_DEV void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

}  // namespace lp
