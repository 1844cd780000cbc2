// lp_splat_walk.h -- the Splatter's run-merged scatter walk on voxel grids, shared by the plain Splatter
// (lp_splatter.hip: the splatted vector sits in registers) and the MFMA MLP-Splatter (lp_splatter_mlp_loop.h:
// in an LDS tile).
#pragma once
#include "lp_device.h"

namespace lp {

// value sources of splat_walk_vox: load8(j, c8, out) = channels (lane & 15) + 16 j of rays 8 c8 .. 8 c8 + 7
template <int CPL, int RPW>
struct SplatSrcRegs {
  const float (&enc)[CPL][RPW];
  LP_DEV void load8(int j, int c8, float (&out)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = enc[j][8 * c8 + i];
  }
};
template <int CPL>
struct SplatSrcConst {  // every item of the walk splats the same vector (the samples of ONE ray: transposed march)
  float v[CPL];
  LP_DEV void load8(int j, int, float (&out)[8]) const {
#pragma unroll
    for (int i = 0; i < 8; ++i) out[i] = v[j];
  }
};
struct SplatSrcLds {  // tile [channel][ld] in LDS
  const float* tile;
  int ld, sub;
  LP_DEV void load8(int j, int c8, float (&out)[8]) const {
    const float4* p = reinterpret_cast<const float4*>(tile + (sub + 16 * j) * ld);
    const float4 d0 = p[2 * c8], d1 = p[2 * c8 + 1];
    out[0] = d0.x; out[1] = d0.y; out[2] = d0.z; out[3] = d0.w;
    out[4] = d1.x; out[5] = d1.y; out[6] = d1.z; out[7] = d1.w;
  }
};

// Unit weights of a voxel walk: x-neighbouring rows are neighbouring floats of the weight grid, so a 16-cell window of x per
// (y, z) pair is kept in the lanes (lane = pair * 16 + x - window base) and written with ONE atomic
// instruction of four 64-byte segments when the walk leaves it -- the per-run version issued four segments
// per run (about eight runs per 16 rays).  row0 / iu: per-lane values of the wave's rays (lane r = ray r), mask: run heads.
template <int RPW, int WLD>
LP_DEV void splat_walk_vox_weights(float* wgt, const LpGrid& g, int row0, int iu, int sv_, int st_, unsigned mask, int lane,
                                   const float* wT, int dbg) {
  const int sub = lane & 15, grp = lane >> 4;
  struct { int sv, st; } tp = {sv_, st_};
  {
    const int koff = (grp & 1) * tp.sv + (grp >> 1) * tp.st;
    const float4* wlo = reinterpret_cast<const float4*>(wT + (2 * grp) * WLD);
    const float4* whi = reinterpret_cast<const float4*>(wT + (2 * grp + 1) * WLD);
    const int W = g.W;
    float acc = 0.0f;
    // (an opaque copy of the head mask: with the same value visibly tested in both walks the compiler keeps all 32 head bits of the
    // first walk as lane masks in 64 scalar registers and tests each bit twice there)
    unsigned mask_w = (unsigned)__builtin_amdgcn_readfirstlane((int)mask);
    asm volatile("; head mask of the weight walk" : "+s"(mask_w));
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    int s_iu = __builtin_amdgcn_readlane(iu, 0);
    int wb = ((s_iu & 15) == 15) ? s_iu : (s_iu & ~15);  // window base (x of lane sub == 0)
    int rowb = s_row - s_iu;                              // row of x = 0 in the (y0, z0) line of the window
    bool m0 = (wb + sub) == s_iu, m1 = (wb + sub) == s_iu + 1;
    const bool on = !(dbg & 2);
#pragma unroll
    for (int c8 = 0; c8 < RPW / 8; ++c8) {
      const float4 a0 = wlo[2 * c8], a1 = wlo[2 * c8 + 1], b0 = whi[2 * c8], b1 = whi[2 * c8 + 1];
      const float w0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float w1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask_w >> rr) & 1u)) {
          const int n_row = __builtin_amdgcn_readlane(row0, rr);
          const int n_iu = __builtin_amdgcn_readlane(iu, rr);
          // (same (y, z) line and 0 <= n_iu - wb <= 14: nothing to do; one opaque integer per case, as above)
          const int dl = (n_row - n_iu) - rowb;                       // how the window's (y0, z0) line moves
          const int out_x = (int)((unsigned)(n_iu - wb) >= 15u);
          int leave = dl | out_x;
          // a step of one line along y or z inside the window's x range (round 6: views oblique to the grid take it at most run heads):
          // the two pairs on the far side become the near ones -- only the two left behind are written (two segments instead of four)
          int carry = out_x | (int)!(dl == tp.sv || dl == -tp.sv || dl == tp.st || dl == -tp.st);
          leave = __builtin_amdgcn_readfirstlane(leave);
          carry = __builtin_amdgcn_readfirstlane(carry);
          asm volatile("" : "+s"(leave), "+s"(carry));
          if (leave != 0) {
            const int xl = wb + sub;
            if (carry == 0) {
              const bool along_y = dl == tp.sv || dl == -tp.sv;
              const int side = along_y ? (grp & 1) : (grp >> 1);      // this lane's corner bit along the stepping axis
              const bool behind = side == (dl > 0 ? 0 : 1);
              if (behind && acc != 0.0f && xl >= 0 && xl < W && on) atomic_add_f32(wgt + (int64_t)(rowb + koff + xl), acc);
              const float other = __shfl_xor(acc, along_y ? 16 : 32);
              acc = behind ? other : 0.0f;                            // (the lanes left behind take over the line that stays)
              rowb += dl;
            } else {
              if (acc != 0.0f && xl >= 0 && xl < W && on) atomic_add_f32(wgt + (int64_t)(rowb + koff + xl), acc);
              acc = 0.0f;
              wb = ((n_iu & 15) == 15) ? n_iu : (n_iu & ~15);
              rowb = n_row - n_iu;
            }
          }
          m0 = (wb + sub) == n_iu;
          m1 = (wb + sub) == n_iu + 1;
        }
        acc += m0 ? w0[i] : (m1 ? w1[i] : 0.0f);
      }
    }
    const int xl = wb + sub;
    if (acc != 0.0f && xl >= 0 && xl < W && on) atomic_add_f32(wgt + (int64_t)(rowb + koff + xl), acc);
  }
}

// Voxel grids: the walk of splat_walk with the two columns of a cell along one axis kept in separate accumulators.
// Lane group grp = corner pair over the two other axes, lane = channel.  Image-adjacent rays mostly step from a
// cell to its neighbour along one grid axis (which one depends on the camera; it is read off the first cell
// change of the walk): the far column of the old cell is the near column of the new one, so it stays in
// registers and only the column that is left behind is flushed -- half the atomic segments of the per-slot
// walk (the Splatter forward is bound by the rate of 64-byte atomic segments), and one pass
// instead of two.
// SPLAT: interpolation convention (true = Splatter, false = Renderer: the gradient scatter of its backward);
// wgt == nullptr: no unit-weight grid (Renderer).
// WLD: row stride (floats) of the weight table wT[8][WLD].  With WLD = RPW = 32 the four corner pairs a ds_read_b128 lane group
// serves read rows 64 floats apart -- the same banks (64 x 4 B) at different addresses: a 2-way conflict on every weight read
// (47 % of the forward walk's LDS cycles, profiles/r04_pmc_summary.json).  WLD = RPW + 8 puts the rows of any corner-pair set
// (k_lo in {0,2,4,6}, {0,1,4,5}, {0,1,2,3}) on distinct 16-byte bank slots; callers whose table has the room pass it.
template <int C, int RPW, class Src, bool SPLAT = true, int WLD = RPW>
LP_DEV void splat_walk_vox(float* feat, float* wgt, const LpGrid& g, int b, float x, float y, float z, bool live,
                           int lane, const Src& src, float* wT, int dbg) {
  constexpr int CPL = C / 16;
  constexpr int NQ = 64 / RPW;
  constexpr int SPQ = 8 / NQ;
  const int q = lane / RPW, r = lane % RPW, sub = lane & 15, grp = lane >> 4;
  TapSet tp;
  grid_tapset<SPLAT>(g, b, x, y, z, tp);
  if (!live) {
    tp.ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tp.w[k] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < SPQ; ++i) {
    float v = tp.w[i];
#pragma unroll
    for (int qq = 1; qq < NQ; ++qq) v = (q == qq) ? tp.w[qq * SPQ + i] : v;
    wT[(q * SPQ + i) * WLD + r] = v;
  }
  const int row0 = tp.row0, iu = tp.iu, cell = tp.cell;
  const int ok = (int)tp.ok;
  const int prow_ = lane_prev(row0), pok_ = lane_prev(ok);  // all lanes enabled: see run_head()
  const bool head = run_head(r, row0, prow_, ok, pok_);
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(head));
  if (!(dbg & 64)) {
    // merge axis A (wave-uniform): the axis of the first cell change of this walk
    int A = 0;
    if (mask & ~1u) {
      const int r1 = __builtin_ctz(mask & ~1u);
      const int dc = __builtin_amdgcn_readlane(cell, r1) - __builtin_amdgcn_readlane(cell, 0);
      A = (dc == (1 << 10) || dc == -(1 << 10)) ? 1 : ((dc == (1 << 20) || dc == -(1 << 20)) ? 2 : 0);
    }
    const bool packable = g.W <= 1022 && g.H <= 1022 && g.D <= 1022;
    const int step = packable ? (1 << (10 * A)) : 0x40000000;   // cell-code step of +1 along A
    const int sA = A == 0 ? tp.su : (A == 1 ? tp.sv : tp.st);   // row stride along A
    const int s0 = A == 0 ? tp.sv : tp.su, s1 = A == 2 ? tp.sv : tp.st;  // row strides of the two other axes
    const int b0 = grp & 1, b1 = grp >> 1;
    const int k_lo = A == 0 ? 2 * b0 + 4 * b1 : (A == 1 ? b0 + 4 * b1 : b0 + 2 * b1);
    const int k_hi = k_lo + (1 << A);
    const int koff = b0 * s0 + b1 * s1;  // rows of this corner pair relative to row0
    const unsigned bit_lo = 1u << k_lo, bit_hi = 1u << k_hi;
    const float4* wlo = reinterpret_cast<const float4*>(wT + k_lo * WLD);
    const float4* whi = reinterpret_cast<const float4*>(wT + k_hi * WLD);
    const int64_t hi_off = (int64_t)sA * C;
    const unsigned lane_off = (unsigned)(koff * C + sub);  // this lane's float inside the block of rows that starts at row0
    // slots on the near / far side along A (all four corner pairs): a column is only carried over if the old
    // and the new cell agree on its validity (a masked or padding ray in the neighbouring cell has ok == 0)
    const unsigned lo_slots = A == 0 ? 0x55u : (A == 1 ? 0x33u : 0x0Fu);
    const int sh = 1 << A;
    float lo[CPL], hi[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) lo[j] = hi[j] = 0.0f;
    int s_row = __builtin_amdgcn_readlane(row0, 0);
    int s_cell = __builtin_amdgcn_readlane(cell, 0);
    unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
    const bool on = !(dbg & 1);
#pragma unroll
    for (int c8 = 0; c8 < RPW / 8; ++c8) {
      const float4 a0 = wlo[2 * c8], a1 = wlo[2 * c8 + 1], b0 = whi[2 * c8], b1 = whi[2 * c8 + 1];
      const float w0[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float w1[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      float ev[CPL][8];  // the splatted vector: channels sub + 16 j of rays 8 c8 ..
#pragma unroll
      for (int j = 0; j < CPL; ++j) src.load8(j, c8, ev[j]);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int rr = 8 * c8 + i;
        if (rr > 0 && ((mask >> rr) & 1u)) {
          const int n_row = __builtin_amdgcn_readlane(row0, rr);
          const int n_cell = __builtin_amdgcn_readlane(cell, rr);
          const int d = n_cell - s_cell, dr = n_row - s_row;  // dr: also tells grids of different batch entries apart
          const unsigned n_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
          // up == 0: +1 along A, the far column becomes the near one; down == 0: -1 along A.  A column is only carried over if the old and
          // the new cell agree on its validity.  (One integer per case, kept opaque: a conjunction of wave-uniform compares is lowered to
          // lane-mask logic -- s_cselect_b64 / s_and_b64 vcc / s_cbranch_vccnz, three times the scalar instructions of s_cmp + s_cbranch_scc;
          // the walks are bound by their instruction count, profiles/r06_splat_launch_order.txt.)
          int up = (d ^ step) | (dr ^ sA) | (int)(((n_ok & lo_slots) << sh) ^ (s_ok & (lo_slots << sh)));
          int down = (d ^ -step) | (dr ^ -sA) | (int)(((s_ok & lo_slots) << sh) ^ (n_ok & (lo_slots << sh)));
          up = __builtin_amdgcn_readfirstlane(up);
          down = __builtin_amdgcn_readfirstlane(down);
          asm volatile("" : "+s"(up), "+s"(down));
          float* const rowp = feat + (int64_t)s_row * C;  // (wave-uniform: scalar base + per-lane 32-bit offset)
          if (down != 0) {  // the near column is left behind
            if ((s_ok & bit_lo) && on) {
#pragma unroll
              for (int j = 0; j < CPL; ++j) atomic_add_f32(rowp + lane_off + 16 * j, lo[j]);
            }
          }
          if (up != 0) {  // the far column is left behind
            if ((s_ok & bit_hi) && on) {
#pragma unroll
              for (int j = 0; j < CPL; ++j) atomic_add_f32(rowp + hi_off + lane_off + 16 * j, hi[j]);
            }
          }
          const bool is_up = up == 0, is_down = down == 0;
#pragma unroll
          for (int j = 0; j < CPL; ++j) {
            const float l = lo[j], h = hi[j];
            lo[j] = is_up ? h : 0.0f;
            hi[j] = is_down ? l : 0.0f;
          }
          s_row = n_row;
          s_cell = n_cell;
          s_ok = n_ok;
        }
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
          lo[j] = fmaf(w0[i], ev[j][i], lo[j]);
          hi[j] = fmaf(w1[i], ev[j][i], hi[j]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float* const rowp = feat + (int64_t)s_row * C;
    if ((s_ok & bit_lo) && on) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) atomic_add_f32(rowp + lane_off + 16 * j, lo[j]);
    }
    if ((s_ok & bit_hi) && on) {
#pragma unroll
      for (int j = 0; j < CPL; ++j) atomic_add_f32(rowp + hi_off + lane_off + 16 * j, hi[j]);
    }
  }
  if (!SPLAT || (dbg & 32)) return;  // the Renderer's gradient scatter has no weight grid  (dbg & 32 / 64: timing experiments)
  splat_walk_vox_weights<RPW, WLD>(wgt, g, row0, iu, tp.sv, tp.st, mask, lane, wT, dbg);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Two carry axes (round 6).  The walk above keeps the two columns of a cell along ONE axis and carries the far one over a +-1
// step along it; any other step flushes both.  Camera rows that are oblique to the grid step along two axes in turn: cfg 5's
// thirteen views take the "other" branch at 42 % of their run heads (cfg 3's single axis-aligned view at 5 %) and write 2.6
// atomic segments per ray-sample instead of 1.7.  Here a lane keeps all FOUR columns of the cell over two axes A, B (the two
// axes the wave's run heads step along most often, read off the cells of the heads), lane group = corner along the third axis,
// lane = channel of 32: a +-1 step along A or B flushes the two columns left behind, a diagonal step three, and what stays is
// re-labelled.  Same segments per flushed column as before (two rows of C x 4 bytes per atomic instruction).
// enc[j][i]: channel (lane & 31) + 32 j of ray i.  Splatter forward only (SPLAT convention, unit weights by the caller).
// DIAG: also diagonal steps over the two carry axes, as two successive unit steps through a virtual cell (three of four columns leave).
// Pays for rays marched along (the transposed march: -6 % on the reference's splatter benchmark), not for image rows (cfg 5 -1 %, cfg 3 + 3 %:
// the second pass through the step code costs every run head a loop).
template <int C, int RPW, int WLD, class Enc, bool DIAG = false>
LP_DEV void splat_walk_vox_feat2(float* feat, const LpGrid& g, int row0, int cell, int ok, int su, int sv, int st, unsigned mask,
                                 int lane, const Enc& enc, const float* wT, int dbg) {
  static_assert(C % 32 == 0, "32 channels per lane group");
  constexpr int CPL = C / 32;
  const int ch = lane & 31, gc = lane >> 5;
  const bool packable = g.W <= 1022 && g.H <= 1022 && g.D <= 1022;
  const int pcell = lane_prev(cell), prow = lane_prev(row0);
  const unsigned pok = (unsigned)lane_prev(ok);
  // the axis the heads change least along = the corner axis of the lane groups
  int A = 0, B = 1, X = 2;
  {
    const int xr = cell ^ pcell;
    const bool hd = lane < RPW && lane > 0 && ((mask >> (lane & 31)) & 1u);
    const int n0 = __builtin_popcountll(__ballot(hd && (xr & 0x3FF) != 0));
    const int n1 = __builtin_popcountll(__ballot(hd && ((xr >> 10) & 0x3FF) != 0));
    const int n2 = __builtin_popcountll(__ballot(hd && ((xr >> 20) & 0x3FF) != 0));
    if (n0 <= n1 && n0 <= n2) { X = 0; A = 1; B = 2; }
    else if (n1 <= n2) { X = 1; A = 0; B = 2; }
  }
  const int sA = A == 0 ? su : sv;                 // (A < B: A is x or y, B is y or z)
  const int sB = B == 1 ? sv : st;
  const int sX = X == 0 ? su : (X == 1 ? sv : st);
  const int shA = 10 * A, shB = 10 * B, shX = 10 * X;
  const unsigned mA0 = A == 0 ? 0x55u : 0x33u, mB0 = B == 1 ? 0x33u : 0x0Fu;   // slots with the A / B bit clear
  const int bA = 1 << A, bB = 1 << B;                                          // slot-index step along A / B
  // How every ray's cell lies to its predecessor's, for all rays at once (lane = ray; a run head's predecessor is the last ray of the
  // previous run, i.e. the previous run's cell): 0 / 1 = +1 / -1 along A, 2 / 3 = +1 / -1 along B, 4 = anything else.  A column is
  // carried only if the old and the new cell agree on its validity (a masked / padding ray in the neighbouring cell has ok == 0),
  // and the rows have to move with the cell (grids of other batch entries; unpackable sizes).
  int codev;
  {
    const int da = ((cell >> shA) & 0x3FF) - ((pcell >> shA) & 0x3FF);
    const int db = ((cell >> shB) & 0x3FF) - ((pcell >> shB) & 0x3FF);
    const int dx = ((cell >> shX) & 0x3FF) - ((pcell >> shX) & 0x3FF);
    const int dr = row0 - prow;
    const unsigned uok = (unsigned)ok;
    const bool okAp = ((pok & (mA0 << bA)) >> bA) == (uok & mA0), okAm = ((pok & mA0) << bA) == (uok & (mA0 << bA));
    const bool okBp = ((pok & (mB0 << bB)) >> bB) == (uok & mB0), okBm = ((pok & mB0) << bB) == (uok & (mB0 << bB));
    const bool along_a = db == 0 && dx == 0 && packable, along_b = da == 0 && dx == 0 && packable;
    codev = 4;
    codev = (along_a && da == 1 && dr == sA && okAp) ? 0 : codev;
    codev = (along_a && da == -1 && dr == -sA && okAm) ? 1 : codev;
    codev = (along_b && db == 1 && dr == sB && okBp) ? 2 : codev;
    codev = (along_b && db == -1 && dr == -sB && okBm) ? 3 : codev;
    // a DIAGONAL step over A and B = the A step into a virtual intermediate cell, then the B step: three of the four columns leave, the one
    // both cells share stays (its validity has to agree: old column ((da + 1) / 2, (db + 1) / 2) against the new one on the other side).
    // Second step in bits 3..: 2 / 3 as above, 7 = none.
    const int oa = (da + 1) >> 1, ob = (db + 1) >> 1;
    const int so = oa * bA + ob * bB, sn = (1 - oa) * bA + (1 - ob) * bB;    // slot of that column (+ 0 / bX) in the old / new cell
    const int bX = 1 << X;
    const unsigned pair = 1u | (1u << bX);                                   // the two corners of a column along the third axis
    const bool okD = ((pok >> so) & pair) == ((uok >> sn) & pair);
    const bool diag = dx == 0 && packable && (da == 1 || da == -1) && (db == 1 || db == -1) && dr == da * sA + db * sB && okD;
    codev = (DIAG && diag) ? ((da > 0 ? 0 : 1) | ((db > 0 ? 2 : 3) << 3)) : (codev | (7 << 3));
  }
  const int k00 = gc << X;
  const unsigned bit[2][2] = {{1u << k00, 1u << (k00 + bB)}, {1u << (k00 + bA), 1u << (k00 + bA + bB)}};
  const float4* wrow[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) wrow[a][b] = reinterpret_cast<const float4*>(wT + (k00 + a * bA + b * bB) * WLD);
  const unsigned lane_off = (unsigned)(gc * sX * C + ch);
  const bool on = !(dbg & 1);
  float acc[2][2][CPL];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int j = 0; j < CPL; ++j) acc[a][b][j] = 0.0f;
  int s_row = __builtin_amdgcn_readlane(row0, 0);
  unsigned s_ok = (unsigned)__builtin_amdgcn_readlane(ok, 0);
  auto flush = [&](int a, int b) {  // column (a, b) of the cell at s_row leaves the registers
    if ((s_ok & bit[a][b]) && on) {
      float* const rowp = feat + ((int64_t)s_row + a * sA + b * sB) * C;  // (wave-uniform base + per-lane 32-bit offset)
#pragma unroll
      for (int j = 0; j < CPL; ++j) atomic_add_f32(rowp + lane_off + 32 * j, acc[a][b][j]);
    }
  };
#pragma unroll
  for (int c8 = 0; c8 < RPW / 8; ++c8) {
    float w[2][2][8];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const float4 v0 = wrow[a][b][2 * c8], v1 = wrow[a][b][2 * c8 + 1];
        w[a][b][0] = v0.x; w[a][b][1] = v0.y; w[a][b][2] = v0.z; w[a][b][3] = v0.w;
        w[a][b][4] = v1.x; w[a][b][5] = v1.y; w[a][b][6] = v1.z; w[a][b][7] = v1.w;
      }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int rr = 8 * c8 + i;
      if (rr > 0 && ((mask >> rr) & 1u)) {
        const int code2 = __builtin_amdgcn_readlane(codev, rr);
        int code = code2 & 7;
#pragma unroll 1
        for (int pass = 0; pass < (DIAG ? 2 : 1); ++pass) {
          if (code == 0) {         // +1 along A: the columns a = 0 are left behind, a = 1 becomes a = 0
            flush(0, 0); flush(0, 1);
#pragma unroll
            for (int j = 0; j < CPL; ++j) { acc[0][0][j] = acc[1][0][j]; acc[0][1][j] = acc[1][1][j]; acc[1][0][j] = acc[1][1][j] = 0.0f; }
          } else if (code == 1) {  // -1 along A
            flush(1, 0); flush(1, 1);
#pragma unroll
            for (int j = 0; j < CPL; ++j) { acc[1][0][j] = acc[0][0][j]; acc[1][1][j] = acc[0][1][j]; acc[0][0][j] = acc[0][1][j] = 0.0f; }
          } else if (code == 2) {  // +1 along B
            flush(0, 0); flush(1, 0);
#pragma unroll
            for (int j = 0; j < CPL; ++j) { acc[0][0][j] = acc[0][1][j]; acc[1][0][j] = acc[1][1][j]; acc[0][1][j] = acc[1][1][j] = 0.0f; }
          } else if (code == 3) {  // -1 along B
            flush(0, 1); flush(1, 1);
#pragma unroll
            for (int j = 0; j < CPL; ++j) { acc[0][1][j] = acc[0][0][j]; acc[1][1][j] = acc[1][0][j]; acc[0][0][j] = acc[1][0][j] = 0.0f; }
          } else {
            flush(0, 0); flush(0, 1); flush(1, 0); flush(1, 1);
#pragma unroll
            for (int j = 0; j < CPL; ++j) acc[0][0][j] = acc[0][1][j] = acc[1][0][j] = acc[1][1][j] = 0.0f;
          }
          if (!DIAG || (code2 >> 3) == 7) break;   // (the usual case: one step)
          // the virtual intermediate cell of a diagonal step: one over along A; the columns that came along keep their validity, the
          // other side is empty (nothing to flush from it)
          const bool up = (code2 & 7) == 0;
          s_row += up ? sA : -sA;
          s_ok = up ? ((s_ok & (mA0 << bA)) >> bA) : ((s_ok & mA0) << bA);
          code = code2 >> 3;
        }
        s_row = __builtin_amdgcn_readlane(row0, rr);
        s_ok = (unsigned)__builtin_amdgcn_readlane(ok, rr);
      }
#pragma unroll
      for (int j = 0; j < CPL; ++j) {
        const float e = enc[j][rr];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) acc[a][b][j] = fmaf(w[a][b][i], e, acc[a][b][j]);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) flush(a, b);
}

// the Splatter forward's voxel walk with two carry axes: tap set, weight table, run heads -- as splat_walk_vox --, then the feature walk
// above and the unit-weight walk
// (Enc: enc[j][i] = channel (lane & 31) + 32 j of item i -- a register array [C / 32][RPW], or SplatEncConst for the transposed march)
template <int C, int RPW, int WLD, class Enc, bool DIAG = false>
LP_DEV void splat_walk_vox2(float* feat, float* wgt, const LpGrid& g, int b, float x, float y, float z, bool live, int lane,
                            const Enc& enc, float* wT, int dbg) {
  constexpr int NQ = 64 / RPW;
  constexpr int SPQ = 8 / NQ;
  const int q = lane / RPW, r = lane % RPW;
  TapSet tp;
  grid_tapset<true>(g, b, x, y, z, tp);
  if (!live) {
    tp.ok = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) tp.w[k] = 0.0f;
  }
#pragma unroll
  for (int i = 0; i < SPQ; ++i) {
    float v = tp.w[i];
#pragma unroll
    for (int qq = 1; qq < NQ; ++qq) v = (q == qq) ? tp.w[qq * SPQ + i] : v;
    wT[(q * SPQ + i) * WLD + r] = v;
  }
  const int row0 = tp.row0, ok = (int)tp.ok;
  const int prow_ = lane_prev(row0), pok_ = lane_prev(ok);  // all lanes enabled: see run_head()
  const bool head = run_head(r, row0, prow_, ok, pok_);
  const unsigned mask = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)__ballot(head));
  if (!(dbg & 64)) splat_walk_vox_feat2<C, RPW, WLD, Enc, DIAG>(feat, g, row0, tp.cell, ok, tp.su, tp.sv, tp.st, mask, lane, enc, wT, dbg);
  if (dbg & 32) return;
  splat_walk_vox_weights<RPW, WLD>(wgt, g, row0, tp.iu, tp.sv, tp.st, mask, lane, wT, dbg);
}

}  // namespace lp
