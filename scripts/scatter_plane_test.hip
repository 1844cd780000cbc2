// scatter_plane_test.hip -- unit test of the run-merged plane scatter (lp_mfma_common.h scatter_plane_ax and the
// generic per-slot walk of scatter_grid) on ONE wave with hand-made ray patterns, against a CPU loop.  (standalone)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I lightplane_amd/csrc scripts/scatter_plane_test.hip -o /tmp/spt && /tmp/spt
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "lp_mfma_common.h"

namespace lp {
int set_error(int code, const char*, ...) { return code; }
int check_launch(const char*) { return 0; }
}  // namespace lp
using namespace lp;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// one wave: lane (h, r) = ray r; x, y, z per ray; dx0 [32][C]; mode 0 = scatter_plane (GM_TRIPLANE), 1 = per-slot walk
template <int C>
__global__ void __launch_bounds__(64) k(float* gg, LpGrid g, const float* xyz, const float* dx0, const int* live_in, int mode) {
  __shared__ __attribute__((aligned(16))) float xt[32 * DX_LD];
  __shared__ __attribute__((aligned(16))) float yt[32 * DX_LD];
  const int lane = threadIdx.x, h = lane >> 5, r = lane & 31;
  const float x = xyz[3 * r], y = xyz[3 * r + 1], z = xyz[3 * r + 2];
  // dx0 tile [channel][ray], as the backward kernel writes it
  for (int q = 0; q < C / 2; ++q) xt[featq(q, h) * DX_LD + r] = dx0[r * C + featq(q, h)];
  __syncthreads();
  const bool live = live_in[r] != 0;
  if (mode == 0) scatter_grid<C, GM_TRIPLANE>(gg, g, 0, x, y, z, live, lane, xt, yt, 0);
  else scatter_grid<C, GM_GENERIC>(gg, g, 0, x, y, z, live, lane, xt, yt, 0);
}

static void cpu_ref(std::vector<float>& out, const LpGrid& g, int C, const float* xyz, const float* dx0, const int* live) {
  const int U = g.W, V = g.H;  // xy plane
  for (int r = 0; r < 32; ++r) {
    if (!live[r]) continue;
    const float tu = ((xyz[3 * r] + 1.0f) * U - 1.0f) / 2.0f, tv = ((xyz[3 * r + 1] + 1.0f) * V - 1.0f) / 2.0f;
    const float fu = floorf(tu), fv = floorf(tv);
    for (int bv = 0; bv < 2; ++bv)
      for (int bu = 0; bu < 2; ++bu) {
        const int iu = (int)fu + bu, iv = (int)fv + bv;
        if (iu < 0 || iu >= U || iv < 0 || iv >= V) continue;
        const float w = (bu ? tu - fu : (fu + 1.0f) - tu) * (bv ? tv - fv : (fv + 1.0f) - tv);
        for (int c = 0; c < C; ++c) out[(g.row_offset + (int64_t)iv * U + iu) * C + c] += w * dx0[r * C + c];
      }
  }
}

template <int C>
static int run_case(const char* name, const float* xs, const float* ys, const int* live, int U, int V, int row_offset) {
  LpGrid g;
  g.B = 1; g.D = 1; g.H = V; g.W = U; g.row_offset = row_offset;
  const int rows = row_offset + U * V + 8;
  std::vector<float> xyz(96), dx0(32 * C), ref((size_t)rows * C, 0.0f);
  for (int r = 0; r < 32; ++r) { xyz[3 * r] = xs[r]; xyz[3 * r + 1] = ys[r]; xyz[3 * r + 2] = 0.1f; }
  for (int i = 0; i < 32 * C; ++i) dx0[i] = (float)((i * 7919) % 1000) / 500.0f - 1.0f;
  cpu_ref(ref, g, C, xyz.data(), dx0.data(), live);
  float *d_g, *d_xyz, *d_dx;
  int* d_live;
  CK(hipMalloc(&d_g, ref.size() * 4)); CK(hipMalloc(&d_xyz, 96 * 4)); CK(hipMalloc(&d_dx, dx0.size() * 4)); CK(hipMalloc(&d_live, 32 * 4));
  CK(hipMemcpy(d_xyz, xyz.data(), 96 * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_dx, dx0.data(), dx0.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_live, live, 32 * 4, hipMemcpyHostToDevice));
  int bad_total = 0;
  for (int mode = 0; mode < 2; ++mode) {
    CK(hipMemset(d_g, 0, ref.size() * 4));
    hipLaunchKernelGGL((k<C>), dim3(1), dim3(64), 0, 0, d_g, g, d_xyz, d_dx, d_live, mode);
    CK(hipDeviceSynchronize());
    std::vector<float> got(ref.size());
    CK(hipMemcpy(got.data(), d_g, ref.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0, first = -1;
    float scale = 0.0f;
    for (float v : ref) scale = fmaxf(scale, fabsf(v));
    for (size_t i = 0; i < ref.size(); ++i)
      if (fabsf(got[i] - ref[i]) > 1e-5f * scale + 1e-12f) { if (first < 0) first = (int)i; ++bad; }
    printf("%-40s C=%d U=%d V=%d off=%d %-10s: %d bad entries", name, C, U, V, row_offset, mode == 0 ? "plane_ax" : "per-slot", bad);
    if (bad) printf("  first at row %d ch %d: got %g want %g", first / C, first % C, got[first], ref[first]);
    printf("\n");
    bad_total += bad;
  }
  CK(hipFree(d_g)); CK(hipFree(d_xyz)); CK(hipFree(d_dx)); CK(hipFree(d_live));
  return bad_total;
}

int main() {
  const int U = 24, V = 24;
  auto coord = [](float t, int n) { return (2.0f * t + 1.0f) / n - 1.0f; };  // un-normalised t -> normalised coordinate
  float xs[32], ys[32];
  int live[32];
  int bad = 0;
  // the failing pattern of round 2: iu = -1,-1,-1,0,0,0,1,1,1,..., iv = 0, first ray dead
  for (int r = 0; r < 32; ++r) { xs[r] = coord(-0.8f + 0.3f * r, U); ys[r] = coord(0.37f, V); live[r] = r > 0; }
  for (int off : {0, 576}) {
    bad += run_case<16>("left border, iv = 0, ray 0 dead", xs, ys, live, U, V, off);
    bad += run_case<32>("left border, iv = 0, ray 0 dead", xs, ys, live, U, V, off);
  }
  for (int r = 0; r < 32; ++r) live[r] = 1;
  bad += run_case<16>("left border, iv = 0, all live", xs, ys, live, U, V, 0);
  for (int r = 0; r < 32; ++r) ys[r] = coord(5.37f, V);
  bad += run_case<16>("left border, iv = 5", xs, ys, live, U, V, 0);
  for (int r = 0; r < 32; ++r) { xs[r] = coord(21.2f + 0.3f * r, U); ys[r] = coord(0.37f, V); }
  bad += run_case<16>("right border (some rays leave the plane)", xs, ys, live, U, V, 0);
  for (int r = 0; r < 32; ++r) { xs[r] = coord(3.3f, U); ys[r] = coord(-0.9f + 0.25f * r, V); }
  bad += run_case<16>("top border walk in v", xs, ys, live, U, V, 0);
  for (int r = 0; r < 32; ++r) { xs[r] = coord(-0.7f + 0.21f * r, U); ys[r] = coord(-0.6f + 0.17f * r, V); live[r] = (r % 5) != 2; }
  bad += run_case<16>("diagonal from the corner, dead rays inside", xs, ys, live, U, V, 0);
  bad += run_case<32>("diagonal from the corner, dead rays inside", xs, ys, live, U, V, 48);
  for (int r = 0; r < 32; ++r) { xs[r] = coord(7.4f, U); ys[r] = coord(9.6f, V); live[r] = 1; }
  bad += run_case<16>("all rays in one cell", xs, ys, live, U, V, 0);
  printf(bad ? "FAILED\n" : "ALL OK\n");
  return bad ? 1 : 0;
}
