// dw_bf16_test.hip -- unit test of lightplane_amd/csrc/lp_dw_bf16.h on the GPU: four waves hold X / dY of 128 rays in
// the kernels' lane layout (lane (h, r): ray r, features feat(q, h)), store the two leading bf16 limbs in the swizzled
// tiles, and every wave forms one 16 x 16 quadrant of dW = X^T dY (+ the bias row trick); compared with fp64 on the host.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I scripts/experiments -I lightplane_amd/csrc -I include scripts/experiments/dw_bf16_test.hip -o /tmp/dwt && /tmp/dwt
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "lp_dw_bf16.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
using namespace lp;

__global__ void __launch_bounds__(256) k(const float* X, const float* Y, float* dW, float* db, int xf) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  char* area = lds + wave * LimbTiles::BYTES;
  const LtWriter wr(lane);
  float x[16], y[16];
  for (int q = 0; q < 16; ++q) {
    x[q] = X[(32 * wave + r) * 32 + featq(q, h)];
    y[q] = Y[(32 * wave + r) * 32 + featq(q, h)];
  }
  for (int c = 0; c < xf / 16; ++c) {
    u32x4_t l1, l2, l3;
    split3_chunk(x + 8 * c, l1, l2, l3);
    wr.store(area + LimbTiles::X_HI, c, l1, l2);
  }
  for (int c = 0; c < 2; ++c) {
    u32x4_t l1, l2, l3;
    split3_chunk(y + 8 * c, l1, l2, l3);
    wr.store(area + LimbTiles::Y_HI, c, l1, l2);
  }
  __syncthreads();
  f32x4_t dq = {0, 0, 0, 0}, dbq = {0, 0, 0, 0};
  int mi, ni, v0, v1;
  if (xf == 32) { mi = (wave & 3) >> 1; ni = wave & 1; v0 = 0; v1 = 4; }
  else { mi = 0; ni = wave & 1; v0 = 2 * (wave >> 1); v1 = v0 + 2; }  // 16 input features: two ray groups
  const LtReader rd(lane);
  dw_quadrant_bf16<true>(lds, LimbTiles::BYTES, rd, mi, ni, v0, v1, dq, dbq, lt_row_indicator(lane, 2));
  const int n = lane & 15, g = lane >> 4;
  for (int i = 0; i < 4; ++i) atomicAdd(dW + (16 * mi + 4 * g + i) * 32 + 16 * ni + n, dq[i]);
  if (g == 0 && (mi == 0) && (xf == 32 || true)) atomicAdd(db + 16 * ni + n, dbq[2] * ((xf == 32) ? 1.0f : 1.0f));
}

int main() {
  std::vector<float> X(128 * 32), Y(128 * 32);
  srand(1);
  for (auto& v : X) v = (rand() / (float)RAND_MAX - 0.3f) * 3.0f;
  for (auto& v : Y) v = (rand() / (float)RAND_MAX - 0.5f) * 0.01f;
  float *dX, *dY, *dWd, *dbd;
  CK(hipMalloc(&dX, X.size() * 4)); CK(hipMalloc(&dY, Y.size() * 4)); CK(hipMalloc(&dWd, 32 * 32 * 4)); CK(hipMalloc(&dbd, 32 * 4));
  CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice));
  int bad = 0;
  for (int xf : {32, 16}) {
    CK(hipMemset(dWd, 0, 32 * 32 * 4)); CK(hipMemset(dbd, 0, 32 * 4));
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 4 * LimbTiles::BYTES, 0, dX, dY, dWd, dbd, xf);
    CK(hipDeviceSynchronize());
    std::vector<float> W(32 * 32), b(32);
    CK(hipMemcpy(W.data(), dWd, 32 * 32 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(b.data(), dbd, 32 * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale = 0, worst_b = 0, scale_b = 0, worst32 = 0;
    for (int i = 0; i < xf; ++i)
      for (int j = 0; j < 32; ++j) {
        double s = 0; float s32 = 0;
        for (int ray = 0; ray < 128; ++ray) { s += (double)X[ray * 32 + i] * Y[ray * 32 + j]; s32 = fmaf(X[ray * 32 + i], Y[ray * 32 + j], s32); }
        worst = fmax(worst, fabs(W[i * 32 + j] - s)); scale = fmax(scale, fabs(s)); worst32 = fmax(worst32, fabs(s32 - s));
      }
    for (int j = 0; j < 32; ++j) {
      double s = 0;
      for (int ray = 0; ray < 128; ++ray) s += Y[ray * 32 + j];
      worst_b = fmax(worst_b, fabs(b[j] - s)); scale_b = fmax(scale_b, fabs(s));
    }
    printf("input features %2d: dW max err / max |dW| = %.3e (fp32 fma chain: %.3e)   db: %.3e\n", xf, worst / scale, worst32 / scale, worst_b / scale_b);
    if (!(worst / scale < 5e-5) || !(worst_b / scale_b < 5e-5)) bad = 1;
  }
  printf(bad ? "FAILED\n" : "OK\n");
  return bad;
}
