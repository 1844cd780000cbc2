// mfma_valu_overlap.hip -- does matrix-pipe time overlap with vector-ALU time on MI355X (gfx950)?  (standalone)
//
// The Renderer kernels spend their time in two instruction classes: MFMA (the decoder's matrix products) and plain
// VALU (interpolation, activations, compositing, the scatter walk).  DESIGN.md 4 claims that with fp32 MFMA
// (v_mfma_f32_32x32x2_f32) the two ADD UP instead of overlapping -- which caps everything else -- and that claim
// deserves its own measurement.  This probe times, per SIMD,
//   M     : a stream of NM MFMA instructions (4 independent accumulators, no dependency stalls)
//   V     : a stream of NV independent v_fma_f32
//   MV1   : both in ONE wave, interleaved (1 MFMA, then NV/NM VALU, ...), one wave per SIMD
//   MV2   : the same interleaved stream in TWO waves per SIMD
//   M|V   : two waves per SIMD, one issues only the MFMA stream, the other only the VALU stream
// for the fp32 MFMA the kernels use today and for the bf16 / f16 MFMA (v_mfma_f32_32x32x16_{bf16,f16}) that a
// split-precision decoder would use.  If time(MV) ~ max(M, V) the pipes overlap; if ~ M + V they do not.
//
//   hipcc --offload-arch=gfx950 -O3 scripts/mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { K_F32 = 0, K_BF16 = 1, K_F16 = 2, K_F32_16 = 3, K_BF16_16 = 4 };

template <int KIND>
__device__ __forceinline__ void mfma32(f32x16& acc, const f32x4& a, const f32x4& b) {
  if (KIND == K_F32) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.x), "v"(b.x));
  else if (KIND == K_BF16) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int KIND>
__device__ __forceinline__ void mfma16(f32x4& acc, const f32x4& a, const f32x4& b) {
  if (KIND == K_F32_16) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a.x), "v"(b.x));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}

#define VFMA(x) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x) : "v"(va), "v"(vb))

// role: 0 = MFMA stream only, 1 = VALU stream only, 2 = interleaved; NM MFMA and NM * R VALU per iteration
template <int KIND, int NM, int R>
__global__ void __launch_bounds__(512) probe(float* out, unsigned long long* cyc, int iters, int role_lo, int role_hi) {
  const int wave = threadIdx.x >> 6;
  const int role = __builtin_amdgcn_readfirstlane((wave & 4) ? role_hi : role_lo);  // waves w and w + 4 share a SIMD
  f32x16 acc[4];
  f32x4 acc4[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
    acc4[i] = (f32x4){0, 0, 0, 0};
  }
  const float t = (float)threadIdx.x * 1e-3f;
  f32x4 a = {t, t + 1.0f, t + 2.0f, t + 3.0f}, b = {0.5f, 0.25f, 0.125f, 0.0625f};
  if (KIND != K_F32 && KIND != K_F32_16) {  // finite 16-bit payloads
    a = (f32x4){0, 0, 0, 0};
    b = (f32x4){0, 0, 0, 0};
  }
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = t + (float)i;
  const float va = 0.999f, vb = 1e-3f;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (role == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (KIND <= K_F16) mfma32<KIND>(acc[m & 3], a, b); else mfma16<KIND>(acc4[m & 3], a, b);
      }
    }
  } else if (role == 1) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < NM * R; ++m) VFMA(v[m & 7]);
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < NM; ++m) {
        if (KIND <= K_F16) mfma32<KIND>(acc[m & 3], a, b); else mfma16<KIND>(acc4[m & 3], a, b);
#pragma unroll
        for (int r = 0; r < R; ++r) VFMA(v[(m * R + r) & 7]);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int j = 0; j < 16; ++j) s += acc[i][j];
    s += acc4[i].x + acc4[i].y + acc4[i].z + acc4[i].w;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int NM, int R>
static double run(int threads, int role_lo, int role_hi, float* out, unsigned long long* cyc, double* wall_ms) {
  const int blocks = 256, iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e30, best_ms = 1e30;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(cyc, 0, blocks * 8 * sizeof(unsigned long long)));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((probe<KIND, NM, R>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, role_lo, role_hi);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[256 * 8];
    CK(hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost));
    double mx = 0;
    for (int i = 0; i < blocks * 8; ++i) mx = h[i] > mx ? (double)h[i] : mx;
    if (mx < best) best = mx;
    if (ms < best_ms) best_ms = ms;
  }
  *wall_ms = best_ms;
  return best / iters;  // s_memtime ticks (100 MHz constant clock on gfx9: converted by the caller through wall time)
}

template <int KIND, int NM, int R>
static void family(const char* name, float* out, unsigned long long* cyc) {
  double ms[6];
  const double m1 = run<KIND, NM, R>(256, 0, 0, out, cyc, &ms[0]);
  const double v1 = run<KIND, NM, R>(256, 1, 1, out, cyc, &ms[1]);
  const double mv1 = run<KIND, NM, R>(256, 2, 2, out, cyc, &ms[2]);
  const double mv2 = run<KIND, NM, R>(512, 2, 2, out, cyc, &ms[3]);
  const double m_v = run<KIND, NM, R>(512, 0, 1, out, cyc, &ms[4]);
  const double m2 = run<KIND, NM, R>(512, 0, 0, out, cyc, &ms[5]);
  // wall time per iteration in shader cycles at 2.4 GHz (the kernel is one resident wave set: wall = per-wave time)
  const double f = 2.4e9 * 1e-3 / 2000.0;
  printf("%-34s NM=%2d NV=%3d | cycles/iter @2.4GHz: M %7.0f  V %7.0f  MV(1 wave) %7.0f  MV(2 waves, per wave) %7.0f  "
         "M|V(2 waves) %7.0f  M(2 waves) %7.0f | M+V %7.0f max %7.0f\n",
         name, NM, NM * R, ms[0] * f, ms[1] * f, ms[2] * f, ms[3] * f / 2, ms[4] * f, ms[5] * f / 2,
         (ms[0] + ms[1]) * f, (ms[0] > ms[1] ? ms[0] : ms[1]) * f);
  (void)m1; (void)v1; (void)mv1; (void)mv2; (void)m_v; (void)m2;
}

int main() {
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * sizeof(float)));
  CK(hipMalloc(&cyc, 256 * 8 * sizeof(unsigned long long)));
  printf("per-SIMD cost of an instruction stream; MV = MFMA and VALU interleaved; M|V = one wave each (same SIMD)\n");
  family<K_F32, 16, 8>("v_mfma_f32_32x32x2_f32 + v_fma", out, cyc);
  family<K_F32, 16, 16>("v_mfma_f32_32x32x2_f32 + v_fma", out, cyc);
  family<K_F32_16, 16, 8>("v_mfma_f32_16x16x4_f32 + v_fma", out, cyc);
  family<K_BF16, 16, 4>("v_mfma_f32_32x32x16_bf16 + v_fma", out, cyc);
  family<K_BF16, 16, 8>("v_mfma_f32_32x32x16_bf16 + v_fma", out, cyc);
  family<K_F16, 16, 8>("v_mfma_f32_32x32x16_f16 + v_fma", out, cyc);
  family<K_BF16_16, 16, 4>("v_mfma_f32_16x16x32_bf16 + v_fma", out, cyc);
  return 0;
}
