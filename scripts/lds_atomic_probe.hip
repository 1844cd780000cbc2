// lds_atomic_probe.hip -- throughput of ds_add_f32 (no return) on MI355X, conflict-free and with
// the address pattern of the gradient-scatter "box" accumulation (4 cells x 16 channels per wave
// instruction).  Standalone: hipcc --offload-arch=gfx950 -O3 scripts/lds_atomic_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// MODE 0: ds_add_f32, every lane its own float (lane-contiguous)          -> conflict-free
// MODE 1: ds_add_f32, 4 cells (c, c+1, c+BU, c+BU+1) x 16 channels, c varies per iteration
// MODE 2: like 1 but all waves of the block hit the SAME cells (cross-wave same-address)
// MODE 3: plain ds_write_b32 in the pattern of mode 1 (reference)
// MODE 4: ds_read_b32 + v_add + ds_write_b32 (non-atomic RMW) pattern 1
template <int MODE>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ float tile[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) tile[i] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = lane >> 4, sub = lane & 15;
  const int BU = 13;
  float* base = tile + ((MODE == 2) ? 0 : wave * 2048);
  unsigned h = 12345u + wave * 7u;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 16
    for (int j = 0; j < 16; ++j) {
      h = h * 1664525u + 1013904223u;
      int idx;
      if (MODE == 0) idx = ((h >> 8) % 24) * 64 + lane;
      else {
        const int c = (h >> 8) % 96;
        const int cell = c + (grp & 1) + (grp >> 1) * BU;
        idx = cell * 16 + sub;
      }
      const float v = (float)(j + 1);
      if (MODE == 3) base[idx] = v;
      else if (MODE == 4) base[idx] = base[idx] + v;
      else atomicAdd(base + idx, v);
    }
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[blockIdx.x] = tile[5]; atomicAdd(cyc, t1 - t0); }
}

int main() {
  float* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&cyc, 8));
  const int iters = 256;
  const char* names[] = {"ds_add_f32 lane-contiguous", "ds_add_f32 4 cells x 16 ch (per-wave tiles)", "ds_add_f32 4 cells x 16 ch (shared tile)", "ds_write_b32 4 cells x 16 ch", "read+add+write 4 cells x 16 ch"};
  for (int waves = 4; waves <= 8; waves += 4) {
    for (int mode = 0; mode < 5; ++mode) {
      CK(hipMemset(cyc, 0, 8));
      const int blocks = 256;  // one per CU
      switch (mode) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64 * waves), 0, 0, out, iters, cyc); break;
      }
      CK(hipDeviceSynchronize());
      unsigned long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
      const double per_block = (double)c / blocks;
      const double n_instr = (double)iters * 16 * waves;  // wave-instructions per CU
      printf("waves/CU=%d  %-48s %8.1f cycles per wave-instruction per CU (%.0f cycles, %.0f instr)\n", waves, names[mode], per_block / n_instr, per_block, n_instr);
    }
  }
  return 0;
}
