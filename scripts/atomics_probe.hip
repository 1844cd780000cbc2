// atomics_probe.hip -- micro-benchmark of fp32 global atomics on MI355X (standalone).
// Measures the throughput of global_atomic_add_f32 under the contention patterns of the
// Renderer backward (many rays hitting few plane cells) to decide on LDS pre-aggregation.
//   hipcc --offload-arch=gfx950 -O3 scripts/atomics_probe.hip -o gpurun_out/atomics_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// each thread does `iters` x 16 atomics to row (hash % n_rows), 16 consecutive floats (one 64B line)
template <int SCOPE>
__global__ void k_atomic(float* buf, uint32_t n_rows, int iters, int coherent_lanes) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = (coherent_lanes ? tid / coherent_lanes : tid) * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    float* p = buf + (size_t)((h >> 8) % n_rows) * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      if (SCOPE == 0) unsafeAtomicAdd(p + c, 1.0f);
      else __hip_atomic_fetch_add(p + c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
}
// channel-parallel: 16 adjacent lanes add to the 16 floats of one row
__global__ void k_atomic_coalesced(float* buf, uint32_t n_rows, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = (tid / 16) * 2654435761u + 12345u;
  for (int i = 0; i < iters * 16; ++i) {
    h = h * 1664525u + 1013904223u;
    unsafeAtomicAdd(buf + (size_t)((h >> 8) % n_rows) * 16 + (tid & 15), 1.0f);
  }
}
// channel-parallel, drained (s_waitcnt vmcnt(0)) after every batch of 16 instructions: exposed latency
__global__ void k_atomic_drain(float* buf, uint32_t n_rows, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = (tid / 16) * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    for (int j = 0; j < 16; ++j) {
      h = h * 1664525u + 1013904223u;
      unsafeAtomicAdd(buf + (size_t)((h >> 8) % n_rows) * 16 + (tid & 15), 1.0f);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
// plain (non-atomic) scattered float4 stores for reference
__global__ void k_store(float* buf, uint32_t n_rows, int iters) {
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = tid * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    float4* p = reinterpret_cast<float4*>(buf + (size_t)((h >> 8) % n_rows) * 16);
    for (int c = 0; c < 4; ++c) p[c] = make_float4(1, 1, 1, 1);
  }
}
// LDS atomics: ds_add_f32 into a 16K-float LDS table with random (conflicting) addresses
__global__ void k_lds_atomic(float* out, int iters, uint32_t n_rows) {
  __shared__ float tab[16384];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) tab[i] = 0;
  __syncthreads();
  uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t h = tid * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    float* p = tab + ((h >> 8) % n_rows) * 16;
#pragma unroll
    for (int c = 0; c < 16; ++c) atomicAdd(p + c, 1.0f);
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = tab[0];
}

int main() {
  const int blocks = 2048, threads = 256, iters = 64;
  float* buf;
  const size_t max_rows = 1 << 22;  // 256 MB
  CK(hipMalloc(&buf, max_rows * 64));
  CK(hipMemset(buf, 0, max_rows * 64));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const double total = (double)blocks * threads * iters * 16;
  uint32_t rows_list[] = {64, 256, 1024, 4096, 12288, 196608, 1u << 22};
  printf("float atomics: %d blocks x %d thr x %d iters x 16 ch = %.3g adds\n", blocks, threads, iters, total);
  for (uint32_t rows : rows_list) {
    for (int variant = 0; variant < 7; ++variant) {
      float ms = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        switch (variant) {
          case 0: hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters, 0); break;
          case 1: hipLaunchKernelGGL(k_atomic<1>, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters, 0); break;
          case 2: hipLaunchKernelGGL(k_atomic<0>, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters, 8); break;
          case 3: hipLaunchKernelGGL(k_atomic_coalesced, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters); break;
          case 4: hipLaunchKernelGGL(k_store, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters); break;
          case 6: hipLaunchKernelGGL(k_atomic_drain, dim3(blocks), dim3(threads), 0, 0, buf, rows, iters); break;
          case 5: hipLaunchKernelGGL(k_lds_atomic, dim3(blocks), dim3(threads), 0, 0, buf, iters, rows > 1024 ? 1024u : rows); break;
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        if (rep == 0 || t < ms) ms = t;
      }
      const char* names[] = {"agent-scope lane-per-row", "wg-scope lane-per-row", "agent 8 lanes same row", "agent channel-parallel(16 lanes/row)", "plain float4 stores", "LDS ds_add_f32 (1024 rows)", "channel-parallel, drained every 16"};
      printf("rows=%8u  %-38s %8.3f ms  %8.2f Gadds/s  %8.1f GB/s payload\n", rows, names[variant], ms, total / ms / 1e6, total * 4 / ms / 1e6);
    }
  }
  return 0;
}
