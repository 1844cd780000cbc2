// atomics_probe3.hip -- does the SCOPE of a returnless fp32 atomic add change where it executes on MI355X (XCD L2 vs memory side)?
// 4 rows x 64 B per wave instruction (the shape of the scatter walks), footprints from L2-sized to 256 MB, and an XCD-PRIVATE variant
// (every workgroup adds into the copy of its own XCD, HW_REG_XCC_ID) -- the layout a privatised gradient accumulator would have.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int SCOPE> __device__ void add(float* p, float v) {
  if constexpr (SCOPE < 0) unsafeAtomicAdd(p, v);
  else __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, SCOPE);
}
__device__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & 15u; }  // HW_REG_XCC_ID[3:0]

template <int SCOPE, bool PRIV>
__global__ void k(float* buf, uint32_t n_rows, int iters, unsigned* seen) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) atomicOr(seen, 1u << x);
  float* base = PRIV ? buf + (size_t)x * n_rows * 16 : buf;
  uint32_t h = (tid / 16) * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    add<SCOPE>(base + (size_t)((h >> 8) % n_rows) * 16 + (tid % 16), 1.0f);
  }
}
__global__ void sum(const float* buf, size_t n, double* out) {
  double s = 0; for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += buf[i];
  atomicAdd(out, s);
}

template <int SCOPE, bool PRIV>
static void run(const char* name, float* buf, size_t bytes, unsigned* seen, double* out) {
  const int blocks = 4096, threads = 256, iters = 256;
  const uint32_t rows = (uint32_t)(bytes / 64);  // rows per copy
  const size_t total = PRIV ? bytes * 8 : bytes;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(buf, 0, total)); CK(hipMemset(out, 0, 8)); CK(hipMemset(seen, 0, 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<SCOPE, PRIV>), dim3(blocks), dim3(threads), 0, 0, buf, rows, iters, seen);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
  }
  hipLaunchKernelGGL(sum, dim3(1024), dim3(256), 0, 0, buf, total / 4, out);
  double s; unsigned m; CK(hipMemcpy(&s, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&m, seen, 4, hipMemcpyDeviceToHost));
  const double seg = (double)blocks * threads / 64 * iters * 4;
  printf("%-58s %8.3f ms  %7.2f G segments/s   sum %s (xcc mask %x)\n", name, best, seg / best / 1e6,
         s == (double)blocks * threads * iters ? "exact" : "WRONG", m);
}

int main() {
  float* buf; unsigned* seen; double* out;
  CK(hipMalloc(&buf, (size_t)2 << 30)); CK(hipMalloc(&seen, 4)); CK(hipMalloc(&out, 8));
  for (size_t mb : {1, 8, 64, 256}) {
    const size_t bytes = mb << 20;
    printf("-- footprint %zu MB (shared by all XCDs)\n", mb);
    run<-1, false>("unsafeAtomicAdd", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_WAVEFRONT, false>("__hip_atomic_fetch_add relaxed, wavefront scope", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, false>("__hip_atomic_fetch_add relaxed, workgroup scope", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_AGENT, false>("__hip_atomic_fetch_add relaxed, agent scope", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_SYSTEM, false>("__hip_atomic_fetch_add relaxed, system scope", buf, bytes, seen, out);
  }
  for (size_t kb : {512, 1024, 2048, 8192}) {
    const size_t bytes = kb << 10;
    printf("-- %zu KB per XCD copy, each workgroup adds into its own XCD's copy\n", kb);
    run<-1, true>("unsafeAtomicAdd", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_WAVEFRONT, true>("wavefront scope", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, true>("workgroup scope", buf, bytes, seen, out);
    run<__HIP_MEMORY_SCOPE_AGENT, true>("agent scope", buf, bytes, seen, out);
  }
  return 0;
}
