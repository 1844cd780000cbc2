// atomics_probe2.hip -- cost model of global_atomic_add_f32 on MI355X: how does the rate depend on the
// number of distinct rows / cache lines one wave instruction touches?  (standalone)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// LANES_PER_ROW lanes share one random row of LANES_PER_ROW floats; ACTIVE of 64 lanes take part
template <int LANES_PER_ROW, int ACTIVE>
__global__ void k(float* buf, uint32_t n_rows, int iters) {
  const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t h = (tid / LANES_PER_ROW) * 2654435761u + 12345u;
  for (int i = 0; i < iters; ++i) {
    h = h * 1664525u + 1013904223u;
    float* p = buf + (size_t)((h >> 8) % n_rows) * LANES_PER_ROW + (tid % LANES_PER_ROW);
    if (lane < ACTIVE) unsafeAtomicAdd(p, 1.0f);
  }
}

template <int L, int A>
static void run(const char* name, float* buf, size_t bytes) {
  const int blocks = 4096, threads = 256, iters = 256;
  const uint32_t rows = (uint32_t)(bytes / (L * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<L, A>), dim3(blocks), dim3(threads), 0, 0, buf, rows, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float t; CK(hipEventElapsedTime(&t, e0, e1)); if (t < best) best = t;
  }
  const double instr = (double)blocks * threads / 64 * iters;
  const double rows_per_instr = (double)A / L;
  printf("%-44s %7.3f ms  %6.2f G instr/s  %6.2f G rows/s  %7.1f GB/s payload\n", name, best, instr / best / 1e6,
         instr * rows_per_instr / best / 1e6, instr * A * 4 / best / 1e6);
}

int main() {
  float* buf; const size_t bytes = 256u << 20;
  CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 0, bytes));
  run<16, 64>("4 rows x 64 B per instr", buf, bytes);
  run<32, 64>("2 rows x 128 B per instr", buf, bytes);
  run<64, 64>("1 row x 256 B per instr", buf, bytes);
  run<16, 16>("1 row x 64 B per instr (16 lanes)", buf, bytes);
  run<32, 32>("1 row x 128 B per instr (32 lanes)", buf, bytes);
  run<16, 32>("2 rows x 64 B per instr (32 lanes)", buf, bytes);
  run<1, 64>("64 rows x 4 B per instr", buf, bytes);
  run<1, 8>("8 rows x 4 B per instr (8 lanes)", buf, bytes);
  run<1, 2>("2 rows x 4 B per instr (2 lanes)", buf, bytes);
  run<2, 64>("32 rows x 8 B per instr", buf, bytes);
  run<4, 64>("16 rows x 16 B per instr", buf, bytes);
  run<8, 64>("8 rows x 32 B per instr", buf, bytes);
  return 0;
}
