// tr_b16_probe.hip -- what does ds_read_b64_tr_b16 (gfx950) deliver?  Every lane supplies the byte address of four
// consecutive b16 elements; LDS element i holds the value i.  Prints, per lane, the four element indices it received
// for a few address patterns.  (standalone: hipcc --offload-arch=gfx950 -O3 scripts/tr_b16_probe.hip -o /tmp/trp)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ void k(unsigned long long* out, int mode) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned elem;
  if (mode == 0) elem = l * 4;                                  // lane l -> elements 4l .. 4l+3 (a [64][4] matrix, row per lane)
  else if (mode == 1) elem = (l & 15) * 64 + (l >> 4) * 4;      // [16 rows][64 cols]: lane = row (l & 15), column block 4 (l >> 4)
  else if (mode == 2) elem = (l >> 4) * 1024 + (l & 15) * 64;   // four [16][64] matrices, lane = row, columns 0..3
  else elem = (l & 3) * 64 + (l >> 2) * 4;                      // [4 rows][64 cols]: lane -> row l & 3, column block l >> 2
  const unsigned addr = (unsigned)(size_t)lds + elem * 2;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l] = v;
}

int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 64 * 8));
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    CK(hipDeviceSynchronize());
    unsigned long long h[64];
    CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) printf(" %5llu", (h[l] >> (16 * j)) & 0xffffull);
      printf("%s", (l % 4 == 3) ? "\n" : "   |");
    }
  }
  return 0;
}
