// tr_b16_timing.hip -- LDS cycles of ds_read_b64_tr_b16 under the address patterns of the forward weight-operand loader (AColsFwd,
// lp_bf3.h), against ds_read_b64 with the same addresses.  One wave, 256 back-to-back reads per pattern, s_memtime around them.
//   pattern 0: every lane the same address (broadcast: the floor)
//   pattern 1: rounds 2-4 layout, 72-byte rows: lane (s = l & 15, m0 = l & 16, h = l >> 5) -> row 4h + (s >> 2), column m0 + 4 (s & 3)
//   pattern 2: round 5 layout, 64-byte rows + 8 bytes of skew per group of four rows
//   pattern 3: lane l -> 8 l bytes (a dense 512-byte block)
//   patterns 4-7: the BACKWARD loader's row reads (ARowsBwd: lane k reads 8 bytes of row k at one column offset) under four row layouts
// standalone: hipcc --offload-arch=gfx950 -O3 scripts/tr_b16_timing.hip -o /tmp/trt
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <bool TR>
__global__ void k(unsigned long long* out, int pattern) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[16384];
  for (int i = threadIdx.x; i < 16384; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, s = l & 15, m0 = l & 16, h = l >> 5;
  const int row = 4 * h + (s >> 2), col = m0 + 4 * (s & 3);
  unsigned off;
  if (pattern == 0) off = 0;
  else if (pattern == 1) off = (row * 36 + col) * 2;
  else if (pattern == 2) off = row * 64 + (row >> 2) * 8 + col * 2;
  else if (pattern == 3) off = l * 8;
  else if (pattern == 4) off = (l & 31) * 72 + 8 * (l >> 5);                          // backward row reads, 72-byte rows
  else if (pattern == 5) off = (l & 31) * 64 + ((l & 31) >> 2) * 8 + 8 * (l >> 5);   // backward row reads, skewed 64-byte rows
  else if (pattern == 6) off = (l & 31) * 80 + 8 * (l >> 5);                          // 80-byte rows
  else off = (l & 31) * 64 + ((l & 31) >> 2) * 16 + 8 * (l >> 5);                     // 64-byte rows + 16 bytes of skew per four rows
  unsigned addr = (unsigned)(size_t)lds + off;
  unsigned long long acc = 0, v;
  unsigned long long t0, t1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
#pragma unroll 1
  for (int it = 0; it < 16; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (TR) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(v) : "v"(addr) : "memory");
      else asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(v) : "v"(addr) : "memory");
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
      acc += v;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
  out[l] = acc;
  if (l == 0) out[64] = t1 - t0;
}

int main() {
  unsigned long long* d;
  CK(hipMalloc(&d, 65 * 8));
  for (int tr = 0; tr < 2; ++tr)
    for (int p = 0; p < 8; ++p) {
      unsigned long long best = ~0ull;
      for (int rep = 0; rep < 5; ++rep) {
        if (tr) hipLaunchKernelGGL(k<true>, dim3(1), dim3(64), 0, 0, d, p);
        else hipLaunchKernelGGL(k<false>, dim3(1), dim3(64), 0, 0, d, p);
        CK(hipDeviceSynchronize());
        unsigned long long h[65];
        CK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
        if (h[64] < best) best = h[64];
      }
      printf("%s pattern %d: %llu s_memtime ticks for 256 reads = %.2f per read\n", tr ? "ds_read_b64_tr_b16" : "ds_read_b64       ", p, best, best / 256.0);
    }
  return 0;
}
