// Probe: LDS-DMA destination bases above 64 KB (is M0's LDS base wider than 16 bits on gfx950?) and the
// one-dword-per-lane form (256 contiguous bytes per instruction).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int LDSB = 160 * 1024;
__global__ void k(const char* g, unsigned* out, unsigned dst4, unsigned dst1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < LDSB / 4; i += 64) ((unsigned*)smem)[i] = 0xdeadbeef;
  __syncthreads();
  unsigned keep;
  unsigned voff = (threadIdx.x & 63) * 16, voff1 = (threadIdx.x & 63) * 4;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
     : "=&s"(keep) : "v"(voff), "s"(g), "s"(dst4) : "memory");
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
     : "=&s"(keep) : "v"(voff1), "s"(g), "s"(dst1) : "memory");
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = threadIdx.x; i < LDSB / 4; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  std::vector<unsigned> h(4096), o(LDSB / 4);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  char* d; unsigned* dout;
  hipMalloc(&d, 16384); hipMalloc(&dout, LDSB);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
  const unsigned dst4 = 100000, dst1 = 140000 - 2048;      // the dword form: instruction offset 2048 moves both addresses
  hipLaunchKernelGGL(k, dim3(1), dim3(64), LDSB, 0, d, dout, dst4, dst1);
  hipMemcpy(o.data(), dout, LDSB, hipMemcpyDeviceToHost);
  int n = 0;
  for (int i = 0; i < LDSB / 4; ++i) if (o[i] != 0xdeadbeef) { if (n < 4 || (i % 64) == 0) printf("dword %d (byte %d) = %u\n", i, i * 4, o[i]); ++n; }
  printf("written dwords: %d (expect 256 at byte %u holding 0..255 and 64 at byte %u holding 512..575)\n", n, dst4, dst1 + 2048);
  return 0;
}
