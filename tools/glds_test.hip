// Probe: does the instruction offset of global_load_lds apply to the LDS destination too?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const char* g, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  for (int i = threadIdx.x; i < 16384/4; i += 64) ((unsigned*)smem)[i] = 0xdeadbeef;
  __syncthreads();
  unsigned keep; unsigned lds_dst = 4096u;
  unsigned voff = (threadIdx.x & 63) * 16;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\ts_mov_b32 m0, %0"
     : "=&s"(keep) : "v"(voff), "s"(g), "s"(lds_dst) : "memory");
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  for (int i = threadIdx.x; i < 16384/4; i += 64) out[i] = ((unsigned*)smem)[i];
}
int main() {
  std::vector<unsigned> h(4096), o(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;
  char* d; unsigned* dout;
  hipMalloc(&d, 16384); hipMalloc(&dout, 16384);
  hipMemcpy(d, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, d, dout);
  hipMemcpy(o.data(), dout, 16384, hipMemcpyDeviceToHost);
  int first = -1, last = -1, bad = 0;
  for (int i = 0; i < 4096; ++i) if (o[i] != 0xdeadbeef) { if (first < 0) first = i; last = i; }
  printf("written dwords: first %d last %d (expect 1024..1535 if the offset moves the LDS address too)\n", first, last);
  for (int i = 1024; i < 1536; ++i) if (o[i] != (unsigned)(i - 1024)) ++bad;
  printf("mismatches in [1024,1536): %d\n", bad);
  return 0;
}
