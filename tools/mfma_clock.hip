// Microbenchmark: sustained rate of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 on random
// operands held in registers, one wave per SIMD on every CU (equal FLOPs per variant).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
__global__ __launch_bounds__(256) void k32(const bf16x8* in, float* out, int iters) {
  bf16x8 a0 = in[threadIdx.x], a1 = in[256 + threadIdx.x], b0 = in[512 + threadIdx.x], b1 = in[768 + threadIdx.x];
  f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c3, 0, 0, 0);
  }
  float s = 0; for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(const bf16x8* in, float* out, int iters) {
  bf16x8 a0 = in[threadIdx.x], a1 = in[256 + threadIdx.x], b0 = in[512 + threadIdx.x], b1 = in[768 + threadIdx.x];
  f32x4 c[8] = {};
  for (int i = 0; i < iters; ++i) {   // 8 x (16x16x32) = same FLOPs as 4 x (32x32x16)
    c[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c[0], 0, 0, 0);
    c[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c[1], 0, 0, 0);
    c[2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c[2], 0, 0, 0);
    c[3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c[3], 0, 0, 0);
    c[4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c[4], 0, 0, 0);
    c[5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c[5], 0, 0, 0);
    c[6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c[6], 0, 0, 0);
    c[7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c[7], 0, 0, 0);
  }
  float s = 0; for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += c[j][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  std::vector<unsigned short> h(1024 * 8);
  srand(1); for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
  bf16x8* din; float* dout;
  hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 256 * 4);
  hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 3; ++rep) {
    for (int v = 0; v < 2; ++v) {
      hipEventRecord(e0);
      for (int l = 0; l < 5; ++l) { if (v == 0) hipLaunchKernelGGL(k32, dim3(256), dim3(256), 0, 0, din, dout, iters); else hipLaunchKernelGGL(k16, dim3(256), dim3(256), 0, 0, din, dout, iters); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = 5.0 * 256 * 4 * (double)iters * 4 * 32768.0;
      printf("%s: %.3f ms  %.1f TFLOP/s\n", v == 0 ? "32x32x16" : "16x16x32", ms, flops / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
