// Which bf16 MFMA shape sustains more FLOP/s in a loop shaped like the tower kernel's step — 12 ds_read_b128 per 8
// v_mfma_f32_32x32x16_bf16 (or per 16 v_mfma_f32_16x16x32_bf16: the same FLOPs, the same operand bytes), one wave per SIMD,
// one workgroup per CU, random operands in LDS — once the chip holds its clock down under load?  (MI355X_MICROARCH.md,
// DVFS give-back item 7: 1.12-1.14x for the 16x16x32 shape with every operand re-read from LDS.)
//     hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_probe.hip -o /tmp/mfma_shape_probe && /tmp/mfma_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int SHAPE>
__global__ __launch_bounds__(256) void probe(const bf16x8* in, float* out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16x8* lds = reinterpret_cast<bf16x8*>(smem);
    for (int i = threadIdx.x; i < 96 * 1024 / 16; i += 256) lds[i] = in[i & 4095];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 c32[2] = {};
    f32x4 c16[8] = {};
    unsigned base = lane * 16 + wave * 1024;
    bf16x8 a[2][8], b[2][4];
    auto load = [&](int set, int it) {
        const unsigned o = (it & 7) * 8192;
#pragma unroll
        for (int f = 0; f < 8; ++f) a[set][f] = *reinterpret_cast<const bf16x8*>(smem + ((o + f * 1024 + lane * 16) & 0xffff));
#pragma unroll
        for (int k = 0; k < 4; ++k) b[set][k] = *reinterpret_cast<const bf16x8*>(smem + 65536 + ((base + k * 4096 + o / 4) & 0x7fff));
    };
    auto mma = [&](int set) {
        if (SHAPE == 32) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) c32[ms] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[set][2 * k + ms], b[set][k], c32[ms], 0, 0, 0);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int t = 0; t < 4; ++t) c16[(t + 4 * (k & 1)) & 7] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[set][(2 * k + (t & 1)) & 7], b[set][(k + (t >> 1)) & 3], c16[(t + 4 * (k & 1)) & 7], 0, 0, 0);
        }
    };
    load(0, 0);
    for (int it = 0; it < iters; it += 2) {          // register double buffer: the next step's 12 reads under this step's MFMAs
        load(1, it + 1); mma(0);
        __builtin_amdgcn_sched_barrier(0);
        load(0, it + 2); mma(1);
        __builtin_amdgcn_sched_barrier(0);
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c32[0][i] + c32[1][i];
    for (int j = 0; j < 8; ++j) for (int i = 0; i < 4; ++i) s += c16[j][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main()
{
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); v = u >> 16; }
    bf16x8* din; float* dout;
    hipMalloc(&din, h.size() * 2); hipMalloc(&dout, 256 * 256 * 4);
    hipMemcpy(din, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int rep = 0; rep < 4; ++rep)
        for (int v = 0; v < 2; ++v) {
            hipEventRecord(e0);
            for (int l = 0; l < 40; ++l) {
                if (v == 0) hipLaunchKernelGGL(probe<32>, dim3(256), dim3(256), 96 * 1024, 0, din, dout, iters);
                else hipLaunchKernelGGL(probe<16>, dim3(256), dim3(256), 96 * 1024, 0, din, dout, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = 40.0 * 256 * 4 * (double)iters * 8 * 32768.0;
            printf("%s + 12 ds_read_b128 per 256 MFMA-cycles: %.3f ms  %.1f TFLOP/s\n", v == 0 ? "32x32x16" : "16x16x32", ms, flops / (ms * 1e-3) / 1e12);
        }
    return 0;
}
