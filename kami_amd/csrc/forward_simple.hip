// forward_simple.hip — exact-order fp32 forward pass, one launch per layer.
//
// This is the on-device correctness anchor (cfg.dtype = KH_F32): plain fp32 VALU FMAs in the
// same tap-major / channel-inner order as the CPU oracle, so it tracks the reference's
// fp32 libtorch path to accumulation-order noise.  The throughput path is tower_mfma.hip.
//
// Layers (kami/nn/nn.cpp:59-91, 26-34): conv + eval-BatchNorm folded to (scale, shift) in the
// epilogue + ReLU (+ skip added AFTER the ReLU, nn.cpp:31), 1x1 heads, softmax over 4672,
// Linear(64,256) + tanh.
#include "kh_internal.h"

namespace kh {

// One thread per (board, pixel, co); co fastest so weight reads coalesce and the activation
// read is a broadcast within a group of Co threads.
__global__ __launch_bounds__(256) void simple_conv_kernel(const float* __restrict__ wt,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ shift,
                                                          const float* __restrict__ in,
                                                          const float* __restrict__ skip,
                                                          float* __restrict__ out,
                                                          int B, int Ci, int Co, int taps, int relu)
{
    const long total = (long)B * 64 * Co;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long)gridDim.x * blockDim.x) {
        const int co = (int)(idx % Co);
        const int pix = (int)((idx / Co) & 63);
        const long b = idx / ((long)Co * 64);
        const int py = pix >> 3, px = pix & 7;
        const float* xb = in + b * 64 * Ci;
        float acc = 0.0f;
        if (taps == 9) {
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = py + ky - 1;
                if (iy < 0 || iy > 7) continue;
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = px + kx - 1;
                    if (ix < 0 || ix > 7) continue;
                    const float* xi = xb + (iy * 8 + ix) * Ci;
                    const float* wk = wt + (size_t)(ky * 3 + kx) * Ci * Co + co;
                    for (int ci = 0; ci < Ci; ++ci) acc = fmaf(xi[ci], wk[(size_t)ci * Co], acc);
                }
            }
        } else {
            const float* xi = xb + pix * Ci;
            const float* wk = wt + co;
            for (int ci = 0; ci < Ci; ++ci) acc = fmaf(xi[ci], wk[(size_t)ci * Co], acc);
        }
        float v = fmaf(acc, scale[co], shift[co]);
        if (relu) v = v < 0.0f ? 0.0f : v;            // NaN propagates, like torch::relu
        if (skip) v = skip[idx] + v;                  // x + relu(...)   nn.cpp:31
        out[idx] = v;
    }
}

void launch_simple_conv(const SimpleLayer& L, const float* in, const float* skip, float* out,
                        int B, hipStream_t s)
{
    long total = (long)B * 64 * L.Co;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(simple_conv_kernel, dim3(blocks), dim3(256), 0, s, L.wt, L.scale, L.shift,
                       in, skip, out, B, L.Ci, L.Co, L.taps, L.relu);
}

__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup (256 threads) per board.  exp(log_softmax(x)) = exp((x - max) - log(sum exp(x - max))).
__global__ __launch_bounds__(256) void softmax4672_kernel(const float* __restrict__ logits,
                                                          float* __restrict__ policy, int B, int* flags)
{
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        const float* x = logits + (size_t)b * KH_PSIZE;
        float v[19];
        float m = -INFINITY;
        bool nan = false;
#pragma unroll
        for (int j = 0; j < 19; ++j) {
            const int i = tid + 256 * j;
            v[j] = i < KH_PSIZE ? x[i] : -INFINITY;
            nan |= (v[j] != v[j]);
            m = fmaxf(m, v[j]);
        }
        m = wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 19; ++j) s += (tid + 256 * j < KH_PSIZE) ? expf(v[j] - m) : 0.0f;
        s = wave_sum(s);
        if (lane == 0) red[4 + wave] = s;
        __syncthreads();
        s = (red[4] + red[5]) + (red[6] + red[7]);
        const float ls = logf(s);
        float* p = policy + (size_t)b * KH_PSIZE;
#pragma unroll
        for (int j = 0; j < 19; ++j) {
            const int i = tid + 256 * j;
            if (i < KH_PSIZE) p[i] = nan ? v[j] : expf((v[j] - m) - ls);
        }
        // any NaN logit poisons the whole row in the reference (max/sum become NaN)
        if (__syncthreads_or(nan)) {
            for (int i = tid; i < KH_PSIZE; i += 256) p[i] = NAN;
            if (tid == 0) atomicOr(&flags[0], 1);
        }
        __syncthreads();
    }
}

void launch_softmax4672(const float* logits, float* policy, int B, int* flags, hipStream_t s)
{
    int blocks = B < 2048 ? B : 2048;
    hipLaunchKernelGGL(softmax4672_kernel, dim3(blocks), dim3(256), 0, s, logits, policy, B, flags);
}

// One workgroup (256 threads) per board: thread j computes value_full[b][j].
__global__ __launch_bounds__(256) void value_fc_kernel(const float* __restrict__ v64,
                                                       const float* __restrict__ fcw,
                                                       const float* __restrict__ fcb,
                                                       float* __restrict__ value_full, int B, int* flags)
{
    __shared__ float v[64];
    const int j = threadIdx.x;
    for (int b = blockIdx.x; b < B; b += gridDim.x) {
        if (j < 64) v[j] = v64[(size_t)b * 64 + j];
        __syncthreads();
        float a = 0.0f;
        const float* w = fcw + (size_t)j * 64;
#pragma unroll 8
        for (int k = 0; k < 64; ++k) a = fmaf(v[k], w[k], a);
        const float r = tanhf(a + fcb[j]);
        value_full[(size_t)b * KH_VALUE_WIDTH + j] = r;
        if (r != r) atomicOr(&flags[1], 1);
        __syncthreads();
    }
}

void launch_value_fc(const float* v64, const float* fcw, const float* fcb, float* value_full,
                     int B, int* flags, hipStream_t s)
{
    int blocks = B < 2048 ? B : 2048;
    hipLaunchKernelGGL(value_fc_kernel, dim3(blocks), dim3(256), 0, s, v64, fcw, fcb, value_full, B, flags);
}

}  // namespace kh
