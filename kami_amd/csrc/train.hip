// train.hip — NN::train (kami/nn/nn.cpp:224-377) on the device: SGD steps through the same network
// (nn.cpp:59-91) in TRAINING mode — BatchNorm on batch statistics with the running statistics
// updated (momentum 0.1, unbiased variance), the reference's loss (nn.cpp:93-105)
//     L = -sum(obs_p * log(p + 0.001)) + mean((v - obs_v)^2)      (v is [B,256], obs_v broadcasts: Q10)
// and plain SGD p -= lr * dL/dp.
//
// First correct path: fp32 VALU kernels, one launch per operation, every
// reduction done by one workgroup in a fixed order (deterministic).  At the reference's batch of 8
// the work per step is a few MFLOP and the step is launch-bound (~300 launches); batching the
// training set into MFMA kernels is the next step, not this file's.
//
// Parameters live on the device in the canonical blob order (kh_weight_count in kami_hip.h), with
// libtorch's own tensor shapes: conv weight [Co][Ci][3][3] / [Co][Ci][1][1].  Gradients use the same
// layout (the BatchNorm running-statistics slots stay zero), so SGD is one axpy over the blob.
// Activations are [B][64][C] fp32 channels-last like forward_simple.hip.
#include "kh_internal.h"

#include <algorithm>
#include <cstdlib>
#include <vector>

namespace kh {

namespace {

constexpr float BN_EPS = 1e-5f, BN_MOMENTUM = 0.1f;

// ---- conv ----------------------------------------------------------------------------------------
// fp32 VALU kernels, tiled through LDS (the first versions — one thread per output running through global
// memory — spent 92 % of a step in these three: strided weight gathers, one long serial chain per thread).
//
// forward and data gradient are the same loop with the roles swapped:
//   forward : y[b][p][co]  = bias[co] + sum_{tap,ci} x[b][p + off(tap)][ci] * w[co][ci][tap]
//   dgrad   : dx[b][p][ci] (+)=        sum_{tap,co} dy[b][p - off(tap)][co] * w[co][ci][tap]
// One workgroup = one board x OB outputs (8 when the launch would otherwise leave most CUs idle, else 16): the board's input (zero halo, pixel stride K + 1 floats so
// that the 64 pixels of a wave fall in different banks) and the outputs' weights are staged per
// 64-channel slice of the reduction; thread = (pixel, OB / 4 outputs).
constexpr int CV_KC = 64;
template <int T, bool DGRAD, int CV_OB>
__global__ __launch_bounds__(256) void conv_tiled_kernel(const float* __restrict__ w, const float* __restrict__ bias,
                                                         const float* __restrict__ in, float* __restrict__ out,
                                                         int Ci, int Co, int accumulate)
{
    constexpr int NPX = T == 9 ? 100 : 64, KS = CV_KC + 1, CV_PT = CV_OB / 4;
    __shared__ float in_s[NPX * KS];
    __shared__ float w_s[CV_OB * CV_KC * T];
    const int K = DGRAD ? Co : Ci, O = DGRAD ? Ci : Co;
    const int tid = threadIdx.x, b = blockIdx.x, o0 = blockIdx.y * CV_OB;
    const int p = tid & 63, og = tid >> 6, py = p >> 3, px = p & 7;
    for (int i = tid; i < NPX * KS; i += 256) in_s[i] = 0.0f;             // the halo stays zero
    float acc[CV_PT];
#pragma unroll
    for (int j = 0; j < CV_PT; ++j) acc[j] = 0.0f;
    for (int k0 = 0; k0 < K; k0 += CV_KC) {
        const int kn = K - k0 < CV_KC ? K - k0 : CV_KC;
        __syncthreads();                                                   // previous slice consumed (first: zeros written)
        for (int i = tid; i < 64 * CV_KC; i += 256) {
            const int q = i / CV_KC, k = i % CV_KC;
            const int pix = T == 9 ? ((q >> 3) + 1) * 10 + (q & 7) + 1 : q;
            in_s[pix * KS + k] = k < kn ? in[((long)b * 64 + q) * K + k0 + k] : 0.0f;
        }
        for (int i = tid; i < CV_OB * CV_KC * T; i += 256) {
            const int t = i % T, k = (i / T) % CV_KC, ol = i / (T * CV_KC), o = o0 + ol;
            float v = 0.0f;
            if (o < O && k < kn) v = DGRAD ? w[((size_t)(k0 + k) * Ci + o) * T + t] : w[((size_t)o * Ci + k0 + k) * T + t];
            w_s[i] = v;
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            int oy = T == 9 ? t / 3 - 1 : 0, ox = T == 9 ? t % 3 - 1 : 0;
            if (DGRAD) { oy = -oy; ox = -ox; }     // this input pixel p feeds the output pixel p - off(t)
            const float* xi = in_s + (T == 9 ? ((py + oy + 1) * 10 + px + ox + 1) : p) * KS;
            const float* wk = w_s + (size_t)(og * CV_PT) * CV_KC * T + t;
            for (int k = 0; k < kn; ++k) {
                const float xv = xi[k];
#pragma unroll
                for (int j = 0; j < CV_PT; ++j) acc[j] = fmaf(xv, wk[(j * CV_KC + k) * T], acc[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < CV_PT; ++j) {
        const int o = o0 + og * CV_PT + j;
        if (o >= O) continue;
        const long idx = ((long)b * 64 + p) * O + o;
        if (DGRAD) out[idx] = accumulate ? out[idx] + acc[j] : acc[j];
        else out[idx] = acc[j] + bias[o];
    }
}

template <bool DGRAD>
void launch_conv_tiled(const float* w, const float* bias, const float* in, float* out, int B, int Ci, int Co, int T, int accumulate, hipStream_t s)
{
    const int O = DGRAD ? Ci : Co;
    if ((long)B * ((O + 15) / 16) >= 128) {
        const dim3 grid(B, (O + 15) / 16);
        if (T == 9) hipLaunchKernelGGL((conv_tiled_kernel<9, DGRAD, 16>), grid, dim3(256), 0, s, w, bias, in, out, Ci, Co, accumulate);
        else hipLaunchKernelGGL((conv_tiled_kernel<1, DGRAD, 16>), grid, dim3(256), 0, s, w, bias, in, out, Ci, Co, accumulate);
    } else {
        const dim3 grid(B, (O + 7) / 8);
        if (T == 9) hipLaunchKernelGGL((conv_tiled_kernel<9, DGRAD, 8>), grid, dim3(256), 0, s, w, bias, in, out, Ci, Co, accumulate);
        else hipLaunchKernelGGL((conv_tiled_kernel<1, DGRAD, 8>), grid, dim3(256), 0, s, w, bias, in, out, Ci, Co, accumulate);
    }
}

// dw[co][ci][tap] = sum_{b,p} dy[b][p][co] * x[b][p + off(tap)][ci];  db[co] = sum dy[b][p][co]
// One workgroup = WG_T x WG_T (co, ci) pairs — 16 x 16 on 256 threads, or 8 x 8 on ONE WAVE when that is what it
// takes to get 64 workgroups (64 x 64 channels); the sums over the batch stay inside one thread, in a fixed
// order.  Thread = one pair with its T taps in registers; per board the tile's dy channels and (zero-haloed)
// x channels go through LDS.  Sums run b-major, pixel-ascending.
template <int T, int WG_T>
__global__ __launch_bounds__(WG_T * WG_T) void conv_wgrad_tiled_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                              float* __restrict__ dw, float* __restrict__ db, int B, int Ci, int Co)
{
    constexpr int NPX = T == 9 ? 100 : 64, S = WG_T + 1;
    __shared__ float dy_s[64 * S];
    __shared__ float x_s[NPX * S];
    const int tid = threadIdx.x, co0 = blockIdx.x * WG_T, ci0 = blockIdx.y * WG_T;
    const int col = tid / WG_T, cil = tid % WG_T;
    constexpr int NT = WG_T * WG_T;
    for (int i = tid; i < NPX * S; i += NT) x_s[i] = 0.0f;
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0f;
    float bacc = 0.0f;
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        for (int i = tid; i < 64 * WG_T; i += NT) {
            const int q = i / WG_T, c = i % WG_T;
            dy_s[q * S + c] = co0 + c < Co ? dy[((long)b * 64 + q) * Co + co0 + c] : 0.0f;
            const int pix = T == 9 ? ((q >> 3) + 1) * 10 + (q & 7) + 1 : q;
            x_s[pix * S + c] = ci0 + c < Ci ? x[((long)b * 64 + q) * Ci + ci0 + c] : 0.0f;
        }
        __syncthreads();
        for (int q = 0; q < 64; ++q) {
            const float g = dy_s[q * S + col];
            bacc += g;
            const float* xq = x_s + (T == 9 ? ((q >> 3) * 10 + (q & 7)) : q) * S + cil;     // tap (0,0) of the 3x3 window: halo offset folded in
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = fmaf(g, xq[(T == 9 ? (t / 3) * 10 + t % 3 : 0) * S], acc[t]);
        }
    }
    const int co = co0 + col, ci = ci0 + cil;
    if (co < Co && ci < Ci) {
#pragma unroll
        for (int t = 0; t < T; ++t) dw[((size_t)co * Ci + ci) * T + t] = acc[t];
    }
    if (co < Co && cil == 0 && blockIdx.y == 0) db[co] = bacc;
}

// Small layers at small batch (64 x 64 channels, 8 boards): 8 x 8 pairs per workgroup and the 64 pixels of a
// board split over 4 thread groups (16 pixels each), partial sums combined in a fixed order through LDS — 64
// workgroups of 256 threads instead of 16, each thread's serial chain a quarter as long.
template <int T>
__global__ __launch_bounds__(256) void conv_wgrad_split_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               float* __restrict__ dw, float* __restrict__ db, int B, int Ci, int Co)
{
    constexpr int NPX = T == 9 ? 100 : 64, S = 9;
    __shared__ float dy_s[64 * S];
    __shared__ float x_s[NPX * S];
    __shared__ float red[3 * 64 * (T + 1)];
    const int tid = threadIdx.x, co0 = blockIdx.x * 8, ci0 = blockIdx.y * 8;
    const int pg = tid >> 6, pair = tid & 63, col = pair >> 3, cil = pair & 7;
    for (int i = tid; i < NPX * S; i += 256) x_s[i] = 0.0f;
    float acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = 0.0f;
    float bacc = 0.0f;
    for (int b = 0; b < B; ++b) {
        __syncthreads();
        for (int i = tid; i < 64 * 8; i += 256) {
            const int q = i >> 3, c = i & 7;
            dy_s[q * S + c] = co0 + c < Co ? dy[((long)b * 64 + q) * Co + co0 + c] : 0.0f;
            const int pix = T == 9 ? ((q >> 3) + 1) * 10 + (q & 7) + 1 : q;
            x_s[pix * S + c] = ci0 + c < Ci ? x[((long)b * 64 + q) * Ci + ci0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int qq = 0; qq < 16; ++qq) {
            const int q = pg * 16 + qq;
            const float g = dy_s[q * S + col];
            bacc += g;
            const float* xq = x_s + (T == 9 ? ((q >> 3) * 10 + (q & 7)) : q) * S + cil;
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] = fmaf(g, xq[(T == 9 ? (t / 3) * 10 + t % 3 : 0) * S], acc[t]);
        }
    }
    if (pg > 0) {
        float* r = red + ((pg - 1) * 64 + pair) * (T + 1);
#pragma unroll
        for (int t = 0; t < T; ++t) r[t] = acc[t];
        r[T] = bacc;
    }
    __syncthreads();
    if (pg == 0) {
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            const float* r = red + (g * 64 + pair) * (T + 1);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[t] += r[t];
            bacc += r[T];
        }
        const int co = co0 + col, ci = ci0 + cil;
        if (co < Co && ci < Ci) {
#pragma unroll
            for (int t = 0; t < T; ++t) dw[((size_t)co * Ci + ci) * T + t] = acc[t];
        }
        if (co < Co && cil == 0 && blockIdx.y == 0) db[co] = bacc;
    }
}

void launch_conv_wgrad(const float* dy, const float* x, float* dw, float* db, int B, int Ci, int Co, int T, hipStream_t s)
{
    if ((long)((Co + 15) / 16) * ((Ci + 15) / 16) >= 64 || B >= 32) {
        const dim3 grid((Co + 15) / 16, (Ci + 15) / 16);
        if (T == 9) hipLaunchKernelGGL((conv_wgrad_tiled_kernel<9, 16>), grid, dim3(256), 0, s, dy, x, dw, db, B, Ci, Co);
        else hipLaunchKernelGGL((conv_wgrad_tiled_kernel<1, 16>), grid, dim3(256), 0, s, dy, x, dw, db, B, Ci, Co);
    } else {
        const dim3 grid((Co + 7) / 8, (Ci + 7) / 8);
        if (T == 9) hipLaunchKernelGGL(conv_wgrad_split_kernel<9>, grid, dim3(256), 0, s, dy, x, dw, db, B, Ci, Co);
        else hipLaunchKernelGGL(conv_wgrad_split_kernel<1>, grid, dim3(256), 0, s, dy, x, dw, db, B, Ci, Co);
    }
}

// ---- the three convolution GEMMs on the matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32) ----------------------
// forward and data gradient run layers_mfma.hip's conv_f32_kernel (launch_conv_f32_raw): the data gradient of a
// convolution is a convolution of dy with the weights transposed (co <-> ci) and the taps mirrored, so both only
// need the layer's weights as MFMA fragments — re-packed on the device from the canonical blob after every SGD
// step (pack_f32_kernel).  The weight gradient is its own kernel below.
//
// dst[cb][slice][tap][j][ms][lane][i] = W(o = cb*64 + ms*32 + (lane & 31), k = 8j + 4(lane >> 5) + i, tap)
//   forward: W(o, k, tap) = w[o][k][tap]                     (o = co, k = ci)
//   dgrad  : W(o, k, tap) = w[k][o][T - 1 - tap]             (o = ci, k = co; taps mirrored)
// every layer's fragments in one launch (the weights only change at the SGD update): blockIdx.y = job
struct PackJob { const float* src; float* dst; int Co, Ci, T, dgrad; };
constexpr int PACK_JOBS = 40;
struct PackJobs { PackJob j[PACK_JOBS]; };
__global__ __launch_bounds__(256) void pack_f32_jobs_kernel(PackJobs jobs)
{
    const PackJob& jb = jobs.j[blockIdx.y];
    const float* __restrict__ w = jb.src;
    float* __restrict__ dst = jb.dst;
    const int Co = jb.Co, Ci = jb.Ci, T = jb.T, dgrad = jb.dgrad;
    const int O = dgrad ? Ci : Co, K = dgrad ? Co : Ci;
    const int KP = (K + 7) / 8 * 8, OB = (O + 63) / 64;
    const long total = (long)OB * T * (KP / 8) * 2 * 256;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        long r = idx;
        const int i = (int)(r & 3); r >>= 2;
        const int lane = (int)(r & 63); r >>= 6;
        const int ms = (int)(r & 1); r >>= 1;
        const long per_cb = (long)T * (KP / 8);
        const int cb = (int)(r / per_cb);
        long q = r - (long)cb * per_cb;
        int c_lo = 0;
        for (;;) {
            const int cs = KP - c_lo < 128 ? KP - c_lo : 128;
            const long in_slice = (long)T * (cs / 8);
            if (q < in_slice) {
                const int tap = (int)(q / (cs / 8)), j = c_lo / 8 + (int)(q % (cs / 8));
                const int o = cb * 64 + ms * 32 + (lane & 31), k = 8 * j + 4 * (lane >> 5) + i;
                float v = 0.0f;
                if (o < O && k < K) v = dgrad ? w[((size_t)k * Ci + o) * T + (T - 1 - tap)] : w[((size_t)o * Ci + k) * T + tap];
                dst[idx] = v;
                break;
            }
            q -= in_slice; c_lo += 128;
        }
    }
}

inline size_t packed_f32_floats(int O, int K, int T) { return (size_t)((O + 63) / 64) * T * ((K + 7) / 8) * 2 * 256; }

// Weight gradient: dw[co][ci][tap] = sum over boards and pixels of dy[b][p][co] * x[b][p + off(tap)][ci] — per tap a
// [Co x (B*64)] x [(B*64) x Ci] GEMM whose reduction runs over pixels: an MFMA consumes two pixels (K = 2), its A
// operand dy[pixel][co] and its B operand x[shifted pixel][ci] are both read from LDS with the channel on the lane.
// Workgroup = 64 co x 64 ci (wave = 32 x 32, all T taps: 9 accumulators) x a range of boards; the board ranges'
// partial sums go to `part` and are added in range order by wgrad_reduce_kernel (deterministic).
constexpr int WG_PITCH = 12, WG_NPIX = 10 * WG_PITCH, WG_S = 65;
template <int T>
__global__ __launch_bounds__(256) void conv_wgrad_mfma_kernel(const float* __restrict__ dy, const float* __restrict__ x, float* __restrict__ part,
                                                              int B, int Ci, int Co, int xs /* channel stride of x */, int boards_per_split)
{
    using f32x16 = __attribute__((ext_vector_type(16))) float;
    __shared__ float dy_s[64 * WG_S];
    __shared__ float x_s[(T == 9 ? WG_NPIX : 64) * WG_S];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int co0 = blockIdx.x * 64, ci0 = blockIdx.y * 64, split = blockIdx.z;
    const int cq = wave >> 1, iq = wave & 1, r = lane & 31, kk = lane >> 5;
    if (T == 9)
        for (int i = tid; i < WG_NPIX * WG_S; i += 256) x_s[i] = 0.0f;      // the halo stays zero
    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.0f;
    const int b_lo = split * boards_per_split, b_hi = b_lo + boards_per_split < B ? b_lo + boards_per_split : B;
    for (int b = b_lo; b < b_hi; ++b) {
        __syncthreads();
        for (int i = tid; i < 64 * 64; i += 256) {
            const int q = i >> 6, c = i & 63;
            dy_s[q * WG_S + c] = co0 + c < Co ? dy[((long)b * 64 + q) * Co + co0 + c] : 0.0f;
            const int pix = T == 9 ? ((q >> 3) + 1) * WG_PITCH + (q & 7) + 1 : q;
            x_s[pix * WG_S + c] = ci0 + c < Ci ? x[((long)b * 64 + q) * xs + ci0 + c] : 0.0f;
        }
        __syncthreads();
#pragma unroll 4
        for (int p0 = 0; p0 < 64; p0 += 2) {
            const int p = p0 + kk;                                           // this lane's pixel of the pair
            const float av = dy_s[p * WG_S + cq * 32 + r];
            const float* xq = x_s + (T == 9 ? ((p >> 3) * WG_PITCH + (p & 7)) : p) * WG_S + iq * 32 + r;   // tap (0,0) of the window
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, xq[(T == 9 ? (t / 3) * WG_PITCH + t % 3 : 0) * WG_S], acc[t], 0, 0, 0);
        }
    }
    // C/D layout: column (lane & 31) = ci, row (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = co
    float* dst = part + (size_t)split * Co * Ci * T;
    const int ci = ci0 + iq * 32 + r;
    if (ci < Ci) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int co = co0 + cq * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * kk;
            if (co < Co) {
#pragma unroll
                for (int t = 0; t < T; ++t) dst[((size_t)co * Ci + ci) * T + t] = acc[t][reg];
            }
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n, int splits)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float s = part[i];
        for (int k = 1; k < splits; ++k) s += part[(size_t)k * n + i];
        dw[i] = s;
    }
}

// db[co] = sum over boards and pixels of dy: one workgroup per channel, fixed order
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ db, int N, int C)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) s += dy[(long)i * C + c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) db[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

// [B][64][F] -> [B][64][FP] zero-padded
__global__ __launch_bounds__(256) void pad_channels_kernel(const float* __restrict__ in, float* __restrict__ out, long npix, int F, int FP)
{
    const long total = npix * FP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / FP;
        const int c = (int)(i % FP);
        out[i] = c < F ? in[p * F + c] : 0.0f;
    }
}

// ---- BatchNorm (training mode) -----------------------------------------------------------------
__device__ __forceinline__ float block_sum(float v, float* red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per channel: batch mean / biased variance over N = B*64, running statistics updated
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, float* __restrict__ mean, float* __restrict__ invstd,
                                                       float* __restrict__ rm, float* __restrict__ rv, int N, int C)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) s += y[(long)i * C + c];
    const float m = block_sum(s, red) / (float)N;
    float q = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) { const float d = y[(long)i * C + c] - m; q = fmaf(d, d, q); }
    const float var = block_sum(q, red) / (float)N;
    if (threadIdx.x == 0) {
        mean[c] = m;
        invstd[c] = 1.0f / sqrtf(var + BN_EPS);
        rm[c] = (1.0f - BN_MOMENTUM) * rm[c] + BN_MOMENTUM * m;
        rv[c] = (1.0f - BN_MOMENTUM) * rv[c] + BN_MOMENTUM * var * ((float)N / (float)(N - 1));
    }
}

// out = (skip ? skip : 0) + relu(gamma * (y - mean) * invstd + beta)
__global__ __launch_bounds__(256) void bn_relu_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd, const float* __restrict__ g,
                                                          const float* __restrict__ be, const float* __restrict__ skip,
                                                          float* __restrict__ out, long total, int C)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        float z = fmaf(g[c], (y[idx] - mean[c]) * invstd[c], be[c]);
        z = z < 0.0f ? 0.0f : z;
        out[idx] = skip ? skip[idx] + z : z;
    }
}

// one workgroup per channel: dz = dout * (z > 0); dgamma = sum dz * xhat, dbeta = sum dz
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const float* __restrict__ g, const float* __restrict__ be,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int C)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    float sg = 0.0f, sb = 0.0f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const float xh = (y[(long)i * C + c] - mean[c]) * invstd[c];
        const float z = fmaf(g[c], xh, be[c]);
        const float dz = z > 0.0f ? dout[(long)i * C + c] : 0.0f;
        sg = fmaf(dz, xh, sg);
        sb += dz;
    }
    sg = block_sum(sg, red);
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) { dgamma[c] = sg; dbeta[c] = sb; }
}

// dy = gamma * invstd / N * (N * dz - dbeta - xhat * dgamma)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ g, const float* __restrict__ be,
                                                           const float* __restrict__ dgamma, const float* __restrict__ dbeta,
                                                           float* __restrict__ dy, long total, int C, int N)
{
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
        const int c = (int)(idx % C);
        const float xh = (y[idx] - mean[c]) * invstd[c];
        const float z = fmaf(g[c], xh, be[c]);
        const float dz = z > 0.0f ? dout[idx] : 0.0f;
        dy[idx] = g[c] * invstd[c] / (float)N * ((float)N * dz - dbeta[c] - xh * dgamma[c]);
    }
}

// ---- policy head tail: softmax over 4672 + policy loss and its gradient (nn.cpp:78-80, 99-103) ------
// one workgroup per board; loss[b] = -sum obs_p * log(p + 0.001);  dlogit_i = p_i * (g_i - sum_j g_j p_j),
// g_i = -obs_p_i / (p_i + 0.001)
// nanflags[0]: a policy output is NaN (one NaN logit makes the row's log-sum, hence every entry, NaN: nn.cpp:340-341)
__global__ __launch_bounds__(256) void policy_loss_kernel(const float* __restrict__ logits, const float* __restrict__ obsp,
                                                          float* __restrict__ dlogits, float* __restrict__ loss_rows, int* __restrict__ nanflags)
{
    __shared__ float red[4];
    __shared__ float bc;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (size_t)b * KH_PSIZE;
    const float* t = obsp + (size_t)b * KH_PSIZE;
    float m = -INFINITY;
    for (int i = tid; i < KH_PSIZE; i += 256) m = fmaxf(m, x[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float s = 0.0f;
    for (int i = tid; i < KH_PSIZE; i += 256) s += expf(x[i] - m);
    s = block_sum(s, red);
    const float ls = logf(s);
    if (tid == 0 && ls != ls) atomicOr(&nanflags[0], 1);
    float loss = 0.0f, gp = 0.0f;
    for (int i = tid; i < KH_PSIZE; i += 256) {
        const float p = expf((x[i] - m) - ls);
        loss -= t[i] * logf(p + 0.001f);
        gp += -t[i] / (p + 0.001f) * p;
    }
    loss = block_sum(loss, red);
    gp = block_sum(gp, red);
    if (tid == 0) { loss_rows[b] = loss; bc = gp; }
    __syncthreads();
    gp = bc;
    for (int i = tid; i < KH_PSIZE; i += 256) {
        const float p = expf((x[i] - m) - ls);
        dlogits[(size_t)b * KH_PSIZE + i] = p * (-t[i] / (p + 0.001f) - gp);
    }
}

// ---- value head tail: Linear(64,256) + tanh, MSE against the broadcast target (nn.cpp:86-88, 96) ------
// one workgroup per board, thread j = output j: v = tanh(h . W[j] + b[j]);
// dpre[b][j] = 2 (v - obs_v[b]) / (B*256) * (1 - v^2);  sq[b] = sum_j (v - obs_v[b])^2
__global__ __launch_bounds__(256) void value_fwd_loss_kernel(const float* __restrict__ h, const float* __restrict__ fcw,
                                                             const float* __restrict__ fcb, const float* __restrict__ obsv,
                                                             float* __restrict__ dpre, float* __restrict__ sq_rows, int B, int* __restrict__ nanflags)
{
    __shared__ float hs[64];
    __shared__ float red[4];
    const int b = blockIdx.x, j = threadIdx.x;
    if (j < 64) hs[j] = h[(size_t)b * 64 + j];
    __syncthreads();
    float acc = fcb[j];
    for (int k = 0; k < 64; ++k) acc = fmaf(hs[k], fcw[(size_t)j * 64 + k], acc);
    const float v = tanhf(acc), d = v - obsv[b];
    if (v != v) atomicOr(&nanflags[1], 1);           // a value output is NaN (nn.cpp:337-338)
    dpre[(size_t)b * 256 + j] = 2.0f * d / (float)(B * 256) * (1.0f - v * v);
    const float sq = block_sum(d * d, red);
    if (j == 0) sq_rows[b] = sq;
}

// dW[j][k] = sum_b dpre[b][j] h[b][k]; db[j] = sum_b dpre[b][j]   (one thread per (j,k))
__global__ __launch_bounds__(256) void fc_wgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ h,
                                                       float* __restrict__ dw, float* __restrict__ db, int B)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 256 * 64) return;
    const int j = idx / 64, k = idx % 64;
    float acc = 0.0f, bacc = 0.0f;
    for (int b = 0; b < B; ++b) { const float g = dpre[(size_t)b * 256 + j]; acc = fmaf(g, h[(size_t)b * 64 + k], acc); bacc += g; }
    dw[idx] = acc;
    if (k == 0) db[j] = bacc;
}

// dh[b][k] = sum_j dpre[b][j] W[j][k]
__global__ __launch_bounds__(64) void fc_dgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ fcw,
                                                      float* __restrict__ dh)
{
    const int b = blockIdx.x, k = threadIdx.x;
    float acc = 0.0f;
    for (int j = 0; j < 256; ++j) acc = fmaf(dpre[(size_t)b * 256 + j], fcw[(size_t)j * 64 + k], acc);
    dh[(size_t)b * 64 + k] = acc;
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float lr, long n)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = fmaf(-lr, g[i], p[i]);
}

inline int nblocks(long total) { long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b)); }

struct ConvBNOff { size_t w, b, g, be, rm, rv; int Ci, Co, T; };

}  // namespace

struct TrainNet {
    int F, C, R;
    ConvBNOff stem, pconv, vconv;
    std::vector<ConvBNOff> res;
    size_t p2w, p2b, fcw, fcb, total;
};

static TrainNet layout(int F, int C, int R)
{
    TrainNet n;
    n.F = F; n.C = C; n.R = R;
    size_t o = 0;
    auto convbn = [&](int Ci, int Co, int T) {
        ConvBNOff c;
        c.Ci = Ci; c.Co = Co; c.T = T;
        c.w = o; o += (size_t)Co * Ci * T;
        c.b = o; o += Co; c.g = o; o += Co; c.be = o; o += Co; c.rm = o; o += Co; c.rv = o; o += Co;
        return c;
    };
    n.stem = convbn(F, C, 9);
    for (int i = 0; i < 2 * R; ++i) n.res.push_back(convbn(C, C, 9));
    n.pconv = convbn(C, KH_POLICY_MID, 1);
    n.p2w = o; o += (size_t)KH_POLICY_PLANES * KH_POLICY_MID;
    n.p2b = o; o += KH_POLICY_PLANES;
    n.vconv = convbn(C, 1, 1);
    n.fcw = o; o += (size_t)KH_VALUE_WIDTH * 64;
    n.fcb = o; o += KH_VALUE_WIDTH;
    n.total = o;
    return n;
}

// One SGD step on device buffers (StepBuffers: params = blob updated in place, grads = blob-shaped,
// work = train_workspace_floats() floats of activations, conv outputs, statistics and gradients).

// board ranges of the weight gradient: enough workgroups to fill the chip, at least one board each
static int train_wgrad_splits(int B, int Ci, int Co)
{
    const int tiles = ((Co + 63) / 64) * ((Ci + 63) / 64);
    int want = (256 + tiles - 1) / tiles;
    if (want > B) want = B;
    if (want < 1) want = 1;
    const int per = (B + want - 1) / want;
    return (B + per - 1) / per;
}

// floats of the weight gradient's partial-sum buffer: the largest layer's (ranges x Co x Ci x taps)
static size_t train_wgrad_part_floats(int F, int C, int B)
{
    const int FP = (F + 7) / 8 * 8;
    size_t m = (size_t)train_wgrad_splits(B, FP, C) * C * FP * 9;                                    // stem
    m = std::max(m, (size_t)train_wgrad_splits(B, C, C) * C * C * 9);                               // tower
    m = std::max(m, (size_t)train_wgrad_splits(B, C, KH_POLICY_MID) * KH_POLICY_MID * C);           // policyconv
    return m;
}

size_t train_workspace_floats(int F, int C, int R, int B)
{
    const size_t act = (size_t)B * 64 * (size_t)(C > KH_POLICY_MID ? C : KH_POLICY_MID);
    const int L = 1 + 2 * R;
    // per tower layer: output activation + conv output; heads: pconv y/act, logits, dlogits, value pieces; 3 gradient planes
    size_t n = act * (2 * (size_t)L + 12) + (size_t)B * 64 * F + (size_t)B * KH_PSIZE * 2 + (size_t)(2 * L + 8) * 2 * 256 + 65536;
    // MFMA path: the stem input padded to a multiple of 8 planes, every tower / policy-conv layer's weights as forward
    // and data-gradient fragments, the weight gradient's per-board-range partial sums
    const int FP = (F + 7) / 8 * 8, CM = C > KH_POLICY_MID ? C : KH_POLICY_MID;
    n += (size_t)B * 64 * FP;
    n += packed_f32_floats(C, FP, 9) + (size_t)2 * R * 2 * packed_f32_floats(C, C, 9) + 2 * packed_f32_floats(CM, CM, 1);
    n += train_wgrad_part_floats(F, C, B) + 65536;
    return n;
}

hipError_t train_step(const TrainNet& n, const StepBuffers& sb, const float* x_in, const float* obsp, const float* obsv,
                      int B, float lr, float* loss_rows /* [2*B] device: policy rows, value squared-error rows; behind them two ints:
                      a policy / value output of this step's forward is NaN */, hipStream_t s)
{
    int* nanflags = reinterpret_cast<int*>(loss_rows + 2 * B);
    (void)hipMemsetAsync(nanflags, 0, 2 * sizeof(int), s);
    const int C = n.C, N = B * 64;
    float* P = sb.params;
    float* G = sb.grads;
    float* wk = sb.work;
    auto take = [&](size_t nfl) { float* r = wk; wk += nfl; return r; };
    const size_t actC = (size_t)B * 64 * C;
    struct Saved { const float* in; float* y; float* out; float* mean; float* invstd; int in_stride; const float* wf; const float* wd; };
    std::vector<Saved> sv;
    (void)hipMemsetAsync(G, 0, n.total * sizeof(float), s);
    // matrix-core path (exact fp32 MFMA) for the layers that carry the work: channel counts in multiples of 8
    // (KAMI_TRAIN_VALU=1 keeps every convolution on the order-exact VALU kernels above)
    const bool valu_only = getenv("KAMI_TRAIN_VALU") && atoi(getenv("KAMI_TRAIN_VALU")) != 0;   // read when a step is recorded
    auto on_mfma = [&](const ConvBNOff& c) { return !valu_only && c.Co % 8 == 0 && c.Co >= 8 && (c.Ci % 8 == 0 || &c == &n.stem); };
    const int FP = (n.F + 7) / 8 * 8;
    const float* x_pad = x_in;
    if (on_mfma(n.stem) && FP != n.F) {
        float* xp = take((size_t)N * FP);
        hipLaunchKernelGGL(pad_channels_kernel, dim3(nblocks((long)N * FP)), dim3(256), 0, s, x_in, xp, (long)N, n.F, FP);
        x_pad = xp;
    }
    float* wpart = nullptr;
    // every MFMA layer's weights as forward / data-gradient fragments, packed up front in as few launches as it takes
    struct Packed { const float* wf = nullptr; const float* wd = nullptr; };
    std::vector<const ConvBNOff*> mlayers;
    mlayers.push_back(&n.stem);
    for (auto& c : n.res) mlayers.push_back(&c);
    mlayers.push_back(&n.pconv);
    std::vector<Packed> packed(mlayers.size());
    {
        PackJobs jobs;
        int nj = 0;
        long biggest = 0;
        auto flush = [&]() {
            if (!nj) return;
            int bx = (int)((biggest + 255) / 256);
            if (bx > 256) bx = 256;
            hipLaunchKernelGGL(pack_f32_jobs_kernel, dim3(bx, nj), dim3(256), 0, s, jobs);
            nj = 0; biggest = 0;
        };
        auto add = [&](const float* src, float* dst, int Co, int Ci, int T, int dgrad, size_t count) {
            jobs.j[nj++] = PackJob{ src, dst, Co, Ci, T, dgrad };
            biggest = std::max(biggest, (long)count);
            if (nj == PACK_JOBS) flush();
        };
        for (size_t i = 0; i < mlayers.size(); ++i) {
            const ConvBNOff& c = *mlayers[i];
            if (!on_mfma(c)) continue;
            const int CiP = (c.Ci + 7) / 8 * 8;
            float* wf = take(packed_f32_floats(c.Co, CiP, c.T));
            add(P + c.w, wf, c.Co, c.Ci, c.T, 0, packed_f32_floats(c.Co, CiP, c.T));
            packed[i].wf = wf;
            if (&c != &n.stem) {                     // the stem's input needs no gradient
                float* wd = take(packed_f32_floats(c.Ci, c.Co, c.T));
                add(P + c.w, wd, c.Co, c.Ci, c.T, 1, packed_f32_floats(c.Ci, c.Co, c.T));
                packed[i].wd = wd;
            }
        }
        flush();
    }
    auto packed_of = [&](const ConvBNOff& c) -> const Packed& {
        for (size_t i = 0; i < mlayers.size(); ++i)
            if (mlayers[i] == &c) return packed[i];
        static const Packed none;
        return none;
    };

    auto fwd = [&](const ConvBNOff& c, const float* in, const float* skip) {
        Saved v;
        v.in = in; v.in_stride = c.Ci; v.wf = v.wd = nullptr;
        v.y = take((size_t)N * c.Co); v.out = take((size_t)N * c.Co); v.mean = take(256); v.invstd = take(256);
        if (on_mfma(c)) {
            const int CiP = (c.Ci + 7) / 8 * 8;
            if (&c == &n.stem) { v.in = x_pad; v.in_stride = CiP; }
            const Packed& pk = packed_of(c);
            v.wf = pk.wf; v.wd = pk.wd;
            (void)launch_conv_f32_raw(v.in, v.wf, P + c.b, v.y, B, CiP, c.Co, c.T, false, s);
        } else
        launch_conv_tiled<false>(P + c.w, P + c.b, in, v.y, B, c.Ci, c.Co, c.T, 0, s);
        hipLaunchKernelGGL(bn_stats_kernel, dim3(c.Co), dim3(256), 0, s, v.y, v.mean, v.invstd, P + c.rm, P + c.rv, N, c.Co);
        hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(nblocks((long)N * c.Co)), dim3(256), 0, s, v.y, v.mean, v.invstd, P + c.g, P + c.be, skip, v.out,
                           (long)N * c.Co, c.Co);
        sv.push_back(v);
        return v.out;
    };
    // dout: gradient w.r.t. relu(bn(conv(in))) ; writes parameter gradients, returns dL/d(in) in `din` (accumulated if acc)
    auto bwd = [&](const ConvBNOff& c, const Saved& v, const float* dout, float* dy_tmp, float* din, int acc) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(c.Co), dim3(256), 0, s, dout, v.y, v.mean, v.invstd, P + c.g, P + c.be, G + c.g, G + c.be, N, c.Co);
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(nblocks((long)N * c.Co)), dim3(256), 0, s, dout, v.y, v.mean, v.invstd, P + c.g, P + c.be,
                           G + c.g, G + c.be, dy_tmp, (long)N * c.Co, c.Co, N);
        if (v.wf) {
            const int splits = train_wgrad_splits(B, c.Ci, c.Co), per = (B + splits - 1) / splits;
            const size_t nw = (size_t)c.Co * c.Ci * c.T;
            const dim3 grid((c.Co + 63) / 64, (c.Ci + 63) / 64, splits);
            float* part = splits > 1 ? wpart : G + c.w;
            if (c.T == 9) hipLaunchKernelGGL(conv_wgrad_mfma_kernel<9>, grid, dim3(256), 0, s, dy_tmp, v.in, part, B, c.Ci, c.Co, v.in_stride, per);
            else hipLaunchKernelGGL(conv_wgrad_mfma_kernel<1>, grid, dim3(256), 0, s, dy_tmp, v.in, part, B, c.Ci, c.Co, v.in_stride, per);
            if (splits > 1) hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(nblocks((long)nw)), dim3(256), 0, s, part, G + c.w, (long)nw, splits);
            hipLaunchKernelGGL(bias_grad_kernel, dim3(c.Co), dim3(256), 0, s, dy_tmp, G + c.b, N, c.Co);
            if (din) (void)launch_conv_f32_raw(dy_tmp, v.wd, nullptr, din, B, c.Co, c.Ci, c.T, acc != 0, s);
            return;
        }
        launch_conv_wgrad(dy_tmp, v.in, G + c.w, G + c.b, B, c.Ci, c.Co, c.T, s);
        if (din)
            launch_conv_tiled<true>(P + c.w, nullptr, dy_tmp, din, B, c.Ci, c.Co, c.T, acc, s);
    };

    // ---- forward (nn.cpp:59-91, module in train mode) ----
    const float* x = fwd(n.stem, x_in, nullptr);
    for (int r = 0; r < n.R; ++r) {
        const float* t = fwd(n.res[2 * r], x, nullptr);
        x = fwd(n.res[2 * r + 1], t, x);                    // x + relu(bn2(conv2(...)))   nn.cpp:26-34
    }
    const size_t tower_saved = sv.size();
    const float* pm = fwd(n.pconv, x, nullptr);             // [B][64][128]
    float* logits = take((size_t)B * KH_PSIZE);
    launch_conv_tiled<false>(P + n.p2w, P + n.p2b, pm, logits, B, KH_POLICY_MID, KH_POLICY_PLANES, 1, 0, s);     // [B][64][73] = index pixel*73 + plane (nn.cpp:78-79)
    const float* h = fwd(n.vconv, x, nullptr);              // [B][64][1]

    // ---- losses and their gradients ----
    float* dlogits = take((size_t)B * KH_PSIZE);
    hipLaunchKernelGGL(policy_loss_kernel, dim3(B), dim3(256), 0, s, logits, obsp, dlogits, loss_rows, nanflags);
    float* dpre = take((size_t)B * 256);
    hipLaunchKernelGGL(value_fwd_loss_kernel, dim3(B), dim3(256), 0, s, h, P + n.fcw, P + n.fcb, obsv, dpre, loss_rows + B, B, nanflags);

    // ---- backward ----
    wpart = take(train_wgrad_part_floats(n.F, C, B));
    float* dX = take(actC);             // gradient w.r.t. the tower output, then walked down the tower
    float* dT = take((size_t)N * (C > KH_POLICY_MID ? C : KH_POLICY_MID));
    float* dtmp = take((size_t)N * (C > KH_POLICY_MID ? C : KH_POLICY_MID));
    // value head: fc -> relu/bn/conv
    float* dh = take((size_t)B * 64);
    hipLaunchKernelGGL(fc_wgrad_kernel, dim3(64), dim3(256), 0, s, dpre, h, G + n.fcw, G + n.fcb, B);
    hipLaunchKernelGGL(fc_dgrad_kernel, dim3(B), dim3(64), 0, s, dpre, P + n.fcw, dh);
    bwd(n.vconv, sv[tower_saved + 1], dh, dtmp, dX, 0);
    // policy head: conv2 (plain) -> relu/bn/conv
    launch_conv_wgrad(dlogits, pm, G + n.p2w, G + n.p2b, B, KH_POLICY_MID, KH_POLICY_PLANES, 1, s);
    launch_conv_tiled<true>(P + n.p2w, nullptr, dlogits, dT, B, KH_POLICY_MID, KH_POLICY_PLANES, 1, 0, s);
    bwd(n.pconv, sv[tower_saved], dT, dtmp, dX, 1);
    // tower
    for (int r = n.R - 1; r >= 0; --r) {
        // out = xin + relu(bn2(conv2(t))):  d t = through conv2;  d xin = dX (skip) + through conv1
        bwd(n.res[2 * r + 1], sv[2 + 2 * r], dX, dtmp, dT, 0);       // dT = dL/dt
        bwd(n.res[2 * r], sv[1 + 2 * r], dT, dtmp, dX, 1);           // dX += dL/dxin via conv1
    }
    bwd(n.stem, sv[0], dX, dtmp, nullptr, 0);

    hipLaunchKernelGGL(sgd_kernel, dim3(nblocks((long)n.total)), dim3(256), 0, s, P, G, lr, (long)n.total);
    return hipGetLastError();
}

TrainNet* train_layout_new(int F, int C, int R) { return new TrainNet(layout(F, C, R)); }
void train_layout_free(TrainNet* n) { delete n; }

}  // namespace kh
