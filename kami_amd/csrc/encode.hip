// encode.hip — board -> plane encoding on gfx950.
//
// Replaces Env::observe (kami/env.h:202-262) for a batch of compact records (kh_board).
// Pure integer work, bit-exact; HBM-write bound (7 680 B written per position, 80 B read).
//
// One wavefront (64 lanes) per position, lane = POV square:
//   phase 1  every lane builds its square's 30 channel values in registers and writes them
//            to the wave's LDS tile [64][30] with 8-byte stores;
//   phase 2  the tile is streamed to HBM as 480 fully coalesced 16-byte stores.
// The record itself is wave-uniform and is fetched with scalar loads.
#include "kh_internal.h"
#include "encode_square.h"

namespace kh {

constexpr int ENC_WAVES = 4;                 // waves per workgroup
constexpr int ENC_TILE = 64 * KH_NFEATURES;  // floats per position

template <bool WT>
__global__ __launch_bounds__(64 * ENC_WAVES) void encode_f32_kernel(const kh_board* __restrict__ boards,
                                                                    int n, float* __restrict__ planes)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* tile = reinterpret_cast<float*>(smem) + wave * ENC_TILE;
    const int stride = gridDim.x * ENC_WAVES;

    for (int base = blockIdx.x * ENC_WAVES; base < n; base += stride) {
        const int b = base + wave;            // wave-uniform
        if (b < n) {
            float v[KH_NFEATURES];
            encode_square(boards + b, lane, v);

            float2* dst = reinterpret_cast<float2*>(tile + lane * KH_NFEATURES);         // 120 B rows: 8-B aligned
#pragma unroll
            for (int i = 0; i < KH_NFEATURES / 2; ++i) dst[i] = make_float2(v[2 * i], v[2 * i + 1]);
        }
        __syncthreads();
        if (b < n) {
            using f4 = float __attribute__((ext_vector_type(4)));
            const f4* src = reinterpret_cast<const f4*>(tile);
            f4* out = reinterpret_cast<f4*>(planes + (size_t)b * ENC_TILE);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int q = lane + 64 * j;
                if (q < ENC_TILE / 4) {
                    // write-once stream.  Up to the Infinity Cache's size write-through (sc1) stores win: nothing is left
                    // dirty in L2 for the end-of-kernel release (8 192 positions: 9.7 us against 11.1 non-temporal, 10.8
                    // plain); beyond it non-temporal stores do (2^20 positions: 5.89 TB/s against 5.77 / 5.67) —
                    // profiles/r03_encode_store_ab.txt
                    if (WT) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(&out[q]), "v"(src[q]) : "memory");
                    else __builtin_nontemporal_store(src[q], &out[q]);
                }
            }
        }
        __syncthreads();
    }
}

void launch_encode_f32(const kh_board* d_boards, int n, float* d_planes, hipStream_t s)
{
    if (n <= 0) return;
    int wgs = (n + ENC_WAVES - 1) / ENC_WAVES;
    if (wgs > 256 * 5) wgs = 256 * 5;        // 30 KB LDS per workgroup -> 5 resident per CU
    if ((size_t)n * ENC_TILE * sizeof(float) <= ((size_t)128 << 20))
        hipLaunchKernelGGL(encode_f32_kernel<true>, dim3(wgs), dim3(64 * ENC_WAVES), ENC_WAVES * ENC_TILE * sizeof(float), s, d_boards, n, d_planes);
    else
        hipLaunchKernelGGL(encode_f32_kernel<false>, dim3(wgs), dim3(64 * ENC_WAVES), ENC_WAVES * ENC_TILE * sizeof(float), s, d_boards, n, d_planes);
}

// ---------------------------------------------------------------------------------------------
// Legal-move policy gather (MCTS::expand, kami/mcts.h:273-276,296): one wavefront per position,
// lanes stride over the position's legal actions (<= 128 in the reference, any count here).
__global__ __launch_bounds__(256) void gather_legal_kernel(const float* __restrict__ policy,
                                                           const int32_t* __restrict__ offsets,
                                                           const int32_t* __restrict__ actions,
                                                           float* __restrict__ priors, int B,
                                                           const float* __restrict__ vfull, int vstride,
                                                           float* __restrict__ values, const int* __restrict__ flags_in,
                                                           int* __restrict__ flags_out)
{
    const int lane = threadIdx.x & 63;
    const int wpb = blockDim.x >> 6;
    // the host path's compact results: one value per position and the NaN flags behind the priors, so
    // that one copy brings everything back
    if (values) {
        for (int b = blockIdx.x * blockDim.x + threadIdx.x; b < B; b += gridDim.x * blockDim.x) values[b] = vfull[(size_t)b * vstride];
        if (blockIdx.x == 0 && threadIdx.x < 4) flags_out[threadIdx.x] = flags_in[threadIdx.x];
    }
    if (!offsets) return;                       // values + flags only (the zero-copy small-batch path of kh_infer)
    for (int b = blockIdx.x * wpb + (threadIdx.x >> 6); b < B; b += gridDim.x * wpb) {
        const int lo = offsets[b], hi = offsets[b + 1];
        const float* p = policy + (size_t)b * KH_PSIZE;
        float sum = 0.0f;
        for (int k = lo + lane; k < hi; k += 64) {
            const int a = actions[k];
            sum += (a >= 0 && a < KH_PSIZE) ? p[a] : 0.0f;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
        for (int k = lo + lane; k < hi; k += 64) {
            const int a = actions[k];
            priors[k] = (a >= 0 && a < KH_PSIZE) ? p[a] * inv : 0.0f;
        }
    }
}

void launch_gather_legal(const float* policy, const int32_t* offsets, const int32_t* actions,
                         float* priors, int B, hipStream_t s, const float* vfull, int vstride, float* values,
                         const int* flags_in, int* flags_out)
{
    if (B <= 0) return;
    int blocks = (B + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gather_legal_kernel, dim3(blocks), dim3(256), 0, s, policy, offsets, actions, priors, B, vfull, vstride, values,
                       flags_in, flags_out);
}

// One word into page-locked host memory when everything before it on the stream has finished: the queue's dispatcher
// polls that word instead of hipStreamQuery (tools/completion_probe.hip: the word is there 6-8 us before the runtime's
// own completion signal has been processed).  Stream order makes the results of the kernels before it complete; they
// and this word are posted writes of the same device to the same host.
__global__ void signal_kernel(unsigned* host_word, unsigned serial)
{
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_store(host_word, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

void launch_signal(unsigned* host_word, unsigned serial, hipStream_t s)
{
    hipLaunchKernelGGL(signal_kernel, dim3(1), dim3(64), 0, s, host_word, serial);
}

}  // namespace kh
