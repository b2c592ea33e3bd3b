// layers_mfma.hip — bf16 / f16 forward pass for WIDE nets (65..256 filters) and the exact-fp32 path, on the matrix
// cores.  The whole-network kernel (tower_mfma.hip) is specialised for <= 64 filters; from 128 filters up a 3x3 layer is
// >= 4x the work, one layer's weights (up to 1.2 MB) and two boards' activations no longer fit a CU the same way, and
// the tiling is chosen per width and batch (run<T>() at the end of this file; DESIGN.md 5.3 has the anatomy, the
// measurements and what was tried and dropped):
//
//   conv_mfma_kernel      one launch per layer, workgroup = 2 boards x 64 output channels, weights through an LDS ring
//                         fed by LDS-DMA, activations T [B][64][C] (channels-last) in HBM / L2.  Any width 65..256.
//   conv4_mfma_kernel     per layer, 4 boards x 128 channels, a wave owns one board (0.75 LDS reads per MFMA).
//   tower128_kernel       128 filters: the whole 3x3 stack in ONE launch, conv4's tiling, wave-local layer boundaries
//                         (no barrier, no HBM between layers); reads the fp32 planes itself.  BASELINE configs[2].
//   tower2b_kernel<T,CH>  256 (and 128) filters: the whole stack in one launch, 2 boards per workgroup, waves own output
//                         channels, weights global -> registers (no ring), two barriers per boundary.  configs[4]'s net.
//   policy_head4_kernel   policy 1x1 convs + softmax + value head of four boards per workgroup in one launch.
//   conv_f32_kernel / conv_f32_small_kernel   v_mfma_f32_32x32x2_f32 (an exact fp32 fmaf chain): dtype = f32 inference
//                         and the trainer's forward / data gradient (train.hip).
//
// All bf16 / f16 variants walk the reduction in the same order (64-channel slices, taps, k-steps) with the same fp32
// epilogue, so they agree BIT FOR BIT and the launcher may pick per call (tests/test_gpu_parity.py compares them).
// Geometry shared with the tower kernel: input images with a zero halo, pixel stride 2*Ci + 16 bytes, row pitch 12
// (conflict-free ds_read_b128 groups for all nine taps).
#include "kh_internal.h"

#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace kh {
namespace lay {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using f32x4s = __attribute__((ext_vector_type(4))) float;

#ifndef KAMI_WIDE_BUFA
#define KAMI_WIDE_BUFA 1           // tower2b / tower2s weight fragments by buffer loads (0: global loads, the A/B baseline)
#endif
// the policy heads' softmax (policy_head4_kernel and tower128_kernel<T, true>: one arithmetic, every batch size): v_exp_f32 /
// v_log_f32 forms (2 ulp) instead of ocml's expf / logf — 256 calls per lane were 8-10 us of a 28 us head
#ifndef KAMI_HEAD_FAST_EXP
#define KAMI_HEAD_FAST_EXP 1
#endif
#if KAMI_HEAD_FAST_EXP
#define HEAD_EXP(x) __expf(x)
#define HEAD_LOG(x) __logf(x)
#else
#define HEAD_EXP(x) expf(x)
#define HEAD_LOG(x) logf(x)
#endif
#ifndef KAMI_T2B_NA
#define KAMI_T2B_NA 12           // tower2b_kernel: k-steps of weight fragments in flight (4 -> 12: 10x128 at batch 512 198 -> 186 us, 20x256 +0.5 %)
#endif
#ifndef KAMI_T2S_NA
#define KAMI_T2S_NA 12           // tower2s_kernel: the same (4: 653 us, 6: 626, 9: 623, 12: 620 at 20x256 batch 256: the movers' sc1 traffic delays the L2)
#endif
#ifndef KAMI_SPLIT_MIN_B
#define KAMI_SPLIT_MIN_B 1           // 256 filters: smallest batch on tower2s_kernel (against the per-layer kernels, 20x256 f16: 439 vs 480 us at
                                    // batch 1, 449 vs 499 at 32, 456 vs 564 at 128, 641 vs 972 at 254: it wins everywhere, so one kernel family — one
                                    // summation order — serves every batch up to 256)
#endif
constexpr int PITCH = 12;
constexpr int NPIX = 10 * PITCH;

// same lane -> pixel map as tower_mfma.hip (conflict-free ds_read_b128 groups with PITCH 12)
__device__ __constant__ const unsigned char PIXMAP[32] = {
    0, 1, 2, 3, 8, 9, 10, 11, 12, 13, 14, 15, 4, 5, 6, 7,
    24, 25, 26, 27, 16, 17, 18, 19, 20, 21, 22, 23, 28, 29, 30, 31
};

template <typename T> struct Elem;
template <> struct Elem<__bf16> {
    using vec8 = bf16x8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Elem<_Float16> {
    using vec8 = f16x8;
    static __device__ __forceinline__ f32x16 mfma(vec8 a, vec8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <typename T> __device__ __forceinline__ unsigned short to_bits(float v)
{
    const T t = (T)v;
    return __builtin_bit_cast(unsigned short, t);
}
template <typename T> __device__ __forceinline__ float from_bits(unsigned short u) { return (float)__builtin_bit_cast(T, u); }
__device__ __forceinline__ float relu_keep_nan(float v) { return v < 0.0f ? 0.0f : v; }   // torch::relu semantics
// two floats -> one register of two T (round to nearest even, the same values as two to_bits<T>): ONE
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 instead of two scalar converts and a merge
template <typename T> __device__ __forceinline__ unsigned pack2(float a, float b)
{
    using f32x2 = __attribute__((ext_vector_type(2))) float;
    using t2 = __attribute__((ext_vector_type(2))) T;
    const t2 r = __builtin_convertvector(f32x2{ a, b }, t2);
    return __builtin_bit_cast(unsigned, r);
}

// fp32 planes [B][64][F] -> T [B][64][FP] (channels >= F zero).  One thread per 8-channel chunk.
template <typename T>
__global__ __launch_bounds__(256) void planes_to_act_kernel(const float* __restrict__ in, unsigned short* __restrict__ out,
                                                            long npix, int F, int FP)
{
    const int CH = FP / 8;
    const long total = npix * CH;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / CH;
        const int c0 = (int)(i % CH) * 8;
        const float* src = in + p * F + c0;
        unsigned short v[8];
        for (int k = 0; k < 8; ++k) v[k] = to_bits<T>((c0 + k < F) ? src[k] : 0.0f);
        u32x4 o;
        o.x = v[0] | ((unsigned)v[1] << 16); o.y = v[2] | ((unsigned)v[3] << 16);
        o.z = v[4] | ((unsigned)v[5] << 16); o.w = v[6] | ((unsigned)v[7] << 16);
        *reinterpret_cast<u32x4*>(out + p * FP + c0) = o;
    }
}

#ifdef KAMI_WIDE_DIAG
// diagnostic build only (tools/wide_stamps.py): where a workgroup of the 3x3 skip layer spends its time
__device__ unsigned long long g_wide_stamps[2048 * 8];
#define WIDE_STAMP(k) do { if (TAPS == 9 && EPI == 1 && tid == 0 && blockIdx.x + gridDim.x * blockIdx.y < 2048) \
    g_wide_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 8 + (k)] = (k) == 7 ? (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) : __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WIDE_STAMP(k) do {} while (0)
#endif

struct ConvArgs {
    const unsigned short* in;     // T [B][64][Ci]
    const unsigned short* w;      // packed fragments [Co/64][taps][Ci/16][2][64 lanes][8]
    const float* shift;           // [Co] folded BatchNorm shift (or bias)
    const unsigned short* w4;     // the same layer packed for conv4_mfma_kernel (nullptr: not eligible)
    const unsigned short* skip;   // T [B][64][Co] (EPI 1) or nullptr
    void* out;                    // T [B][64][Co] (EPI 0/1) or fp32 [B][4672] (EPI 2)
    int B, Ci, Co;                // Ci % 16 == 0, Co % 64 == 0
};

// ---- weight stream of one (layer, 64-channel block): 8 KB chunks (4 k-steps x 2 row tiles) through a
// 4-slot LDS ring by LDS-DMA, exactly the tower kernel's protocol (tower_mfma.hip: pipe_step) with
// D = 4: the 4 waves of a workgroup would otherwise each pull the same fragments through the
// vector L1 (128 B/clk wanted, 64 available).  The ring sits at LDS offset 0 (M0's 16-bit field).
constexpr int RD = 4;
constexpr int CHUNKB = 8192;
constexpr int LDS_IMG = RD * CHUNKB;

// the same for a chunk given by its address, into slot `slot` (a second stream joining the first: tower128_kernel's heads)
__device__ __forceinline__ void ring_issue_at(const char* chunk, int slot, int wave, int lane)
{
    const char* sbase = chunk + wave * 2048;
    const unsigned dst = (unsigned)(slot * CHUNKB + wave * 2048);
    const unsigned voff = lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

template <int RDN = RD>
__device__ __forceinline__ void ring_issue(const char* stream, int nch, int c, int wave, int lane)
{
    const int cs = c < nch ? c : nch - 1;                  // past the end: re-fetch the last chunk into a
    const char* sbase = stream + (size_t)cs * CHUNKB + wave * 2048;   // free slot (keeps the vmcnt count constant)
    const unsigned dst = (unsigned)((c % RDN) * CHUNKB + wave * 2048);
    const unsigned voff = lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

// EPI: 0 = ReLU -> T;  1 = ReLU, + skip -> T (nn.cpp:31);  2 = raw fp32 logits, planes < 73 (nn.cpp:75-79)
// CPT = Ci / 64 (8 KB weight chunks per tap): a template parameter so that the whole chunk loop unrolls
// with compile-time image offsets and register-set parity, like the tower kernel's layers — with a
// run-time loop hipcc rotated the double buffer through v_mov copies and issued the next chunk's
// activation reads (behind a run-time address computation) at the END of a step, exposing their
// latency after every barrier: 2.6x the tower's time per step.
// RDN = ring slots.  RDN == 2 (3x3 layers with 128 input channels): ring + image = 81 664 B, so TWO workgroups
// share a CU (256 registers each) and one's staging / epilogue runs under the other's MFMAs — with one
// workgroup per CU all CUs stage at the same moment (42 % of a workgroup's time at 128 channels,
// tools/wide_stamps.py) and then leave the memory system idle.  A 2-slot ring refills the slot of the chunk
// that is already in registers (AHEAD), so its reads must have completed before the step's barrier.
// NP = passes over the input channels, CPT * 64 of them each: the image of one pass is 1/NP the size (256
// channels as 2 x 128: two workgroups per CU there too; passes of 64 channels: image + ring = 50 944 B, THREE
// workgroups per CU at 168 registers).  Weights are packed in 64-channel slices (kh_api.hip: pack_layer_generic),
// the order every variant walks (chunk_off).
template <typename T, int TAPS, int EPI, int CPT, int RDN = RD, int NP = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, RDN == 2 ? (CPT == 1 ? 3 : 2) : 1))) void conv_mfma_kernel(ConvArgs a)
{
    constexpr int AHEAD = RDN == 2 ? 1 : 0;
    constexpr int LDS_IMG = RDN * CHUNKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int Ci = CPT * 64;                           // channels of one pass; a chunk never straddles taps
    constexpr int CiTot = NP * Ci;
    const int Co = a.Co;
    constexpr int stride = Ci * 2 + 16;
    constexpr int npx = (TAPS == 9) ? NPIX : 64;          // pixels per board image
    constexpr int board_bytes = npx * stride;
    const int b0 = blockIdx.x * 2, cb = blockIdx.y;
    constexpr int NCH = TAPS * CPT, NCHT = NP * NCH;
    const char* stream = reinterpret_cast<const char*>(a.w) + (size_t)cb * NCHT * CHUNKB;
    char* img = smem + LDS_IMG;
    WIDE_STAMP(0);
    WIDE_STAMP(7);

    // weight stream first: the ring fills while the boards are staged
#pragma unroll
    for (int i = 0; i < RDN - 1 + AHEAD; ++i) ring_issue<RDN>(stream, NCHT, i, wave, lane);

    // ---- zero halo of the two boards' images (staging only ever writes interior pixels)
    if (TAPS == 9) {
        const u32x4 z = { 0, 0, 0, 0 };
        constexpr int per_px = stride / 16;
        for (int i = tid; i < 2 * NPIX; i += 256) {
            const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
            if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
            char* d = img + (i / NPIX) * board_bytes + pp * stride;
            for (int k = 0; k < per_px; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
        }
    }
    // ---- stage the two boards' input channels [pass * Ci, (pass + 1) * Ci) in LDS
    auto stage = [&](int pass) {
        // A thread owns 4 loads of 16 bytes per 64 input channels.  Up to 128 channels: batches of four in
        // flight, then their LDS writes; from 192 channels: all of them at once (one exposed memory
        // latency for the whole image).  Measured on one device: all 16 at once is +5 % end to end at
        // 256 channels, all 8 at once -22 % at 128 channels.
        constexpr int CH = Ci / 8;
        const unsigned short* src = a.in + pass * Ci;
        if (CPT >= 3) {
            constexpr int NL = 2 * 64 * CH / 256;
            u32x4 v[NL];
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int i = tid + u * 256;
                const int bb = i / (64 * CH), p = (i / CH) & 63, c = i % CH;
                v[u] = u32x4{ 0, 0, 0, 0 };
                if (b0 + bb < a.B) v[u] = *reinterpret_cast<const u32x4*>(src + ((size_t)(b0 + bb) * 64 + p) * CiTot + c * 8);
            }
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int i = tid + u * 256;
                const int bb = i / (64 * CH), p = (i / CH) & 63, c = i % CH;
                const int pix = (TAPS == 9) ? ((p >> 3) + 1) * PITCH + (p & 7) + 1 : p;
                *reinterpret_cast<u32x4*>(img + bb * board_bytes + pix * stride + c * 16) = v[u];
            }
        } else {
            for (int i0 = tid; i0 < 2 * 64 * CH; i0 += 4 * 256) {
                u32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 256;
                    const int bb = i / (64 * CH), p = (i / CH) & 63, c = i % CH;
                    v[u] = u32x4{ 0, 0, 0, 0 };
                    if (b0 + bb < a.B) v[u] = *reinterpret_cast<const u32x4*>(src + ((size_t)(b0 + bb) * 64 + p) * CiTot + c * 8);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int i = i0 + u * 256;
                    const int bb = i / (64 * CH), p = (i / CH) & 63, c = i % CH;
                    const int pix = (TAPS == 9) ? ((p >> 3) + 1) * PITCH + (p & 7) + 1 : p;
                    *reinterpret_cast<u32x4*>(img + bb * board_bytes + pix * stride + c * 16) = v[u];
                }
            }
        }
    };

    // ---- this wave: board wave>>1, rows 4*(wave&1)..+3, all 64 channels of block cb
    const int wb = wave >> 1;
    const int lp = PIXMAP[lane & 31];
    const int py = 4 * (wave & 1) + (lp >> 3), px = lp & 7;
    const unsigned b_base = LDS_IMG + wb * board_bytes + ((TAPS == 9) ? (py * PITCH + px) : (py * 8 + px)) * stride + h * 16;
    f32x16 acc[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 s = *reinterpret_cast<const float4*>(a.shift + cb * 64 + ms * 32 + 8 * g + 4 * h);
            acc[ms][4 * g + 0] = s.x; acc[ms][4 * g + 1] = s.y; acc[ms][4 * g + 2] = s.z; acc[ms][4 * g + 3] = s.w;
        }
    // byte offset of the first k-step of chunk n (of a pass) in the image, relative to b_base (a constant after
    // unrolling): chunks come in 64-channel slices, all taps of a slice before the next slice
    auto chunk_off = [](int n) -> unsigned {
        const int q = n / TAPS, tap = n - q * TAPS;
        return ((TAPS == 9) ? (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) : 0u) + q * 128;
    };
    V A[2][8], Bq[2][4];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if (pass > 0) __syncthreads();                     // everybody is done reading the previous pass's image
        stage(pass);
        // image staged + first chunk of the pass landed, for everybody
        if (pass == 0) WIDE_STAMP(1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RDN - 2 + AHEAD)) : "memory");
        if (pass == 0) WIDE_STAMP(2);
        {
            const unsigned a_off = (unsigned)(((pass * NCH) % RDN) * CHUNKB) + lane * 16;
#pragma unroll
            for (int f = 0; f < 8; ++f) A[0][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
#pragma unroll
            for (int k = 0; k < 4; ++k) Bq[0][k] = *reinterpret_cast<const V*>(smem + b_base + chunk_off(0) + k * 32);
        }
        // one chunk step per iteration, fully unrolled: register set n & 1 holds chunk n (see tower_mfma.hip
        // for the ring protocol and the pinned read / MFMA interleave)
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int cur = n & 1, nxt = cur ^ 1, g = pass * NCH + n;
            if (AHEAD) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RDN - 3 + AHEAD)) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (RDN - 3 + AHEAD)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            ring_issue<RDN>(stream, NCHT, g + RDN - 1 + AHEAD, wave, lane);
            const unsigned a_off = (unsigned)(((g + 1) % RDN) * CHUNKB) + lane * 16;
#pragma unroll
            for (int f = 0; f < 8; ++f) A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
            if (n + 1 < NCH) {
#pragma unroll
                for (int k = 0; k < 4; ++k) Bq[nxt][k] = *reinterpret_cast<const V*>(smem + b_base + chunk_off(n + 1) + k * 32);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                acc[0] = Elem<T>::mfma(A[cur][2 * k], Bq[cur][k], acc[0]);
                acc[1] = Elem<T>::mfma(A[cur][2 * k + 1], Bq[cur][k], acc[1]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
        }
    }
    WIDE_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the ring (tail re-fetches) before exit
    WIDE_STAMP(4);

    // ---- epilogue
    const int b = b0 + wb;
    const int p = py * 8 + px;
    if (EPI == 2) {
        if (b >= a.B) return;
        float* lo = reinterpret_cast<float*>(a.out) + (size_t)b * KH_PSIZE + p * KH_POLICY_PLANES;
#pragma unroll
        for (int ms = 0; ms < 2; ++ms)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ch = cb * 64 + ms * 32 + 8 * g + 4 * h + i;
                    if (ch < KH_POLICY_PLANES) lo[ch] = acc[ms][4 * g + i];
                }
        return;
    }
    // The C/D layout leaves 4 consecutive channels of a pixel per lane: written straight out that is 8
    // scattered 8-byte stores per lane.  Transposed through LDS (the image is dead now) every lane
    // gets 8 consecutive channels of a pixel instead: one 16-byte skip load and one 16-byte store per
    // lane, 8 lanes covering a pixel's whole 128-byte line.
    __syncthreads();                                        // everybody is done reading the image
    constexpr int TSTR = 64 * 4 + 16;                       // fp32 row of one pixel + pad
    char* tile = img + wave * 32 * TSTR;                    // this wave's 32 pixels x 64 channels
    const int lr = lane & 31;                               // this lane's pixel column in the tile
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(relu_keep_nan(acc[ms][4 * g]), relu_keep_nan(acc[ms][4 * g + 1]),
                                   relu_keep_nan(acc[ms][4 * g + 2]), relu_keep_nan(acc[ms][4 * g + 3]));
            *reinterpret_cast<float4*>(tile + lr * TSTR + (ms * 32 + 8 * g + 4 * h) * 4) = v;
        }
    // (same wave wrote it: LDS operations of a wave are ordered, no barrier needed)
    if (b >= a.B) return;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int r = (lane >> 3) + 8 * it, c8 = (lane & 7) * 8;       // tile column r = pixel PIXMAP[r] of this wave's rows
        const int lp2 = PIXMAP[r];
        const int pix = (4 * (wave & 1) + (lp2 >> 3)) * 8 + (lp2 & 7);
        const float4 lo = *reinterpret_cast<const float4*>(tile + r * TSTR + c8 * 4);
        const float4 hi = *reinterpret_cast<const float4*>(tile + r * TSTR + c8 * 4 + 16);
        float v[8] = { lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w };
        const size_t o = ((size_t)b * 64 + pix) * Co + cb * 64 + c8;
        if (EPI == 1) {
            const u32x4 sk = *reinterpret_cast<const u32x4*>(a.skip + o);
            const unsigned w4[4] = { sk.x, sk.y, sk.z, sk.w };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[2 * k] += from_bits<T>((unsigned short)(w4[k] & 0xffff));
                v[2 * k + 1] += from_bits<T>((unsigned short)(w4[k] >> 16));
            }
        }
        u32x4 out;
        out.x = to_bits<T>(v[0]) | ((unsigned)to_bits<T>(v[1]) << 16);
        out.y = to_bits<T>(v[2]) | ((unsigned)to_bits<T>(v[3]) << 16);
        out.z = to_bits<T>(v[4]) | ((unsigned)to_bits<T>(v[5]) << 16);
        out.w = to_bits<T>(v[6]) | ((unsigned)to_bits<T>(v[7]) << 16);
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned short*>(a.out) + o) = out;
    }
    WIDE_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// conv4_mfma_kernel — 3x3 layers with 128 (or 2 x 128) input channels and >= 128 output channels, FOUR boards per
// workgroup: each wave owns one whole board (64 pixels) x 128 output channels = 8 accumulators.
//
// Why: conv_mfma_kernel's 64-channel x 32-pixel wave tile needs 1.5 ds_read_b128 per MFMA (all four
// waves re-read the same A fragments) and pulls the layer's weights through the CU once per 128 pixels: at the bf16
// MFMA rate that is 192 B/clk of LDS reads (of 256) and 76 GB/s per CU of L2 -> LDS weight traffic (the L2 gives
// ~70): the step ran 322 clocks for 8 MFMAs.  Here a k-step is 4 A + 2 B fragment reads for 8 MFMAs (0.75 per
// MFMA, 96 B/clk) and the weights pass once per 256 pixels (38 GB/s per CU at peak).  The price is LDS: the four
// boards' 128-channel images (130 560 B) + a 4-slot ring = 163 328 B, one workgroup per CU, so the epilogue goes
// straight from registers to HBM (16-byte stores after a v_permlane32_swap of packed pairs, no transpose tile).
// Same reduction order as conv_mfma_kernel (64-channel slices outermost, then taps, then k-steps), same fp32
// epilogue: the two kernels agree bit for bit and the launcher picks per call.
// Weights: pack_layer_wide128 (kh_api.hip): [Co/128][Ci/64][tap][half][ks2][ms 0..3][lane][8], 8 KB chunks.
template <typename T, int EPI, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv4_mfma_kernel(ConvArgs a)
{
    constexpr int RDN = 4;
    [[maybe_unused]] constexpr int TAPS = 9;              // WIDE_STAMP's condition
    constexpr int LDS_IMG = RDN * CHUNKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int Ci = 128, CiTot = NP * Ci;
    const int Co = a.Co;
    constexpr int stride = Ci * 2 + 16;                    // 272 B per pixel: 17 16-byte slots, odd
    constexpr int board_bytes = NPIX * stride;
    const int b0 = blockIdx.x * 4, cb = blockIdx.y;
    constexpr int NCH = 2 * 9 * 2, NCHT = NP * NCH;        // 8 KB chunks per pass: 2 slices x 9 taps x 2 halves of 2 k-steps
    const char* stream = reinterpret_cast<const char*>(a.w) + (size_t)cb * NCHT * CHUNKB;
    char* img = smem + LDS_IMG;
    WIDE_STAMP(0);
    WIDE_STAMP(7);

#pragma unroll
    for (int i = 0; i < RDN - 1; ++i) ring_issue<RDN>(stream, NCHT, i, wave, lane);

    {   // zero halo of the four images (staging only ever writes interior pixels)
        const u32x4 z = { 0, 0, 0, 0 };
        constexpr int per_px = stride / 16;
        for (int i = tid; i < 4 * NPIX; i += 256) {
            const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
            if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
            char* d = img + (i / NPIX) * board_bytes + pp * stride;
            for (int k = 0; k < per_px; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
        }
    }
    // the four boards' channels [pass * 128, +128): 4096 16-byte pieces, 16 per thread, all in flight at once
    auto stage = [&](int pass) {
        const unsigned short* src = a.in + pass * Ci;
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            const int bb = i >> 10, p = (i >> 4) & 63, c = i & 15;
            v[u] = u32x4{ 0, 0, 0, 0 };
            if (b0 + bb < a.B) v[u] = *reinterpret_cast<const u32x4*>(src + ((size_t)(b0 + bb) * 64 + p) * CiTot + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            const int bb = i >> 10, p = (i >> 4) & 63, c = i & 15;
            const int pix = ((p >> 3) + 1) * PITCH + (p & 7) + 1;
            *reinterpret_cast<u32x4*>(img + bb * board_bytes + pix * stride + c * 16) = v[u];
        }
    };

    // this wave: board `wave`, both 32-pixel halves (rows 0-3 / 4-7), 128 channels of block cb
    const int lp = PIXMAP[lane & 31];
    const int py = lp >> 3, px = lp & 7;
    const unsigned b_base = LDS_IMG + wave * board_bytes + (py * PITCH + px) * stride + h * 16;
    constexpr unsigned HALF = 4 * PITCH * stride;          // second pixel half: four image rows further
    f32x16 acc[8];                                         // [ms * 2 + hp]
#pragma unroll
    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 s = *reinterpret_cast<const float4*>(a.shift + cb * 128 + ms * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                acc[ms * 2 + hp][4 * g + 0] = s.x; acc[ms * 2 + hp][4 * g + 1] = s.y;
                acc[ms * 2 + hp][4 * g + 2] = s.z; acc[ms * 2 + hp][4 * g + 3] = s.w;
            }
        }
    // chunk n of a pass: slice q = n / 18, tap = (n % 18) / 2, half = n & 1 (k-steps 2 * half, 2 * half + 1 of the slice)
    auto chunk_off = [](int n) -> unsigned {
        const int q = n / 18, tap = (n % 18) >> 1, half = n & 1;
        return (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) + q * 128 + half * 64;
    };
    V A[2][8], Bq[2][4];                                   // A[.][ks2 * 4 + ms], Bq[.][ks2 * 2 + hp]
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if (pass > 0) __syncthreads();                     // everybody is done reading the previous pass's image
        stage(pass);
        if (pass == 0) WIDE_STAMP(1);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RDN - 2)) : "memory");
        if (pass == 0) WIDE_STAMP(2);
        {
            const unsigned a_off = (unsigned)(((pass * NCH) % RDN) * CHUNKB) + lane * 16;
#pragma unroll
            for (int f = 0; f < 8; ++f) A[0][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) Bq[0][k * 2 + hp] = *reinterpret_cast<const V*>(smem + b_base + hp * HALF + chunk_off(0) + k * 32);
        }
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int cur = n & 1, nxt = cur ^ 1, g = pass * NCH + n;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (RDN - 3)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            ring_issue<RDN>(stream, NCHT, g + RDN - 1, wave, lane);
            const unsigned a_off = (unsigned)(((g + 1) % RDN) * CHUNKB) + lane * 16;
#pragma unroll
            for (int f = 0; f < 8; ++f) A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
            if (n + 1 < NCH) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) Bq[nxt][k * 2 + hp] = *reinterpret_cast<const V*>(smem + b_base + hp * HALF + chunk_off(n + 1) + k * 32);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) acc[ms * 2 + hp] = Elem<T>::mfma(A[cur][k * 4 + ms], Bq[cur][k * 2 + hp], acc[ms * 2 + hp]);
#pragma unroll
            for (int i = 0; i < 12; ++i) {                  // one operand read per MFMA gap, the last four gaps free
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    }
    WIDE_STAMP(3);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the ring (tail re-fetches) before exit
    WIDE_STAMP(4);

    // ---- epilogue, registers -> HBM.  Lane (pixel, h) holds channels 32 ms + 8 g + 4 h + 0..3: after ReLU (+ skip, in
    // fp32 like nn.cpp:31) and rounding, groups g = 2j / 2j + 1 are exchanged between the lane halves so that lane
    // (pixel, h) owns the 8 consecutive channels 32 ms + 16 j + 8 h + 0..7: one 16-byte store.
    const int b = b0 + wave;
    if (b >= a.B) return;
    unsigned short* outp = reinterpret_cast<unsigned short*>(a.out);
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
        const int pix = (4 * hp + py) * 8 + px;
        const size_t row = ((size_t)b * 64 + pix) * Co + cb * 128;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms) {
            unsigned pk[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[ms * 2 + hp][4 * g + i]);
                if (EPI == 1) {
                    const u32x2 sk = *reinterpret_cast<const u32x2*>(a.skip + row + ms * 32 + 8 * g + 4 * h);
                    v[0] += from_bits<T>((unsigned short)(sk.x & 0xffff)); v[1] += from_bits<T>((unsigned short)(sk.x >> 16));
                    v[2] += from_bits<T>((unsigned short)(sk.y & 0xffff)); v[3] += from_bits<T>((unsigned short)(sk.y >> 16));
                }
                pk[g][0] = pack2<T>(v[0], v[1]);
                pk[g][1] = pack2<T>(v[2], v[3]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // lanes 32-63 of group 2j <-> lanes 0-31 of group 2j + 1
                const auto s0 = __builtin_amdgcn_permlane32_swap(pk[2 * j][0], pk[2 * j + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(pk[2 * j][1], pk[2 * j + 1][1], false, false);
                const u32x4 o = { s0[0], s1[0], s0[1], s1[1] };
                *reinterpret_cast<u32x4*>(outp + row + ms * 32 + 16 * j + 8 * h) = o;
            }
        }
    }
    WIDE_STAMP(5);
}

// ---------------------------------------------------------------------------------------------
// tower128_kernel — the WHOLE 3x3 stack of a 128-filter net (stem + R residual blocks, nn.cpp:62-69) in one launch,
// activations never leaving the CU.  conv4_mfma_kernel's loop already runs at 93 % of the MFMA issue rate, but per
// layer it pays staging (7 000 clocks), an epilogue through HBM (12 000) and a launch for 19 800 clocks of MFMAs,
// and moves 50 MB of activations per layer (0.9 GB per 10x128 forward at batch 1024, 51 MB algorithmic).  In
// conv4's tiling a wave owns a whole board and all 128 channels, so nothing but the weights is shared between waves:
// the layer boundary is wave-local — ReLU (+ the residual stream, kept as packed T in 64 registers: exactly the
// operand the per-layer path re-reads from HBM), round, write the board's next input image in place (the wave's own
// LDS operations are ordered; its MFMAs have all read the old image by then) — and the weight ring simply runs on
// through all (1 + 2R) x 36 chunks.  Same reduction order and epilogue arithmetic as the per-layer kernels: same bits.
#ifdef KAMI_WIDE_DIAG
// diagnostic build only: per layer, wave 0 of the first 256 workgroups stamps loop start / loop end / boundary end
__device__ unsigned long long g_t128_stamps[256 * 64 * 4];
#define T128_STAMP(l, k) do { if (tid == 0 && blockIdx.x < 256 && (l) < 64) g_t128_stamps[(blockIdx.x * 64 + (l)) * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define T128_STAMP(l, k) do {} while (0)
#endif

struct Head4Args {
    const unsigned short* x;      // T [B][64][Ci], Ci = NP * 128
    const unsigned short* w;      // policyconv chunks (Ci / 32 of 8 KB) then policyconv2 chunks (4), pack_layer_wide128 order
    const float* shift1;          // [128] folded pbatchnorm shift
    const float* bias2;           // [128] policyconv2 bias, planes >= 73 zero
    float* policy;                // [B][4672]
    float* logits;                // nullable [B][4672]
    int* flags;                   // [0]: a NaN in some policy row (nn.cpp:176-177); [1]: in the value tensor (nn.cpp:179-180)
    int B;
    // value head (nn.cpp:83-88), nullable as a whole (vw == nullptr: the separate kernels run it)
    const float* vw;              // [Ci] valueconv weight * vbatchnorm scale
    float vshift;
    const float* fcw;             // [256][64] valuefc.weight
    const float* fcb;             // [256]
    const float* fc4;             // valuefc.weight as [k / 4][256][4]
    float* vfull;                 // [B][256]
};

__device__ __forceinline__ float wave_max_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct Tower128Args {
    const float* planes;          // fp32 [B][64][F] (nn.cpp:157), converted while they are staged; nullptr: take `in`
    int F;                        // planes per pixel, <= 128
    unsigned magic;               // ceil(2^32 / F): e / F = umulhi(e, magic) for e < 64 * 128
    const unsigned short* in;     // T [B][64][128]: the planes, converted and zero-padded (planes_to_act_kernel)
    const unsigned short* w;      // (1 + 2R) layers x 36 chunks of 8 KB, pack_layer_wide128 order (Co = Ci = 128)
    const float* shift;           // (1 + 2R) x 128 folded BatchNorm shifts
    unsigned short* out;          // T [B][64][128]: the residual stream after the last block (HEAD: not written)
    int B, R;
    Head4Args hd;                 // HEAD: both heads on the board each wave owns, the weight ring running on into hd.w (8 chunks)
};

// HEAD (round 3): policy_head4_kernel<T, 1>'s work appended — same arithmetic in the same order, so the same bits — on the
// image the last boundary has just written: no second launch, no 16.8 MB of residual stream out and in again per 1 024
// boards, no staging; the ring's prefetch past the tower's last chunk fetches the heads' weights instead of a dummy.
template <typename T, bool HEAD>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tower128_kernel(Tower128Args a)
{
    constexpr int RDN = 4;
    constexpr int LDS_IMG = RDN * CHUNKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int Ci = 128;
    constexpr int stride = Ci * 2 + 16;
    constexpr int board_bytes = NPIX * stride;
    const int b0 = blockIdx.x * 4;
    constexpr int NCH = 36;
    const int NL = 1 + 2 * a.R, NCHT = NL * NCH;
    const char* stream = reinterpret_cast<const char*>(a.w);
    const char* hstream = reinterpret_cast<const char*>(a.hd.w);
    char* img = smem + LDS_IMG;

#pragma unroll
    for (int i = 0; i < RDN - 1; ++i) ring_issue<RDN>(stream, NCHT, i, wave, lane);
    if (a.planes) {
        // fp32 planes straight from the caller's tensor (no planes_to_act launch, no T copy in HBM): a board is
        // 64 F contiguous floats, 16-byte aligned, read as float4 pieces whose four elements may straddle pixels;
        // every element is rounded to T and written to (pixel, plane) of the image.  The whole image is zeroed first
        // (halo, and planes F..127 of the padded stem).
        const u32x4 z = { 0, 0, 0, 0 };
        for (int i = tid; i < 4 * board_bytes / 16; i += 256) *reinterpret_cast<u32x4*>(img + i * 16) = z;
        const int F = a.F, pieces = 16 * F;                 // float4 pieces per board
        constexpr int NU = 32;                               // 4 boards x 16 F pieces / 256 threads <= 32 for F <= 128
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + u * 256;
            const int bb = __umulhi((unsigned)i, a.magic) >> 4;          // i / (16 F)
            const int q = i - bb * pieces;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bb < 4 && b0 + bb < a.B) v[u] = *reinterpret_cast<const float4*>(a.planes + ((size_t)(b0 + bb) * 64) * F + 4 * q);
        }
        __syncthreads();                                     // the zeroes are in place before anybody writes planes over them
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + u * 256;
            const int bb = __umulhi((unsigned)i, a.magic) >> 4;
            if (bb >= 4) continue;
            const int q = i - bb * pieces;
            int p = __umulhi((unsigned)(4 * q), a.magic), c = 4 * q - p * F;
            const float e[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = ((p >> 3) + 1) * PITCH + (p & 7) + 1;
                *reinterpret_cast<unsigned short*>(img + bb * board_bytes + pix * stride + c * 2) = to_bits<T>(e[k]);
                if (++c == F) { c = 0; ++p; }
            }
        }
    } else {
    {
        const u32x4 z = { 0, 0, 0, 0 };
        constexpr int per_px = stride / 16;
        for (int i = tid; i < 4 * NPIX; i += 256) {
            const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
            if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
            char* d = img + (i / NPIX) * board_bytes + pp * stride;
            for (int k = 0; k < per_px; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
        }
    }
    {
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            const int bb = i >> 10, p = (i >> 4) & 63, c = i & 15;
            v[u] = u32x4{ 0, 0, 0, 0 };
            if (b0 + bb < a.B) v[u] = *reinterpret_cast<const u32x4*>(a.in + ((size_t)(b0 + bb) * 64 + p) * Ci + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            const int bb = i >> 10, p = (i >> 4) & 63, c = i & 15;
            const int pix = ((p >> 3) + 1) * PITCH + (p & 7) + 1;
            *reinterpret_cast<u32x4*>(img + bb * board_bytes + pix * stride + c * 16) = v[u];
        }
    }
    }
    const int lp = PIXMAP[lane & 31];
    const int py = lp >> 3, px = lp & 7;
    const unsigned b_base = LDS_IMG + wave * board_bytes + (py * PITCH + px) * stride + h * 16;
    constexpr unsigned HALF = 4 * PITCH * stride;
    // where this lane's outputs go in its board's image: centre of the 3x3 window, its 4-channel groups
    char* const w_base = smem + LDS_IMG + wave * board_bytes + ((py + 1) * PITCH + px + 1) * stride + h * 8;
    f32x16 acc[8];
    // the next layer's 64 shift values of this lane: requested two steps before a layer ends (16 loads, L2), so the
    // boundary finds them in registers instead of waiting out a round trip behind the ring's pieces
    float4 sh[16];
    auto request_shift = [&](int l) {
        const int lc = l < NL ? l : NL - 1;                 // past the last layer: a harmless re-read
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int g = 0; g < 4; ++g) sh[ms * 4 + g] = *reinterpret_cast<const float4*>(a.shift + lc * 128 + ms * 32 + 8 * g + 4 * h);
    };
    auto shift_to_acc = [&]() {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 s4 = sh[ms * 4 + g];
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    acc[ms * 2 + hp][4 * g + 0] = s4.x; acc[ms * 2 + hp][4 * g + 1] = s4.y;
                    acc[ms * 2 + hp][4 * g + 2] = s4.z; acc[ms * 2 + hp][4 * g + 3] = s4.w;
                }
            }
    };
    request_shift(0);
    shift_to_acc();
    auto chunk_off = [](int n) -> unsigned {
        const int q = n / 18, tap = (n % 18) >> 1, half = n & 1;
        return (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) + q * 128 + half * 64;
    };
    V A[2][8], Bq[2][4];
    unsigned xr[8][4][2];                                   // the residual stream of this lane's outputs, packed T (nn.cpp:31's `x`)
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[i][g][0] = xr[i][g][1] = 0;

    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RDN - 2)) : "memory");
    {
        const unsigned a_off = lane * 16;
#pragma unroll
        for (int f = 0; f < 8; ++f) A[0][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
    }
    // ---- layer boundary (this wave's board only), one straight-line copy per kind so that nothing in it branches:
    // KIND 0: stem (nn.cpp:63-65): becomes the residual stream; 1: a block's first conv (nn.cpp:30); 2: its second conv,
    // ReLU before the add of the stream, none after (nn.cpp:31-33): becomes the new stream
    auto boundary = [&](auto kind) {
        constexpr int KIND = decltype(kind)::value;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[ms * 2 + hp][4 * g + i]);
                    if (KIND == 2) {
                        const unsigned s0 = xr[ms * 2 + hp][g][0], s1 = xr[ms * 2 + hp][g][1];
                        v[0] += from_bits<T>((unsigned short)(s0 & 0xffff)); v[1] += from_bits<T>((unsigned short)(s0 >> 16));
                        v[2] += from_bits<T>((unsigned short)(s1 & 0xffff)); v[3] += from_bits<T>((unsigned short)(s1 >> 16));
                    }
                    const unsigned p0 = pack2<T>(v[0], v[1]), p1 = pack2<T>(v[2], v[3]);
                    if (KIND != 1) { xr[ms * 2 + hp][g][0] = p0; xr[ms * 2 + hp][g][1] = p1; }
                    *reinterpret_cast<u32x2*>(w_base + hp * HALF + (ms * 32 + 8 * g) * 2) = u32x2{ p0, p1 };   // (after the last layer: unused)
                }
    };
    for (int l = 0; l < NL; ++l) {
        T128_STAMP(l, 0);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) Bq[0][k * 2 + hp] = *reinterpret_cast<const V*>(smem + b_base + hp * HALF + chunk_off(0) + k * 32);
#pragma unroll
        for (int n = 0; n < NCH; ++n) {
            const int cur = n & 1, nxt = cur ^ 1;
            const int g = l * NCH + n;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (RDN - 3)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (!HEAD || g + RDN - 1 < NCHT) ring_issue<RDN>(stream, NCHT, g + RDN - 1, wave, lane);
            else ring_issue_at(hstream + (size_t)(g + RDN - 1 - NCHT < 8 ? g + RDN - 1 - NCHT : 7) * CHUNKB, (g + RDN - 1) % RDN, wave, lane);   // the heads' chunks follow the tower's
            if (n == NCH - 2) request_shift(l + 1);
            const unsigned a_off = (unsigned)(((n + 1) % RDN) * CHUNKB) + lane * 16;      // 36 chunks per layer: slot = n % 4
#pragma unroll
            for (int f = 0; f < 8; ++f) A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
            if (n + 1 < NCH) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) Bq[nxt][k * 2 + hp] = *reinterpret_cast<const V*>(smem + b_base + hp * HALF + chunk_off(n + 1) + k * 32);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) acc[ms * 2 + hp] = Elem<T>::mfma(A[cur][k * 4 + ms], Bq[cur][k * 2 + hp], acc[ms * 2 + hp]);
            if (n == NCH - 2) __builtin_amdgcn_sched_group_barrier(0x020, 16, 0);       // the shift loads first: they have two steps to land
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        T128_STAMP(l, 1);
        if (l == 0) boundary(std::integral_constant<int, 0>{});
        else if (l & 1) boundary(std::integral_constant<int, 1>{});
        else boundary(std::integral_constant<int, 2>{});
        T128_STAMP(l, 3);
        shift_to_acc();
        T128_STAMP(l, 2);
    }
    if constexpr (HEAD) {
        // ---- both heads (nn.cpp:72-88) on this wave's board, policy_head4_kernel<T, 1>'s arithmetic in its order.  The image
        // holds the residual stream after the last block (the boundary has just written it); chunk NCHT + n of the ring is
        // the heads' chunk n; the tower's register sets (A / Bq parities, accumulators) carry straight on.
        const Head4Args& hd = a.hd;
        const unsigned hb_base = b_base + (PITCH + 1) * stride;         // the centre tap: the board's own pixels
        float vsum[2] = { 0.0f, 0.0f };                      // valueconv partial sums of this lane's two pixels, its half of the channels
        if (hd.vw) {
            const float* vw = hd.vw + 64 * h;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {                 // (the weights once for both pixels; each pixel's sum in channel order as before)
                const float4 w0 = *reinterpret_cast<const float4*>(vw + c8 * 8), w1 = *reinterpret_cast<const float4*>(vw + c8 * 8 + 4);
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    const char* row = smem + LDS_IMG + wave * board_bytes + ((4 * hp + py + 1) * PITCH + px + 1) * stride + h * 128;
                    const u32x4 u = *reinterpret_cast<const u32x4*>(row + c8 * 16);
                    float sacc = vsum[hp];
                    sacc = fmaf(from_bits<T>((unsigned short)(u.x & 0xffff)), w0.x, sacc); sacc = fmaf(from_bits<T>((unsigned short)(u.x >> 16)), w0.y, sacc);
                    sacc = fmaf(from_bits<T>((unsigned short)(u.y & 0xffff)), w0.z, sacc); sacc = fmaf(from_bits<T>((unsigned short)(u.y >> 16)), w0.w, sacc);
                    sacc = fmaf(from_bits<T>((unsigned short)(u.z & 0xffff)), w1.x, sacc); sacc = fmaf(from_bits<T>((unsigned short)(u.z >> 16)), w1.y, sacc);
                    sacc = fmaf(from_bits<T>((unsigned short)(u.w & 0xffff)), w1.z, sacc); sacc = fmaf(from_bits<T>((unsigned short)(u.w >> 16)), w1.w, sacc);
                    vsum[hp] = sacc;
                }
            }
        }
        auto init_acc = [&](const float* shp) {
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 s4 = *reinterpret_cast<const float4*>(shp + ms * 32 + 8 * g4 + 4 * h);
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) {
                        acc[ms * 2 + hp][4 * g4 + 0] = s4.x; acc[ms * 2 + hp][4 * g4 + 1] = s4.y;
                        acc[ms * 2 + hp][4 * g4 + 2] = s4.z; acc[ms * 2 + hp][4 * g4 + 3] = s4.w;
                    }
                }
        };
        auto hchunk_off = [](int n) -> unsigned { return (unsigned)((n >> 1) * 128 + (n & 1) * 64); };
        // four chunk steps of a 1x1 convolution over the board's own pixels; G0: the first chunk's number in the ring
        auto four_steps = [&](int G0) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) Bq[0][k2 * 2 + hp] = *reinterpret_cast<const V*>(smem + hb_base + hp * HALF + hchunk_off(0) + k2 * 32);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int cur = n & 1, nxt = cur ^ 1, g = G0 + n;
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (RDN - 3)) : "memory");
                __builtin_amdgcn_sched_barrier(0);
                ring_issue_at(hstream + (size_t)(g + RDN - 1 - NCHT < 8 ? g + RDN - 1 - NCHT : 7) * CHUNKB, (g + RDN - 1) % RDN, wave, lane);
                const unsigned a_off = (unsigned)(((n + 1) % RDN) * CHUNKB) + lane * 16;       // NCHT % 4 == 0: slot = n
#pragma unroll
                for (int f = 0; f < 8; ++f) A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
                if (n + 1 < 4) {
#pragma unroll
                    for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) Bq[nxt][k2 * 2 + hp] = *reinterpret_cast<const V*>(smem + hb_base + hp * HALF + hchunk_off(n + 1) + k2 * 32);
                }
#pragma unroll
                for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                    for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                        for (int hp = 0; hp < 2; ++hp) acc[ms * 2 + hp] = Elem<T>::mfma(A[cur][k2 * 4 + ms], Bq[cur][k2 * 2 + hp], acc[ms * 2 + hp]);
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            }
        };
        T128_STAMP(NL, 0);
        init_acc(hd.shift1);
        four_steps(NCHT);
        T128_STAMP(NL, 1);
        // policyconv's output (pbatchnorm folded, ReLU, rounded to T) -> this wave's image, in place
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[ms * 2 + hp][4 * g4 + i]);
                    *reinterpret_cast<u32x2*>(w_base + hp * HALF + (ms * 32 + 8 * g4) * 2) = u32x2{ pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]) };
                }
        init_acc(hd.bias2);
        four_steps(NCHT + 4);
        T128_STAMP(NL, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // drain the ring (tail re-fetches) before its memory is reused
        // ---- softmax over the board's 64 pixels x 73 planes, all inside this wave
        const int b = b0 + wave;
        float m = -INFINITY;
        bool nan = false;
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (i8 >> 1) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (ch < KH_POLICY_PLANES) { const float v = acc[i8][r]; nan |= (v != v); m = fmaxf(m, v); }
            }
        m = wave_max_f(m);
        float sum = 0.0f;
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (i8 >> 1) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (ch < KH_POLICY_PLANES) sum += HEAD_EXP(acc[i8][r] - m);
            }
        sum = wave_sum_f(sum);
        const float ls = HEAD_LOG(sum);
        const bool row_nan = __any(nan);
        T128_STAMP(NL, 3);
        __syncthreads();                                   // ring and images are dead: the rows go through their memory
        float* rowbuf = reinterpret_cast<float*>(smem) + wave * KH_PSIZE;
        const bool live = b < a.B;
        const int lpix = py * 8 + px;                      // this lane's pixel within a 32-pixel half
        auto put_row = [&](float* dst_row, bool probs) {
#pragma unroll
            for (int hp = 0; hp < 2; ++hp)
#pragma unroll
                for (int ms = 0; ms < 3; ++ms)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ch = ms * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                        if (ch < KH_POLICY_PLANES) {
                            const float v = acc[ms * 2 + hp][r];
                            rowbuf[(32 * hp + lpix) * KH_POLICY_PLANES + ch] = probs ? (row_nan ? NAN : HEAD_EXP((v - m) - ls)) : v;
                        }
                    }
            if (live) {
                for (int i = lane; i < KH_PSIZE / 4; i += 64)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<f32x4s*>(dst_row) + i), "v"(*reinterpret_cast<const f32x4s*>(rowbuf + 4 * i)) : "memory");
            }
        };
        if (hd.logits) put_row(hd.logits + (size_t)(live ? b : 0) * KH_PSIZE, false);
        put_row(hd.policy + (size_t)(live ? b : 0) * KH_PSIZE, true);
        T128_STAMP(NL + 1, 0);
        if (!live) return;
        if (row_nan && lane == 0) atomicOr(&hd.flags[0], 1);
        if (hd.vw) {
            float* v64 = reinterpret_cast<float*>(smem + 80 * 1024) + wave * 64;     // behind the four rows (74.75 KB), in the dead images
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) {
                const unsigned u = __float_as_uint(vsum[hp]);
                const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
                if (h == 0) v64[32 * hp + lpix] = relu_keep_nan(tot + hd.vshift);
            }
            // lane l: outputs l, l + 64, l + 128, l + 192 (consecutive lanes, consecutive 16 bytes of fc4); the sums run over k in
            // the order they always did, so the same bits
            float o[4] = { hd.fcb[lane], hd.fcb[lane + 64], hd.fcb[lane + 128], hd.fcb[lane + 192] };
#pragma unroll 4
            for (int k4 = 0; k4 < 16; ++k4) {
                const float4 hv = *reinterpret_cast<const float4*>(v64 + k4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 wv = *reinterpret_cast<const float4*>(hd.fc4 + ((size_t)k4 * KH_VALUE_WIDTH + lane + 64 * j) * 4);
                    o[j] = fmaf(hv.x, wv.x, o[j]); o[j] = fmaf(hv.y, wv.y, o[j]); o[j] = fmaf(hv.z, wv.z, o[j]); o[j] = fmaf(hv.w, wv.w, o[j]);
                }
            }
            bool vnan = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float r = tanhf(o[j]);
                vnan |= (r != r);
                hd.vfull[(size_t)b * KH_VALUE_WIDTH + lane + 64 * j] = r;
            }
            if (vnan) atomicOr(&hd.flags[1], 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        T128_STAMP(NL + 1, 1);
        return;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the ring (tail re-fetches) before exit

    // ---- the residual stream after the last block -> HBM (16-byte stores, as conv4_mfma_kernel's epilogue)
    const int b = b0 + wave;
    if (b >= a.B) return;
#pragma unroll
    for (int hp = 0; hp < 2; ++hp) {
        const int pix = (4 * hp + py) * 8 + px;
        const size_t row = ((size_t)b * 64 + pix) * 128;
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(xr[ms * 2 + hp][2 * j][0], xr[ms * 2 + hp][2 * j + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(xr[ms * 2 + hp][2 * j][1], xr[ms * 2 + hp][2 * j + 1][1], false, false);
                const u32x4 o = { s0[0], s1[0], s0[1], s1[1] };
                *reinterpret_cast<u32x4*>(a.out + row + ms * 32 + 16 * j + 8 * h) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// tower256_kernel — the whole 3x3 stack of a 256-filter net (BASELINE configs[4]'s) in one launch, activations on chip.
// Four 256-channel board images do not fit LDS, so a workgroup is TWO boards (2 x 63 360 B).  The first version kept
// tower128's roles — a wave = 32 pixels x all 256 channels, weights through an LDS ring — and ran its 8-MFMA steps in
// 413 clocks instead of 256: every wave re-read the same eight A fragments (36 KB of LDS reads per step) next to the
// ring's 8 KB of LDS-DMA writes, more LDS time than MFMA time, and a workgroup barrier per step.  So the roles are
// turned: a wave owns 64 OUTPUT CHANNELS x all 128 pixels of the two boards (2 row tiles x 4 pixel tiles = 8
// accumulators).  Its two A fragments per k-step are nobody else's, so they go global -> registers directly (16 bytes
// per lane, requested four k-steps ahead; every weight byte still enters the CU once), and LDS only serves the four
// B fragments (16 KB per step for the workgroup, a quarter of its bandwidth): no ring, no LDS-DMA, no barrier inside a
// layer.  A layer boundary is two workgroup barriers (everybody has read the old image; everybody has written its 64
// channels of the new one) around tower128's epilogue.  The stem has 128 (padded) input planes = 72 k-steps, the other
// layers 144.  Weights: pack_layer_wide128(..., CBC = 256): [Ci/64][tap][k-step][row tile 0..7][lane][8] — the
// per-layer kernels' reduction order, so the same bits.
struct Tower256Args {
    const float* planes;          // fp32 [B][64][F], F <= 128, 16-byte aligned
    int F;
    unsigned magic;               // ceil(2^32 / F)
    const unsigned short* w;      // 72 + 2R x 144 k-steps of 8 KB
    const float* shift;           // (1 + 2R) x 256 folded BatchNorm shifts
    unsigned short* out;          // T [B][64][256]: the residual stream after the last block
    int B, R;
};

// CH = 256: as described.  CH = 128 (BASELINE configs[2]'s width): a wave owns 32 channels x both boards (4 accumulators,
// one A fragment per k-step); two 128-channel images are 65 KB, so TWO workgroups share a CU and one's boundary runs
// under the other's MFMAs — and 2 boards per workgroup fill the chip from batch 512 on, where tower128_kernel's four
// boards per workgroup leave half the CUs idle.
template <typename T, int CH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, CH == 128 ? 2 : 1))) void tower2b_kernel(Tower256Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int MSW = CH / 128;                           // row tiles (32 output channels) per wave
    constexpr int KSB = CH / 32 * 1024;                     // bytes of one k-step's fragments
    constexpr int stride = CH * 2 + 16;                     // 528 B per pixel: 33 16-byte slots, odd
    constexpr int board_bytes = NPIX * stride;
    const int b0 = blockIdx.x * 2;
    const int NL = 1 + 2 * a.R;
    constexpr int NSL = CH / 64;                            // 64-channel slices of a tower layer (the stem: 2)
    const int NKT = 72 + 2 * a.R * 36 * NSL;                // k-steps of the whole stack
    char* img = smem;
    // this wave's A fragments of k-step k: row tiles MSW * wave .. of the step's CH / 32
    // (a scalar base per k-step + one 32-bit lane offset: the SGPR-base form of global_load, no 64-bit vector address math)
    const char* wl = reinterpret_cast<const char*>(a.w) + (size_t)(MSW * wave) * 1024;
    const unsigned wlane = lane * 16;
    // requested NA k-steps ahead: 4 x 256 clocks of MFMAs at 256 channels, 8 x 128 at 128 — an L2 round trip under load
    constexpr int NA = KAMI_T2B_NA;                          // k-steps of weight fragments in flight (8 bought nothing at 128 channels and cost the second workgroup per CU its registers)
    V Areg[NA][MSW];
#if KAMI_WIDE_BUFA
    // buffer loads: descriptor in SGPRs, the k-step's byte offset in ONE SGPR, the lane offset a loop-invariant VGPR — one
    // scalar add and the load per k-step (the global_load form cost seven scalar instructions and a v_mov per step, and a
    // 4-MFMA step is issue-bound with one wave per SIMD: 20x256 batch 256 with the weight loads removed 622 -> 487 us).
    // Past the stack's end the range check returns zeros: no clamp.
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wl), 0, NKT * KSB - MSW * wave * 1024, 0x00020000);
    auto load_a = [&](int slot, int k) {
#pragma unroll
        for (int ms = 0; ms < MSW; ++ms)
            Areg[slot][ms] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, wlane + ms * 1024, k * KSB, 0));
    };
#else
    auto load_a = [&](int slot, int k) {
        const size_t kc = (size_t)(k < NKT ? k : NKT - 1) * KSB;         // past the end: a harmless re-read
        unsigned vo = wlane;
        asm volatile("" : "+v"(vo));                         // keeps the lane offset out of a hoisted 64-bit vector base
#pragma unroll
        for (int ms = 0; ms < MSW; ++ms) Areg[slot][ms] = *reinterpret_cast<const V*>(wl + kc + ms * 1024 + vo);
    };
#endif
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j, j);
    {
        // fp32 planes -> T in the image (tower128_kernel's ingest, two boards): everything zeroed first (halo, planes
        // F..255 of the image: the stem only reads 0..127), then float4 pieces scattered to (pixel, plane)
        const u32x4 z = { 0, 0, 0, 0 };
        for (int i = tid; i < 2 * board_bytes / 16; i += 256) *reinterpret_cast<u32x4*>(img + i * 16) = z;
        const int F = a.F, pieces = 16 * F;
        constexpr int NU = 16;                               // 2 boards x 16 F pieces / 256 threads <= 16 for F <= 128
        float4 v[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + u * 256;
            const int bb = __umulhi((unsigned)i, a.magic) >> 4;
            const int q = i - bb * pieces;
            v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bb < 2 && b0 + bb < a.B) v[u] = *reinterpret_cast<const float4*>(a.planes + ((size_t)(b0 + bb) * 64) * F + 4 * q);
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int i = tid + u * 256;
            const int bb = __umulhi((unsigned)i, a.magic) >> 4;
            if (bb >= 2) continue;
            const int q = i - bb * pieces;
            int p = __umulhi((unsigned)(4 * q), a.magic), c = 4 * q - p * F;
            const float e[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = ((p >> 3) + 1) * PITCH + (p & 7) + 1;
                *reinterpret_cast<unsigned short*>(img + bb * board_bytes + pix * stride + c * 2) = to_bits<T>(e[k]);
                if (++c == F) { c = 0; ++p; }
            }
        }
    }
    // pixel tile pt = 0..3: board pt >> 1, rows 4 (pt & 1) .. + 3; this lane's pixel of a tile: (py, px)
    const int lp = PIXMAP[lane & 31];
    const int py = lp >> 3, px = lp & 7;
    const unsigned b_base = (py * PITCH + px) * stride + h * 16;
    auto tile_off = [](int pt) -> unsigned { return (unsigned)((pt >> 1) * board_bytes + (pt & 1) * 4 * PITCH * stride); };
    char* const w_base = smem + ((py + 1) * PITCH + px + 1) * stride + (32 * MSW * wave + 4 * h) * 2;
    f32x16 acc[MSW * 4];                                    // [ms * 4 + pt]
    auto load_shift = [&](int l) {
        const int lc = l < NL ? l : NL - 1;
#pragma unroll
        for (int ms = 0; ms < MSW; ++ms)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 s4 = *reinterpret_cast<const float4*>(a.shift + lc * CH + 32 * MSW * wave + ms * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) {
                    acc[ms * 4 + pt][4 * g + 0] = s4.x; acc[ms * 4 + pt][4 * g + 1] = s4.y;
                    acc[ms * 4 + pt][4 * g + 2] = s4.z; acc[ms * 4 + pt][4 * g + 3] = s4.w;
                }
            }
    };
    load_shift(0);
    // k-step n of a slice: tap = n / 4, k-step kk = n % 4 of the slice's 64 channels
    auto kstep_off = [](int n) -> unsigned {
        const int tap = n >> 2, kk = n & 3;
        return (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) + kk * 32;
    };
    // B fragments one k-step ahead at 256 channels (8 MFMAs = 256 clocks of cover), TWO ahead at 128 (a step is 4 MFMAs:
    // one step of lead is an LDS round trip, and every MFMA waited for its operand)
    constexpr int NBS = MSW == 1 ? 3 : 2;
    V Bq[NBS][4];
    unsigned xr[MSW * 4][4][2];                                 // the residual stream of this lane's outputs, packed T
#pragma unroll
    for (int i = 0; i < MSW * 4; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[i][g][0] = xr[i][g][1] = 0;
    __syncthreads();                                        // the images are staged
    // 36 k-steps: one 64-channel slice x 9 taps x 4; register-set parity and the A slot are compile-time: 36 % 4 == 0,
    // and with 8 slots the slice's first slot is 0 or 4 (`s0`, an integral_constant)
    auto slice_steps = [&](int k0, int q, auto s0) {
        constexpr int S0 = decltype(s0)::value;
#pragma unroll
        for (int n = 0; n < 36; ++n) {
            const int cur = n % NBS, nxt = (n + NBS - 1) % NBS, slot = (n + S0) % NA;
            // activations of k-step n + NBS - 1 (past the slice's end: the next slice's first — or, at a layer's end, a
            // harmless address: the layer start re-reads after the boundary's barriers)
            const int m = n + NBS - 1;
            const unsigned off = m < 36 ? kstep_off(m) + q * 128 : kstep_off(m - 36) + (q + 1) * 128;
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) Bq[nxt][pt] = *reinterpret_cast<const V*>(smem + b_base + tile_off(pt) + off);
#pragma unroll
            for (int ms = 0; ms < MSW; ++ms)
#pragma unroll
                for (int pt = 0; pt < 4; ++pt) acc[ms * 4 + pt] = Elem<T>::mfma(Areg[slot][ms], Bq[cur][pt], acc[ms * 4 + pt]);
            load_a(slot, k0 + n + NA);                       // this slot's fragments have been consumed: request k-step + NA
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // the four B reads over the first MFMA gaps
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            if (MSW == 2) __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, MSW, 0);        // then the weight loads
        }
    };
    auto boundary = [&](auto kind) {
        constexpr int KIND = decltype(kind)::value;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int ms = 0; ms < MSW; ++ms)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[ms * 4 + pt][4 * g + i]);
                    if (KIND == 2) {
                        const unsigned s0 = xr[ms * 4 + pt][g][0], s1 = xr[ms * 4 + pt][g][1];
                        v[0] += from_bits<T>((unsigned short)(s0 & 0xffff)); v[1] += from_bits<T>((unsigned short)(s0 >> 16));
                        v[2] += from_bits<T>((unsigned short)(s1 & 0xffff)); v[3] += from_bits<T>((unsigned short)(s1 >> 16));
                    }
                    const unsigned p0 = pack2<T>(v[0], v[1]), p1 = pack2<T>(v[2], v[3]);
                    if (KIND != 1) { xr[ms * 4 + pt][g][0] = p0; xr[ms * 4 + pt][g][1] = p1; }
                    *reinterpret_cast<u32x2*>(w_base + tile_off(pt) + (ms * 32 + 8 * g) * 2) = u32x2{ p0, p1 };
                }
    };
    int k0 = 0;
    for (int l = 0; l < NL; ++l) {
        T128_STAMP(l, 0);
#pragma unroll
        for (int j = 0; j < NBS - 1; ++j)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) Bq[j][pt] = *reinterpret_cast<const V*>(smem + b_base + tile_off(pt) + kstep_off(j));
        const int nslice = l == 0 ? 2 : NSL;                // the stem's 128 (padded) planes, the tower's CH channels
        for (int q = 0; q < nslice; ++q) {
            if constexpr (36 % NA == 0) slice_steps(k0, q, std::integral_constant<int, 0>{});
            else if ((k0 & 7) == 0) slice_steps(k0, q, std::integral_constant<int, 0>{});
            else slice_steps(k0, q, std::integral_constant<int, 4>{});
            k0 += 36;
        }
        T128_STAMP(l, 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has read what it needs of the old image
        if (l == 0) boundary(std::integral_constant<int, 0>{});
        else if (l & 1) boundary(std::integral_constant<int, 1>{});
        else boundary(std::integral_constant<int, 2>{});
        T128_STAMP(l, 3);
        load_shift(l + 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the new image is complete
        T128_STAMP(l, 2);
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int b = b0 + (pt >> 1);
        if (b >= a.B) continue;
        const int pix = (4 * (pt & 1) + py) * 8 + px;
        const size_t row = ((size_t)b * 64 + pix) * CH + 32 * MSW * wave;
#pragma unroll
        for (int ms = 0; ms < MSW; ++ms)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const auto s0 = __builtin_amdgcn_permlane32_swap(xr[ms * 4 + pt][2 * j][0], xr[ms * 4 + pt][2 * j + 1][0], false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(xr[ms * 4 + pt][2 * j][1], xr[ms * 4 + pt][2 * j + 1][1], false, false);
                const u32x4 o = { s0[0], s1[0], s0[1], s1[1] };
                *reinterpret_cast<u32x4*>(a.out + row + ms * 32 + 16 * j + 8 * h) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------
// tower2s_kernel — tower2b_kernel<T, 256> for batches that leave half the chip idle (BASELINE configs[4] per GPU: 20 x 256
// at batch 256 = 128 board pairs on 256 CUs).  TWO workgroups per board pair, each computing HALF the output channels
// (128 of 256) of every layer for both boards; after each layer they exchange their halves of the new image through
// global memory: write-through (sc1) 16-byte stores, drained, then one relaxed agent-scope flag per workgroup and layer;
// the partner polls that flag and reads the bytes with sc1 loads (never through its L1; placement-independent — partners
// are blocks g and g ^ 8, which share an XCD under the usual round-robin placement, but nothing relies on it).
// The exchange hides behind half a layer: a workgroup walks the reduction OWN channels first (64-channel slices 2h, 2h+1
// — their activations are already in its LDS), the partner's (2(1-h), 2(1-h)+1) second, and TWO more waves do all of the
// moving (LDS -> global, flag, poll, global -> LDS) while the four compute waves run the first half.  The price of that
// order: workgroups with h = 1 sum the slices as 2,3,0,1, so their channels differ in fp32 rounding from
// tower2b_kernel / the per-layer kernels (which walk 0,1,2,3) — this variant agrees with them to tolerance, not bit for
// bit (tests/test_gpu_parity.py::test_wide_split_channels_vs_two_board_tower).
// A wave owns 32 output channels x all 128 pixels (4 accumulators), one weight fragment per k-step global -> registers
// (tower2b<T,128>'s step); 6 waves on 4 SIMDs, so at most 256 registers (it needs ~190).  Bounded spin: a partner that
// does not show up within a second raises the policy NaN flag (the call fails) instead of hanging the device; with at most
// one workgroup per CU and no more workgroups than CUs every partner is resident or becomes so.
#ifdef KAMI_WIDE_DIAG
__device__ unsigned long long g_t2s_stamps[256 * 64 * 8];
#define T2S_STAMP(l, k) do { if (lane == 0 && blockIdx.x < 256 && (l) < 64) g_t2s_stamps[(blockIdx.x * 64 + (l)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define T2S_STAMP(l, k) do {} while (0)
#endif
struct Tower2sArgs {
    const float* planes;          // fp32 [B][64][F], F <= 128, 16-byte aligned
    int F;
    unsigned magic;               // ceil(2^32 / F)
    const unsigned short* w;      // tower2b's packing: 72 + 2R x 144 k-steps of 8 KB ([k-step][row tile 0..7][lane][8])
    const float* shift;           // (1 + 2R) x 256
    unsigned short* out;          // T [B][64][256]
    int B, R;
    unsigned short* xbuf;         // [2 parities][npairs][2 halves][128 pixels][128 channels] T
    unsigned* xflag;              // [npairs][2] flags, 64 bytes apart; zeroed by the launcher before every launch
    int* flags;                   // the engine's NaN flags ([0] is raised on an exchange time-out, [3] = 'X')
    int npairs;
    int abl;                      // timing-only ablations (KAMI_T2S_ABL): 1 no global exchange, 2 no mover copies at all
};

template <typename T>
__global__ __launch_bounds__(384) __attribute__((amdgpu_waves_per_eu(1, 2))) void tower2s_kernel(Tower2sArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int CH = 256;
    constexpr int KSB = CH / 32 * 1024;                     // bytes of one k-step's fragments
    constexpr int stride = CH * 2 + 16;
    constexpr int board_bytes = NPIX * stride;
    const int g = blockIdx.x;
    const int half = (g >> 3) & 1;
    const int pair = ((g >> 4) << 3) | (g & 7);             // blocks g and g ^ 8 are the two halves of a pair
    if (pair >= a.npairs) return;
    const int b0 = pair * 2;
    const int NL = 1 + 2 * a.R;
    const int NKT = 72 + 2 * a.R * 144;
    char* img = smem;
    {
        // the planes of both boards, as tower2b_kernel stages them (both workgroups of a pair read them: the second read
        // is an L2 hit when they share an XCD); the fifth wave only helps with the zeroes
        const u32x4 z = { 0, 0, 0, 0 };
        for (int i = tid; i < 2 * board_bytes / 16; i += 384) *reinterpret_cast<u32x4*>(img + i * 16) = z;
        const int F = a.F, pieces = 16 * F;
        constexpr int NU = 16;
        float4 v[NU];
        if (wave < 4) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = tid + u * 256;
                const int bb = __umulhi((unsigned)i, a.magic) >> 4;
                const int q = i - bb * pieces;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (bb < 2 && b0 + bb < a.B) v[u] = *reinterpret_cast<const float4*>(a.planes + ((size_t)(b0 + bb) * 64) * F + 4 * q);
            }
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = tid + u * 256;
                const int bb = __umulhi((unsigned)i, a.magic) >> 4;
                if (bb >= 2) continue;
                const int q = i - bb * pieces;
                int p = __umulhi((unsigned)(4 * q), a.magic), c = 4 * q - p * F;
                const float e[4] = { v[u].x, v[u].y, v[u].z, v[u].w };
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int pix = ((p >> 3) + 1) * PITCH + (p & 7) + 1;
                    *reinterpret_cast<unsigned short*>(img + bb * board_bytes + pix * stride + c * 2) = to_bits<T>(e[k]);
                    if (++c == F) { c = 0; ++p; }
                }
            }
        }
    }
    __syncthreads();                                        // the images are staged
    // layer l's barriers, for all five waves: [X] (l > 0: the partner's half of layer l - 1's output is in the image),
    // [R] (every wave has read what it needs of the old image), [W] (the own half of the new image is written)
    if (wave >= 4) {
        // ---- the two movers, one board each: 16 lanes per pixel (16 x 16 B = the 256 B of one half of a pixel's channels),
        // 4 pixels per instruction, a board's 16 instructions in flight at once.  Each has a flag of its own and talks to
        // the partner's mover of the same board (one mover did it in 6-7 us, longer than the half layer it hides behind)
        const int hw = wave - 4;
        unsigned* const myflag = a.xflag + (size_t)(pair * 2 + half) * 16 + hw * 8;
        unsigned* const itsflag = a.xflag + (size_t)(pair * 2 + (half ^ 1)) * 16 + hw * 8;
        const size_t half_elems = (size_t)128 * 128;
        const unsigned lane_px = lane >> 4, lane_b = (lane & 15) * 16;
        auto lds_off = [&](int p, int hh) -> unsigned {     // pixel p = 0..127 of the pair, channels of half hh
            const int bb = p >> 6, pp = p & 63;
            return (unsigned)(bb * board_bytes + (((pp >> 3) + 1) * PITCH + (pp & 7) + 1) * stride + hh * 256) + lane_b;
        };
        for (int l = 0; l < NL; ++l) {
            if (l > 0 && !(a.abl & 2)) {
                // the partner's channels of layer l - 1's output: epoch l in its flag
                bool ok = true;
                if (!(a.abl & 1))
                if (lane == 0) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    while (__hip_atomic_load(itsflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)l) {
                        __builtin_amdgcn_s_sleep(8);
                        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) { ok = false; break; }     // 1 s at 100 MHz
                    }
                    if (!ok) { atomicOr(&a.flags[0], 1); a.flags[3] = 'X'; }
                }
                if (hw == 0) T2S_STAMP(l, 6);
                const unsigned short* src = a.xbuf + ((size_t)(((l - 1) & 1) * a.npairs + pair) * 2 + (half ^ 1)) * half_elems;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned short*>(src), 0, (int)(half_elems * 2), 0x00020000);
                {
                    // all 32 loads in flight at once (128 registers of this wave's 256): four batches of eight were four
                    // fabric round trips one after the other — longer than the half layer they have to hide behind
                    u32x4 r[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int p = hw * 64 + k * 4 + lane_px;
                        if (!(a.abl & 1)) r[k] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, p * 256 + lane_b, 0, 16);        // aux 16 = sc1
                        else r[k] = u32x4{ 0, 0, 0, 0 };
                        if (a.abl & 4) __builtin_amdgcn_s_sleep(1);      // paced: the compute waves' weight loads go through the same L1
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int p = hw * 64 + k * 4 + lane_px;
                        *reinterpret_cast<u32x4*>(smem + lds_off(p, half ^ 1)) = r[k];
                    }
                }
            }
            if (l > 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (hw == 0) T2S_STAMP(l, 7);
                asm volatile("s_barrier" ::: "memory");                                                      // [X]
            }
            asm volatile("s_barrier" ::: "memory");                                                          // [R]
            asm volatile("s_barrier" ::: "memory");                                                          // [W]
            if (l + 1 < NL && !(a.abl & 2)) {
                unsigned short* dst = a.xbuf + ((size_t)((l & 1) * a.npairs + pair) * 2 + half) * half_elems;
                const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(dst, 0, (int)(half_elems * 2), 0x00020000);
                {
                    u32x4 r[16];
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int p = hw * 64 + k * 4 + lane_px;
                        r[k] = *reinterpret_cast<const u32x4*>(smem + lds_off(p, half));
                    }
#pragma unroll
                    for (int k = 0; k < 16; ++k) {
                        const int p = hw * 64 + k * 4 + lane_px;
                        if (!(a.abl & 1)) __builtin_amdgcn_raw_buffer_store_b128(r[k], rsrc, p * 256 + lane_b, 0, 16);         // write-through
                        if (a.abl & 4) __builtin_amdgcn_s_sleep(1);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                             // drained: then the flag
                if (lane == 0) __hip_atomic_store(myflag, (unsigned)(l + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (hw == 0) T2S_STAMP(l + 1, 5);
            }
        }
        return;
    }
    // ---- the four compute waves
    const int rt = 4 * half + wave;                         // this wave's row tile (32 output channels) of the 8 of a k-step
    const char* wl = reinterpret_cast<const char*>(a.w) + (size_t)rt * 1024;
    const unsigned wlane = lane * 16;
    constexpr int NA = KAMI_T2S_NA;                          // k-steps of weight fragments in flight
    static_assert(36 % NA == 0, "a slice's first k-step must land on slot 0");
    V Areg[NA];
#if KAMI_WIDE_BUFA
    const auto a_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wl), 0, NKT * KSB - rt * 1024, 0x00020000);   // (see tower2b_kernel)
    auto load_a = [&](int slot, int k) {
        Areg[slot] = __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(a_rsrc, wlane, k * KSB, 0));
    };
#else
    auto load_a = [&](int slot, int k) {
        const size_t kc = (size_t)(k < NKT ? k : NKT - 1) * KSB;         // past the end: a harmless re-read
        unsigned vo = wlane;
        asm volatile("" : "+v"(vo));
        Areg[slot] = *reinterpret_cast<const V*>(wl + kc + vo);
    };
#endif
#pragma unroll
    for (int j = 0; j < NA; ++j) load_a(j, j);
    const int lp = PIXMAP[lane & 31];
    const int py = lp >> 3, px = lp & 7;
    const unsigned b_base = (py * PITCH + px) * stride + h * 16;
    auto tile_off = [](int pt) -> unsigned { return (unsigned)((pt >> 1) * board_bytes + (pt & 1) * 4 * PITCH * stride); };
    char* const w_base = smem + ((py + 1) * PITCH + px + 1) * stride + (32 * rt + 4 * h) * 2;
    f32x16 acc[4];
    auto load_shift = [&](int l) {
        const int lc = l < NL ? l : NL - 1;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const float4 s4 = *reinterpret_cast<const float4*>(a.shift + lc * CH + 32 * rt + 8 * gq + 4 * h);
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) {
                acc[pt][4 * gq + 0] = s4.x; acc[pt][4 * gq + 1] = s4.y;
                acc[pt][4 * gq + 2] = s4.z; acc[pt][4 * gq + 3] = s4.w;
            }
        }
    };
    load_shift(0);
    auto kstep_off = [](int n) -> unsigned {
        const int tap = n >> 2, kk = n & 3;
        return (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) + kk * 32;
    };
    constexpr int NBS = 3;                                   // activation fragments two k-steps ahead
    V Bq[NBS][4];
    unsigned xr[4][4][2];                                    // the residual stream of this lane's outputs, packed T
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) xr[i][gq][0] = xr[i][gq][1] = 0;
    // 36 k-steps of activation slice `q` (weights: k-steps kcur ..; the fragments requested past the slice's end are the
    // NEXT slice of this workgroup's walk: knext / qnext)
    auto slice_steps = [&](int kcur, int knext, int q, int qnext) {
#pragma unroll
        for (int n = 0; n < 36; ++n) {
            const int cur = n % NBS, nxt = (n + NBS - 1) % NBS, slot = n % NA;                  // 36 % NA == 0: a slice starts on slot 0
            const int m = n + NBS - 1;
            const unsigned off = m < 36 ? kstep_off(m) + q * 128 : kstep_off(m - 36) + qnext * 128;
            __builtin_amdgcn_sched_barrier(0);               // a step's loads stay in their step
#if !defined(KAMI_T2S_NOB)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) Bq[nxt][pt] = *reinterpret_cast<const V*>(smem + b_base + tile_off(pt) + off);
#endif
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) acc[pt] = Elem<T>::mfma(Areg[slot], Bq[cur][pt], acc[pt]);
#if !defined(KAMI_T2S_NOA)
            load_a(slot, n + NA < 36 ? kcur + n + NA : knext + n + NA - 36);
#endif
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };
    auto preload_b = [&](int q) {
#pragma unroll
        for (int j = 0; j < NBS - 1; ++j)
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) Bq[j][pt] = *reinterpret_cast<const V*>(smem + b_base + tile_off(pt) + kstep_off(j) + q * 128);
    };
    auto boundary = [&](auto kind) {
        constexpr int KIND = decltype(kind)::value;
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[pt][4 * gq + i]);
                if (KIND == 2) {
                    const unsigned s0 = xr[pt][gq][0], s1 = xr[pt][gq][1];
                    v[0] += from_bits<T>((unsigned short)(s0 & 0xffff)); v[1] += from_bits<T>((unsigned short)(s0 >> 16));
                    v[2] += from_bits<T>((unsigned short)(s1 & 0xffff)); v[3] += from_bits<T>((unsigned short)(s1 >> 16));
                }
                const unsigned p0 = pack2<T>(v[0], v[1]), p1 = pack2<T>(v[2], v[3]);
                if (KIND != 1) { xr[pt][gq][0] = p0; xr[pt][gq][1] = p1; }
                *reinterpret_cast<u32x2*>(w_base + tile_off(pt) + (8 * gq) * 2) = u32x2{ p0, p1 };
            }
    };
    // slice i of this workgroup's walk through a tower layer -> the 64-channel slice it is: own channels first
    // (ONE call site of slice_steps inside a loop, as in tower2b_kernel: with the four slices written out one after the
    //  other hipcc sank every prefetched fragment down to its use — load, wait, MFMA, 3.8x the time)
    const int sx = 2 * half;
    auto slice_of = [&](int l, int i) -> int { return l == 0 ? i : (i ^ sx); };
    auto kbase_of = [&](int l) -> int { return l == 0 ? 0 : 72 + (l - 1) * 144; };
    for (int l = 0; l < NL; ++l) {
        const int ns = l == 0 ? 2 : 4;
        if (wave == 0) T2S_STAMP(l, 0);
        preload_b(slice_of(l, 0));
        for (int i = 0; i < ns; ++i) {
            if (l > 0 && i == 2) {
                if (wave == 0) T2S_STAMP(l, 1);
                asm volatile("s_barrier" ::: "memory");                                   // [X]
                if (wave == 0) T2S_STAMP(l, 2);
                preload_b(slice_of(l, 2));                   // (what the previous slice prefetched here was not the partner's yet)
            }
            const int q = slice_of(l, i);
            const bool last = i + 1 == ns;
            const int qn = last ? slice_of(l + 1, 0) : slice_of(l, i + 1);
            const int kcur = kbase_of(l) + q * 36, knext = (last ? kbase_of(l + 1) : kbase_of(l)) + qn * 36;
            slice_steps(kcur, knext, q, qn);
        }
        if (wave == 0) T2S_STAMP(l, 3);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                   // [R]
        if (l == 0) boundary(std::integral_constant<int, 0>{});
        else if (l & 1) boundary(std::integral_constant<int, 1>{});
        else boundary(std::integral_constant<int, 2>{});
        load_shift(l + 1);
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                   // [W]
        if (wave == 0) T2S_STAMP(l, 4);
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
        const int b = b0 + (pt >> 1);
        if (b >= a.B) continue;
        const int pix = (4 * (pt & 1) + py) * 8 + px;
        const size_t row = ((size_t)b * 64 + pix) * CH + 32 * rt;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(xr[pt][2 * j][0], xr[pt][2 * j + 1][0], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(xr[pt][2 * j][1], xr[pt][2 * j + 1][1], false, false);
            const u32x4 o = { s0[0], s1[0], s0[1], s1[1] };
            *reinterpret_cast<u32x4*>(a.out + row + 16 * j + 8 * h) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// policy_head4_kernel — the whole policy head (nn.cpp:72-80) of four boards per workgroup in one launch:
// policyconv 1x1 (C -> 128) + pbatchnorm + ReLU, policyconv2 1x1 (128 -> 73) + bias, softmax over the board's 4 672
// logits, policy row out.  conv4's tiling again: a wave owns one board and all output channels, so the 128-channel
// intermediate goes back into the wave's own image (no HBM, no barrier) and the softmax is a WAVE reduction over the
// 8 accumulators (planes >= 73 are padding).  Replaces three launches that moved 16.8 MB (intermediate) + 2 x 19 MB
// (fp32 logits written, then read by the softmax) per 1 024 boards for 6.6 MFLOP per board.  The convolutions walk
// the reduction in conv_mfma_kernel's order (logits bit-identical); the softmax sums in another order than
// softmax4672_kernel, and is used at every batch size of a configuration, so batch splits still agree bit for bit.
template <typename T, int NP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void policy_head4_kernel(Head4Args a)
{
    constexpr int RDN = 4;
    constexpr int LDS_IMG = RDN * CHUNKB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using V = typename Elem<T>::vec8;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int Ci = 128, CiTot = NP * Ci;
    constexpr int stride = Ci * 2 + 16;
    constexpr int board_bytes = 64 * stride;               // 1x1 convolutions: no halo
    const int b0 = blockIdx.x * 4;
    constexpr int NC1 = NP * 4, NCHT = NC1 + 4;
    const char* stream = reinterpret_cast<const char*>(a.w);
    char* img = smem + LDS_IMG;
#pragma unroll
    for (int i = 0; i < RDN - 1; ++i) ring_issue<RDN>(stream, NCHT, i, wave, lane);
    auto stage = [&](int pass) {
        const unsigned short* src = a.x + pass * Ci;
        u32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            const int bb = i >> 10, p = (i >> 4) & 63, c = i & 15;
            v[u] = u32x4{ 0, 0, 0, 0 };
            if (b0 + bb < a.B) v[u] = *reinterpret_cast<const u32x4*>(src + ((size_t)(b0 + bb) * 64 + p) * CiTot + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = tid + u * 256;
            const int bb = i >> 10, p = (i >> 4) & 63, c = i & 15;
            *reinterpret_cast<u32x4*>(img + bb * board_bytes + p * stride + c * 16) = v[u];
        }
    };
    const int lp = PIXMAP[lane & 31];                       // this lane's pixel within a 32-pixel half
    const unsigned b_base = LDS_IMG + wave * board_bytes + lp * stride + h * 16;
    constexpr unsigned HALF = 32 * stride;
    char* const w_base = smem + LDS_IMG + wave * board_bytes + lp * stride + h * 8;
    f32x16 acc[8];
    auto init_acc = [&](const float* sh) {
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 s4 = *reinterpret_cast<const float4*>(sh + ms * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    acc[ms * 2 + hp][4 * g + 0] = s4.x; acc[ms * 2 + hp][4 * g + 1] = s4.y;
                    acc[ms * 2 + hp][4 * g + 2] = s4.z; acc[ms * 2 + hp][4 * g + 3] = s4.w;
                }
            }
    };
    init_acc(a.shift1);
    auto chunk_off = [](int n) -> unsigned { return (unsigned)((n >> 1) * 128 + (n & 1) * 64); };   // slice, half of the slice
    V A[2][8], Bq[2][4];
    // four chunk steps: conv1's of one pass (G0 = 4 * pass), or conv2's (G0 = NC1)
    auto four_steps = [&](int G0) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int hp = 0; hp < 2; ++hp) Bq[0][k * 2 + hp] = *reinterpret_cast<const V*>(smem + b_base + hp * HALF + chunk_off(0) + k * 32);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int cur = n & 1, nxt = cur ^ 1, g = G0 + n;
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * (RDN - 3)) : "memory");
            __builtin_amdgcn_sched_barrier(0);
            ring_issue<RDN>(stream, NCHT, g + RDN - 1, wave, lane);
            const unsigned a_off = (unsigned)(((n + 1) % RDN) * CHUNKB) + lane * 16;       // 4 chunks per call: slot = n
#pragma unroll
            for (int f = 0; f < 8; ++f) A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
            if (n + 1 < 4) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) Bq[nxt][k * 2 + hp] = *reinterpret_cast<const V*>(smem + b_base + hp * HALF + chunk_off(n + 1) + k * 32);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int ms = 0; ms < 4; ++ms)
#pragma unroll
                    for (int hp = 0; hp < 2; ++hp) acc[ms * 2 + hp] = Elem<T>::mfma(A[cur][k * 4 + ms], Bq[cur][k * 2 + hp], acc[ms * 2 + hp]);
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };
    float vsum[2] = { 0.0f, 0.0f };                         // valueconv partial sums of this lane's two pixels, its half of the channels
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        if (pass > 0) __syncthreads();
        stage(pass);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (RDN - 2)) : "memory");
        if (pass == 0) {
#pragma unroll
            for (int f = 0; f < 8; ++f) A[0][f] = *reinterpret_cast<const V*>(smem + lane * 16 + f * 1024);
        }
        if (a.vw) {
            // valueconv (C -> 1) on the staged board: lane (pixel, h) takes channels 64 h .. 64 h + 63 of this pass
            const float* vw = a.vw + pass * Ci + 64 * h;
#pragma unroll
            for (int c8 = 0; c8 < 8; ++c8) {
                const float4 w0 = *reinterpret_cast<const float4*>(vw + c8 * 8), w1 = *reinterpret_cast<const float4*>(vw + c8 * 8 + 4);
#pragma unroll
                for (int hp = 0; hp < 2; ++hp) {
                    const char* row = smem + LDS_IMG + wave * board_bytes + (32 * hp + lp) * stride + h * 128;
                    const u32x4 u = *reinterpret_cast<const u32x4*>(row + c8 * 16);
                    float s = vsum[hp];
                    s = fmaf(from_bits<T>((unsigned short)(u.x & 0xffff)), w0.x, s); s = fmaf(from_bits<T>((unsigned short)(u.x >> 16)), w0.y, s);
                    s = fmaf(from_bits<T>((unsigned short)(u.y & 0xffff)), w0.z, s); s = fmaf(from_bits<T>((unsigned short)(u.y >> 16)), w0.w, s);
                    s = fmaf(from_bits<T>((unsigned short)(u.z & 0xffff)), w1.x, s); s = fmaf(from_bits<T>((unsigned short)(u.z >> 16)), w1.y, s);
                    s = fmaf(from_bits<T>((unsigned short)(u.w & 0xffff)), w1.z, s); s = fmaf(from_bits<T>((unsigned short)(u.w >> 16)), w1.w, s);
                    vsum[hp] = s;
                }
            }
        }
        four_steps(pass * 4);
    }
    // policyconv's output (pbatchnorm folded, ReLU, rounded to T like the per-layer path's intermediate) -> this wave's image
#pragma unroll
    for (int hp = 0; hp < 2; ++hp)
#pragma unroll
        for (int ms = 0; ms < 4; ++ms)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(acc[ms * 2 + hp][4 * g + i]);
                *reinterpret_cast<u32x2*>(w_base + hp * HALF + (ms * 32 + 8 * g) * 2) = u32x2{ pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]) };
            }
    init_acc(a.bias2);
    four_steps(NC1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the ring (tail re-fetches) before exit

    // ---- softmax over the board's 64 pixels x 73 planes (nn.cpp:78-80: exp(log_softmax)), all inside this wave
    const int b = b0 + wave;
    float m = -INFINITY;
    bool nan = false;
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = (i8 >> 1) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (ch < KH_POLICY_PLANES) { const float v = acc[i8][r]; nan |= (v != v); m = fmaxf(m, v); }
        }
    m = wave_max_f(m);
    float sum = 0.0f;
#pragma unroll
    for (int i8 = 0; i8 < 8; ++i8)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = (i8 >> 1) * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (ch < KH_POLICY_PLANES) sum += HEAD_EXP(acc[i8][r] - m);
        }
    sum = wave_sum_f(sum);
    const float ls = HEAD_LOG(sum);
    const bool row_nan = __any(nan);                        // any NaN logit poisons the whole row in the reference
    // The lane layout (4 consecutive planes of one pixel per lane, pixels 292 bytes apart) makes a direct store 64
    // scattered 4-byte writes per instruction.  The row goes through LDS instead (ring and images are dead: one barrier
    // so that no wave is still reading them): [pixel][73] fp32 = the row's own memory order, then 16-byte coalesced
    // non-temporal stores of the wave's 18 688 contiguous bytes.
    __syncthreads();
    float* rowbuf = reinterpret_cast<float*>(smem) + wave * KH_PSIZE;
    const bool live = b < a.B;
    auto put_row = [&](float* dst_row, bool probs) {
#pragma unroll
        for (int hp = 0; hp < 2; ++hp)
#pragma unroll
            for (int ms = 0; ms < 3; ++ms)                  // planes 96..127 are padding
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = ms * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    if (ch < KH_POLICY_PLANES) {
                        const float v = acc[ms * 2 + hp][r];
                        rowbuf[(32 * hp + lp) * KH_POLICY_PLANES + ch] = probs ? (row_nan ? NAN : HEAD_EXP((v - m) - ls)) : v;
                    }
                }
        // (same wave wrote it: its LDS operations are ordered)
        if (live) {
            for (int i = lane; i < KH_PSIZE / 4; i += 64)
                // write-through (sc1): the rows are final outputs of the forward's last launch — nothing of them stays dirty
                // in L2 for the end-of-kernel release to write back (tower8_mfma.hip: store_wt)
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(reinterpret_cast<f32x4s*>(dst_row) + i), "v"(*reinterpret_cast<const f32x4s*>(rowbuf + 4 * i)) : "memory");
        }
    };
    if (a.logits) put_row(a.logits + (size_t)(live ? b : 0) * KH_PSIZE, false);
    put_row(a.policy + (size_t)(live ? b : 0) * KH_PSIZE, true);
    if (!live) return;
    if (row_nan && lane == 0) atomicOr(&a.flags[0], 1);
    // ---- value head: vbatchnorm + ReLU on the 64 valueconv sums, Linear(64, 256), tanh (nn.cpp:84-88); this wave's board
    if (a.vw) {
        float* v64 = reinterpret_cast<float*>(smem + LDS_IMG + 4 * board_bytes) + wave * 64;
#pragma unroll
        for (int hp = 0; hp < 2; ++hp) {
            // the two channel halves of a pixel sit in lanes l and l + 32
            const unsigned u = __float_as_uint(vsum[hp]);
            const auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            const float tot = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            if (h == 0) v64[32 * hp + lp] = relu_keep_nan(tot + a.vshift);
        }
        // (same wave wrote it: its LDS operations are ordered)
        // (the same sums as tower128_kernel's fused heads: lane l takes outputs l, l + 64, l + 128, l + 192 out of fc4)
        float o[4] = { a.fcb[lane], a.fcb[lane + 64], a.fcb[lane + 128], a.fcb[lane + 192] };
#pragma unroll 4
        for (int k4 = 0; k4 < 16; ++k4) {
            const float4 hv = *reinterpret_cast<const float4*>(v64 + k4 * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 wv = *reinterpret_cast<const float4*>(a.fc4 + ((size_t)k4 * KH_VALUE_WIDTH + lane + 64 * j) * 4);
                o[j] = fmaf(hv.x, wv.x, o[j]); o[j] = fmaf(hv.y, wv.y, o[j]); o[j] = fmaf(hv.z, wv.z, o[j]); o[j] = fmaf(hv.w, wv.w, o[j]);
            }
        }
        bool vnan = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float r = tanhf(o[j]);
            vnan |= (r != r);
            a.vfull[(size_t)b * KH_VALUE_WIDTH + lane + 64 * j] = r;
        }
        if (vnan) atomicOr(&a.flags[1], 1);
    }
}

// valueconv + vbatchnorm + relu (nn.cpp:83-85) on T activations: one thread per (board, pixel)
template <typename T>
__global__ __launch_bounds__(256) void value_conv_kernel(const unsigned short* __restrict__ x, const float* __restrict__ vw,
                                                         float vshift, float* __restrict__ v64, long npix, int C)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const unsigned short* xp = x + i * C;
        float s = 0.0f;
        for (int c = 0; c < C; c += 8) {
            const u32x4 u = *reinterpret_cast<const u32x4*>(xp + c);
            const unsigned w[4] = { u.x, u.y, u.z, u.w };
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s = fmaf(from_bits<T>((unsigned short)(w[k] & 0xffff)), vw[c + 2 * k], s);
                s = fmaf(from_bits<T>((unsigned short)(w[k] >> 16)), vw[c + 2 * k + 1], s);
            }
        }
        v64[i] = relu_keep_nan(s + vshift);
    }
}

// ---------------------------------------------------------------------------------------------
// fp32 on the matrix cores: v_mfma_f32_32x32x2_f32 is an exact f32 fmaf chain (1/16 of the bf16
// rate, 157 TFLOP/s peak).  Same structure as conv_mfma_kernel with fp32 activations [B][64][C]:
// per 8 input channels one ds_read_b128 (lane (pixel r, h) holds channels 8j + 4h + 0..3), two
// 16-byte weight loads (lane (co r, h) holds W[co][8j + 4h + 0..3]) and 8 MFMAs — MFMA i pairs
// element i of both, i.e. k = h <-> channel 8j + 4h + i on either side.
using f32x4v = __attribute__((ext_vector_type(4))) float;

struct ConvArgsF32 {
    const float* in;      // [B][64][Ci]
    const float* w;       // packed [Co/64][Ci slices of <= 128][taps][slice/8][2][64 lanes][4]  (one slice when Ci <= 128)
    const float* shift;   // [Co] initial value of the accumulators (folded BatchNorm shift / bias), nullptr = 0
    const float* skip;    // [B][64][Co] or nullptr
    float* out;           // [B][64][Co] (EPI 0/1/3/4) or [B][4672] (EPI 2)
    int B, Ci, Co;        // Ci % 8 == 0; the kernel computes Co rounded up to 64 and stores channels < Co
};

// EPI: 0 = ReLU; 1 = ReLU, + skip (nn.cpp:31); 2 = raw fp32 logits, planes < 73 (nn.cpp:75-79);
//      3 = raw (training: the convolution + bias, BatchNorm follows in its own kernels); 4 = raw, added to what
//      `out` already holds (training: a data gradient that joins another one)
// PLAIN: one staged slice (Ci <= 128), whole 64-channel blocks (Co % 64 == 0) and a shift vector — the inference shapes
// of a <= 128-filter net.  The general form (slice loop with a run-time trip count, partial channel blocks, optional
// shift: added in round 2 for 256 filters and the trainer) made the SAME arithmetic 6-8 % slower at 64 filters
// (conv_f32_kernel<9,0> 29.6 -> 32.0 us per layer at batch 512, same box, profiles/r03_f32_ab.txt): the branches are
// compile-time here and the generated loop is round 1's again.
template <int TAPS, int EPI, bool PLAIN = false>
__global__ __launch_bounds__(256) void conv_f32_kernel(ConvArgsF32 a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int Ci = a.Ci, Co = a.Co;
    const int CS = PLAIN ? Ci : (Ci < 128 ? Ci : 128);                   // channels of one staged slice (Ci > 128: several passes)
    const int stride = CS * 4 + 16;                       // (CS/4 + 1) 16-byte slots: odd -> conflict-free
    const int npx = (TAPS == 9) ? NPIX : 64;
    const int board_bytes = npx * stride;
    const int b0 = blockIdx.x * 2, cb = blockIdx.y;
    if (TAPS == 9) {
        const u32x4 z = { 0, 0, 0, 0 };
        const int per_px = stride / 16;
        for (int i = tid; i < 2 * NPIX; i += 256) {
            const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
            if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
            char* d = smem + (i / NPIX) * board_bytes + pp * stride;
            for (int k = 0; k < per_px; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
        }
    }
    const int wb = wave >> 1;
    const int lp = PIXMAP[lane & 31];
    const int py = 4 * (wave & 1) + (lp >> 3), px = lp & 7;
    const unsigned b_base = wb * board_bytes + ((TAPS == 9) ? (py * PITCH + px) : (py * 8 + px)) * stride + h * 16;
    f32x16 acc[2];
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            const int c0 = cb * 64 + ms * 32 + 8 * g + 4 * h;
            if (PLAIN) s = *reinterpret_cast<const float4*>(a.shift + c0);
            else if (a.shift) {
                if (c0 + 3 < Co) s = *reinterpret_cast<const float4*>(a.shift + c0);
                else { if (c0 < Co) s.x = a.shift[c0]; if (c0 + 1 < Co) s.y = a.shift[c0 + 1]; if (c0 + 2 < Co) s.z = a.shift[c0 + 2]; }
            }
            acc[ms][4 * g + 0] = s.x; acc[ms][4 * g + 1] = s.y; acc[ms][4 * g + 2] = s.z; acc[ms][4 * g + 3] = s.w;
        }
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    for (int c_lo = 0; c_lo < (PLAIN ? 1 : Ci); c_lo += 128) {
        const int cs = PLAIN ? Ci : (Ci - c_lo < 128 ? Ci - c_lo : 128), KJ = cs / 8, TK = TAPS * KJ;
        if (!PLAIN && c_lo) __syncthreads();              // everybody is done reading the previous slice
        {
            const int CH = cs / 4;
            for (int i = tid; i < 2 * 64 * CH; i += 256) {
                const int bb = i / (64 * CH), p = (i / CH) & 63, c = i % CH;
                u32x4 v = { 0, 0, 0, 0 };
                if (b0 + bb < a.B) v = *reinterpret_cast<const u32x4*>(a.in + ((size_t)(b0 + bb) * 64 + p) * Ci + c_lo + c * 4);
                const int pix = (TAPS == 9) ? ((p >> 3) + 1) * PITCH + (p & 7) + 1 : p;
                *reinterpret_cast<u32x4*>(smem + bb * board_bytes + pix * stride + c * 16) = v;
            }
        }
        __syncthreads();
        // 256 floats per fragment; this block's slices are consecutive, each TAPS * (slice / 8) * 2 fragments
        const float* wp = a.w + ((size_t)cb * TAPS * (Ci / 8) * 2 + (size_t)TAPS * (c_lo / 8) * 2) * 256 + lane * 4;
        auto boff = [&](int kk) -> unsigned {
            const int tap = kk / KJ, j = kk - tap * KJ;
            return ((TAPS == 9) ? (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) : 0u) + j * 32;
        };
        // operands two k-groups ahead in flight: one group is 8 MFMAs x 64 cycles, about one L2 round trip
        auto ld = [&](int kk, f32x4v& x0, f32x4v& x1, f32x4v& xb) {
            const int k = kk < TK ? kk : TK - 1;                       // past the end: harmless re-read
            xb = *reinterpret_cast<const f32x4v*>(smem + b_base + boff(k));
            x0 = *reinterpret_cast<const f32x4v*>(wp + (size_t)k * 512);
            x1 = *reinterpret_cast<const f32x4v*>(wp + (size_t)k * 512 + 256);
        };
        // Three operand sets in fixed roles (no register rotation): k-group kk multiplies out of set kk % 3
        // while the loads of k-group kk + 2 land in set (kk + 2) % 3.  The loads are pinned to the top of
        // each step — left alone hipcc sinks them down to their first use two steps later and every step
        // then waits for a full L2 round trip.
        f32x4v wa[3], wb2[3], xb[3];
        ld(0, wa[0], wb2[0], xb[0]);
        ld(1, wa[1], wb2[1], xb[1]);
#define KH_F32_STEP(kk, CUR, NXT2)                                                                     \
    {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        ld((kk) + 2, wa[NXT2], wb2[NXT2], xb[NXT2]);                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[CUR][i], xb[CUR][i], acc[0], 0, 0, 0);    \
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb2[CUR][i], xb[CUR][i], acc[1], 0, 0, 0);   \
        }                                                                                              \
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);   /* the two weight loads */                \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   /* the activation read */                 \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   /* then the MFMAs */                      \
    }
        int kk = 0;
        for (; kk + 2 < TK; kk += 3) {
            KH_F32_STEP(kk, 0, 2)
            KH_F32_STEP(kk + 1, 1, 0)
            KH_F32_STEP(kk + 2, 2, 1)
        }
        if (kk < TK) { KH_F32_STEP(kk, 0, 2) ++kk; }
        if (kk < TK) { KH_F32_STEP(kk, 1, 0) }
#undef KH_F32_STEP
    }
    const int b = b0 + wb;
    if (b >= a.B) return;
    const int p = py * 8 + px;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = cb * 64 + ms * 32 + 8 * g + 4 * h;
            float v[4] = { acc[ms][4 * g], acc[ms][4 * g + 1], acc[ms][4 * g + 2], acc[ms][4 * g + 3] };
            if (EPI == 2) {
                float* lo = a.out + (size_t)b * KH_PSIZE + p * KH_POLICY_PLANES;
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (ch + i < KH_POLICY_PLANES) lo[ch + i] = v[i];
            } else {
                const size_t o = ((size_t)b * 64 + p) * Co + ch;
                if (EPI <= 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(v[i]);
                }
                if (PLAIN || (ch + 3 < Co && (Co & 3) == 0)) {
                    if (EPI == 1) {
                        const float4 s = *reinterpret_cast<const float4*>(a.skip + o);
                        v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                    }
                    if (EPI == 4) {
                        const float4 s = *reinterpret_cast<const float4*>(a.out + o);
                        v[0] += s.x; v[1] += s.y; v[2] += s.z; v[3] += s.w;
                    }
                    *reinterpret_cast<float4*>(a.out + o) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (ch + i < Co) {
                            float r = v[i];
                            if (EPI == 1) r += a.skip[o + i];
                            if (EPI == 4) r += a.out[o + i];
                            a.out[o + i] = r;
                        }
                }
            }
        }
}

// Small batches (the trainer's 8..64 boards): conv_f32_kernel's workgroup is 2 boards x 64 channels and a wave walks the
// whole reduction alone — 16 workgroups at batch 32, each 144 k-groups x 8 MFMAs x 64 clocks long.  Here a workgroup is
// ONE board x 32 channels and its four waves are (pixel half) x (HALF of the reduction): 8x the workgroups, a quarter of
// the MFMAs per wave; the two halves of the reduction meet through LDS (fixed order: first half + second half).
template <int TAPS, int EPI>
__global__ __launch_bounds__(256) void conv_f32_small_kernel(ConvArgsF32 a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using f32x4v = __attribute__((ext_vector_type(4))) float;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const int Ci = a.Ci, Co = a.Co;
    const int CS = Ci < 128 ? Ci : 128;
    const int stride = CS * 4 + 16;
    const int b = blockIdx.x, cb = blockIdx.y;           // cb: 32-channel block
    const int ph = wave & 1, kh = wave >> 1;
    if (TAPS == 9) {
        const u32x4 z = { 0, 0, 0, 0 };
        const int per_px = stride / 16;
        for (int i = tid; i < NPIX; i += 256) {
            const int yy = i / PITCH, xx = i % PITCH;
            if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
            char* d = smem + i * stride;
            for (int k = 0; k < per_px; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
        }
    }
    const int lp = PIXMAP[lane & 31];
    const int py = 4 * ph + (lp >> 3), px = lp & 7;
    const unsigned b_base = ((TAPS == 9) ? (py * PITCH + px) : (py * 8 + px)) * stride + h * 16;
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c = cb * 32 + 8 * g + 4 * h + i;
            acc[4 * g + i] = (kh == 0 && a.shift && c < Co) ? a.shift[c] : 0.0f;
        }
    for (int c_lo = 0; c_lo < Ci; c_lo += 128) {
        const int cs = Ci - c_lo < 128 ? Ci - c_lo : 128, KJ = cs / 8, TK = TAPS * KJ;
        if (c_lo) __syncthreads();
        {
            const int CH = cs / 4;
            for (int i = tid; i < 64 * CH; i += 256) {
                const int p = i / CH, c = i % CH;
                const u32x4 v = *reinterpret_cast<const u32x4*>(a.in + ((size_t)b * 64 + p) * Ci + c_lo + c * 4);
                const int pix = (TAPS == 9) ? ((p >> 3) + 1) * PITCH + (p & 7) + 1 : p;
                *reinterpret_cast<u32x4*>(smem + pix * stride + c * 16) = v;
            }
        }
        __syncthreads();
        // fragments of 64-channel block cb >> 1, row tile cb & 1 (the packing interleaves the two row tiles)
        const float* wp = a.w + ((size_t)(cb >> 1) * TAPS * (Ci / 8) * 2 + (size_t)TAPS * (c_lo / 8) * 2) * 256 + (cb & 1) * 256 + lane * 4;
        const int k_lo = kh ? (TK + 1) / 2 : 0, k_hi = kh ? TK : (TK + 1) / 2;
        auto boff = [&](int kk) -> unsigned {
            const int tap = kk / KJ, j = kk - tap * KJ;
            return ((TAPS == 9) ? (unsigned)(((tap / 3) * PITCH + (tap % 3)) * stride) : 0u) + j * 32;
        };
        auto ld = [&](int kk, f32x4v& x0, f32x4v& xb) {
            const int k = kk < k_hi ? kk : (k_hi > k_lo ? k_hi - 1 : 0);
            xb = *reinterpret_cast<const f32x4v*>(smem + b_base + boff(k));
            x0 = *reinterpret_cast<const f32x4v*>(wp + (size_t)k * 512);
        };
        f32x4v wa[3], xb[3];
        ld(k_lo, wa[0], xb[0]);
        ld(k_lo + 1, wa[1], xb[1]);
#define KH_F32S_STEP(kk, CUR, NXT2)                                                                    \
    {                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                             \
        ld((kk) + 2, wa[NXT2], xb[NXT2]);                                                              \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                  \
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[CUR][i], xb[CUR][i], acc, 0, 0, 0);          \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                             \
    }
        int kk = k_lo;
        for (; kk + 2 < k_hi; kk += 3) {
            KH_F32S_STEP(kk, 0, 2)
            KH_F32S_STEP(kk + 1, 1, 0)
            KH_F32S_STEP(kk + 2, 2, 1)
        }
        if (kk < k_hi) { KH_F32S_STEP(kk, 0, 2) ++kk; }
        if (kk < k_hi) { KH_F32S_STEP(kk, 1, 0) }
#undef KH_F32S_STEP
    }
    // second half of the reduction -> LDS -> first half's waves (the image is dead)
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + (size_t)ph * 64 * 16;
    if (kh == 1) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[i * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (kh == 1) return;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] += red[i * 64 + lane];
    const int p = py * 8 + px;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int ch = cb * 32 + 8 * g + 4 * h;
        float v[4] = { acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3] };
        const size_t o = ((size_t)b * 64 + p) * Co + ch;
        if (EPI <= 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = relu_keep_nan(v[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (ch + i < Co) {
                float r = v[i];
                if (EPI == 1) r += a.skip[o + i];
                if (EPI == 4) r += a.out[o + i];
                a.out[o + i] = r;
            }
    }
}

// fp32 planes [B][64][F] -> [B][64][FP] zero-padded (FP = F rounded up to 8)
__global__ __launch_bounds__(256) void pad_planes_kernel(const float* __restrict__ in, float* __restrict__ out, long npix, int F, int FP)
{
    const long total = npix * FP;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long p = i / FP;
        const int c = (int)(i % FP);
        out[i] = c < F ? in[p * F + c] : 0.0f;
    }
}

__global__ __launch_bounds__(256) void value_conv_f32_kernel(const float* __restrict__ x, const float* __restrict__ vw,
                                                             float vshift, float* __restrict__ v64, long npix, int C)
{
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (long)gridDim.x * blockDim.x) {
        const float* xp = x + i * C;
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s = fmaf(xp[c], vw[c], s);
        v64[i] = relu_keep_nan(s + vshift);
    }
}

template <int TAPS, int EPI> static hipError_t launch_conv_f32(const ConvArgsF32& a, hipStream_t s)
{
    const int lds = 2 * ((TAPS == 9) ? NPIX : 64) * ((a.Ci < 128 ? a.Ci : 128) * 4 + 16);
    static std::atomic<bool> attr_done{ false };      // engines are called from many host threads; setting it twice is harmless
    if (!attr_done.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_kernel<TAPS, EPI>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done.store(true, std::memory_order_release);
    }
    // few boards (the trainer): one board x 32 channels per workgroup, the reduction split over its waves
    if constexpr (EPI == 3 || EPI == 4) {
        if ((long)((a.B + 1) / 2) * ((a.Co + 63) / 64) < 128) {
            static std::atomic<bool> attr2_done{ false };
            if (!attr2_done.load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_small_kernel<TAPS, EPI>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr2_done.store(true, std::memory_order_release);
            }
            const int lds1 = std::max(lds / 2, 2 * 64 * 16 * 4);
            hipLaunchKernelGGL((conv_f32_small_kernel<TAPS, EPI>), dim3(a.B, (a.Co + 31) / 32), dim3(256), lds1, s, a);
            return hipGetLastError();
        }
    }
    if constexpr (EPI <= 2) {
        if (a.Ci <= 128 && a.Co % 64 == 0 && a.shift) {
            static std::atomic<bool> attr3_done{ false };
            if (!attr3_done.load(std::memory_order_acquire)) {
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_kernel<TAPS, EPI, true>),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                if (e != hipSuccess) return e;
                attr3_done.store(true, std::memory_order_release);
            }
            hipLaunchKernelGGL((conv_f32_kernel<TAPS, EPI, true>), dim3((a.B + 1) / 2, a.Co / 64), dim3(256), lds, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((conv_f32_kernel<TAPS, EPI>), dim3((a.B + 1) / 2, (a.Co + 63) / 64), dim3(256), lds, s, a);
    return hipGetLastError();
}

static hipError_t run_f32(const LayersArgs& L, hipStream_t s)
{
    const long npix = (long)L.B * 64;
    const float* in = L.in;
    if (L.FP != L.F) {
        int blocks = (int)((npix * L.FP + 255) / 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(pad_planes_kernel, dim3(blocks), dim3(256), 0, s, L.in, reinterpret_cast<float*>(L.act_in), npix, L.F, L.FP);
        in = reinterpret_cast<const float*>(L.act_in);
    }
    float *x = reinterpret_cast<float*>(L.act[0]), *t = reinterpret_cast<float*>(L.act[1]), *u = reinterpret_cast<float*>(L.act[2]);
    const float* wbase = reinterpret_cast<const float*>(L.w);
    hipError_t e;
    size_t li = 0;
    ConvArgsF32 a;
    a.B = L.B;
    a.in = in; a.w = wbase + L.w_off[li]; a.shift = L.shift + L.shift_off[li]; a.skip = nullptr; a.out = x; a.Ci = L.FP; a.Co = L.CP; ++li;
    if ((e = launch_conv_f32<9, 0>(a, s)) != hipSuccess) return e;
    for (int r = 0; r < L.R; ++r) {
        a.in = x; a.w = wbase + L.w_off[li]; a.shift = L.shift + L.shift_off[li]; a.skip = nullptr; a.out = t; a.Ci = L.CP; a.Co = L.CP; ++li;
        if ((e = launch_conv_f32<9, 0>(a, s)) != hipSuccess) return e;
        a.in = t; a.w = wbase + L.w_off[li]; a.shift = L.shift + L.shift_off[li]; a.skip = x; a.out = u; ++li;
        if ((e = launch_conv_f32<9, 1>(a, s)) != hipSuccess) return e;
        float* tmp = x; x = u; u = tmp;
    }
    a.in = x; a.w = wbase + L.w_off[li]; a.shift = L.shift + L.shift_off[li]; a.skip = nullptr; a.out = reinterpret_cast<float*>(L.pmid); a.Ci = L.CP; a.Co = KH_POLICY_MID; ++li;
    if ((e = launch_conv_f32<1, 0>(a, s)) != hipSuccess) return e;
    a.in = reinterpret_cast<const float*>(L.pmid); a.w = wbase + L.w_off[li]; a.shift = L.shift + L.shift_off[li]; a.out = L.logits; a.Ci = KH_POLICY_MID; a.Co = 128; ++li;
    if ((e = launch_conv_f32<1, 2>(a, s)) != hipSuccess) return e;
    launch_softmax4672(L.logits, L.policy, L.B, L.flags, s);                  // nn.cpp:80
    int vb = (int)((npix + 255) / 256);
    if (vb > 4096) vb = 4096;
    hipLaunchKernelGGL(value_conv_f32_kernel, dim3(vb), dim3(256), 0, s, x, L.vw, L.vshift, L.v64, npix, L.CP);
    launch_value_fc(L.v64, L.fcw, L.fcb, L.vfull, L.B, L.flags, s);           // nn.cpp:86-88
    return hipGetLastError();
}

template <typename T, int TAPS, int EPI, int CPT, int RDN, int NP = 1> static hipError_t launch_conv_rd(const ConvArgs& a, hipStream_t s)
{
    const int stride = CPT * 64 * 2 + 16;                  // image of one pass
    const int image = 2 * ((TAPS == 9) ? NPIX : 64) * stride, tiles = 4 * 32 * (64 * 4 + 16);   // the epilogue's transpose tiles reuse the image
    const int lds = RDN * CHUNKB + (image > tiles ? image : tiles);
    static std::atomic<bool> attr_done{ false };      // engines are called from many host threads; setting it twice is harmless
    if (!attr_done.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mfma_kernel<T, TAPS, EPI, CPT, RDN, NP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<T, TAPS, EPI, CPT, RDN, NP>), dim3((a.B + 1) / 2, a.Co / 64), dim3(256), lds, s, a);
    return hipGetLastError();
}

template <typename T, int EPI, int NP> static hipError_t launch_conv4(const ConvArgs& a, hipStream_t s)
{
    const int lds = 4 * CHUNKB + 4 * NPIX * (128 * 2 + 16);          // 163 328 B: one workgroup per CU
    static std::atomic<bool> attr_done{ false };
    if (!attr_done.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv4_mfma_kernel<T, EPI, NP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done.store(true, std::memory_order_release);
    }
    ConvArgs b = a;
    b.w = a.w4;
    hipLaunchKernelGGL((conv4_mfma_kernel<T, EPI, NP>), dim3((a.B + 3) / 4, a.Co / 128), dim3(256), lds, s, b);
    return hipGetLastError();
}

template <typename T, int TAPS, int EPI, int CPT> static hipError_t launch_conv_cpt(const ConvArgs& a, hipStream_t s)
{
    // 3x3 layers with more workgroups than CUs: the 2-slot-ring variants, two workgroups per CU (128 input
    // channels; 256 as two passes of 128) or three (passes of 64 channels) when there are enough of them.
    // 10x128 at batch 1024: 726 -> 600 us per forward with two; with one workgroup per CU anyway the short ring
    // only costs (219 -> 251 us at batch 256), so this is decided per launch.
    const long wgs = (long)((a.B + 1) / 2) * (a.Co / 64);
    static const int force = getenv("KAMI_WIDE_VARIANT") ? atoi(getenv("KAMI_WIDE_VARIANT")) : 0;   // experiments: 1, 2, 3 workgroups per CU; 4 = four boards per workgroup
    // four boards x 128 output channels per workgroup (conv4_mfma_kernel) once that still gives every CU a workgroup
    if constexpr (TAPS == 9 && EPI != 2 && (CPT == 2 || CPT == 4)) {
        const long wg4 = (long)((a.B + 3) / 4) * (a.Co / 128);
        if (a.w4 && a.Co % 128 == 0 && (force == 4 || (!force && wg4 >= 256))) return launch_conv4<T, EPI, CPT / 2>(a, s);
    }
    // measured (tools/wide_variants.py): 128 channels: three per CU is ahead from 512 workgroups on (+2 %, +14 % at
    // 2048); 256 channels: two and three are within 2 % of each other either way
    if constexpr (TAPS == 9 && CPT == 2) {
        const int per_cu = force ? force : (wgs > 256 ? 3 : 1);
        if (per_cu == 3) return launch_conv_rd<T, TAPS, EPI, 1, 2, 2>(a, s);
        if (per_cu == 2) return launch_conv_rd<T, TAPS, EPI, 2, 2>(a, s);
    }
    if constexpr (TAPS == 9 && CPT == 4) {
        const int per_cu = force ? force : (wgs > 256 ? 2 : 1);
        if (per_cu == 3) return launch_conv_rd<T, TAPS, EPI, 1, 2, 4>(a, s);
        if (per_cu == 2) return launch_conv_rd<T, TAPS, EPI, 2, 2, 2>(a, s);
    }
    return launch_conv_rd<T, TAPS, EPI, CPT, RD>(a, s);
}

template <typename T, int TAPS, int EPI> static hipError_t launch_conv(const ConvArgs& a, hipStream_t s)
{
    switch (a.Ci / 64) {                   // input channels are padded to a multiple of 64, at most 256
    case 1: return launch_conv_cpt<T, TAPS, EPI, 1>(a, s);
    case 2: return launch_conv_cpt<T, TAPS, EPI, 2>(a, s);
    case 3: return launch_conv_cpt<T, TAPS, EPI, 3>(a, s);
    case 4: return launch_conv_cpt<T, TAPS, EPI, 4>(a, s);
    default: return hipErrorInvalidValue;
    }
}

template <typename T> static hipError_t run(const LayersArgs& L, hipStream_t s)
{
    const long npix = (long)L.B * 64;
    int blocks = (int)((npix * (L.FP / 8) + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    static const int force = getenv("KAMI_WIDE_VARIANT") ? atoi(getenv("KAMI_WIDE_VARIANT")) : 0;
    // 128 planes (padded) and 128 filters: the whole 3x3 stack in one launch, activations on chip (tower128_kernel),
    // reading the fp32 planes itself.  One workgroup per four boards: worth it once that keeps most CUs busy.
    const bool fused = L.FP == 128 && L.CP == 128 && L.w4 && L.w4_off[0] != (size_t)-1 && force != 6 && (force == 5 || (!force && L.B >= 640));
    const bool direct = fused && (reinterpret_cast<uintptr_t>(L.in) & 15) == 0;
    // two boards per workgroup, waves own output channels (tower2b_kernel): 256 filters from batch 256 on; 128 filters
    // for the batches where tower128_kernel's four boards per workgroup do not fill the chip (KAMI_WIDE_VARIANT=6 forces it).
    // It reads the fp32 planes itself: ONE condition decides both that and whether planes_to_act_kernel runs (they used to
    // disagree for 256 filters at 256 <= batch < 384: a launch whose output nobody read).
    const bool aligned = (reinterpret_cast<uintptr_t>(L.in) & 15) == 0;
    // 256 filters, up to one board pair per TWO CUs: two workgroups per pair, each half the output channels, halves
    // exchanged per layer (tower2s_kernel; KAMI_WIDE_VARIANT=7 forces it, 6 keeps tower2b_kernel).  Every workgroup must
    // be resident (partners wait for each other): never more workgroups than CUs.
    const int pairs = (L.B + 1) / 2, grid2s = 16 * ((pairs + 7) / 8);
    const bool split = aligned && L.FP == 128 && L.CP == 256 && L.w2b && L.xbuf && L.xflag && pairs <= L.x_pairs && grid2s <= L.num_cus &&
                       (force == 7 || (!force && L.B >= KAMI_SPLIT_MIN_B));
    const bool fused256 = split || (aligned && L.FP == 128 && ((L.CP == 256 && L.w2b && (force == 6 || (!force && L.B >= 256))) ||
                                                                (L.CP == 128 && L.w4 && L.w4_off[0] != (size_t)-1 && (force == 6 || (!force && L.B >= 256 && !fused)))));
    static const bool dbg = getenv("KAMI_WIDE_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[kami wide] B %d FP %d CP %d aligned %d w2b %d xbuf %d x_pairs %d num_cus %d grid2s %d force %d -> split %d fused256 %d fused %d\n",
                     L.B, L.FP, L.CP, (int)aligned, L.w2b != nullptr, L.xbuf != nullptr, L.x_pairs, L.num_cus, grid2s, force, (int)split, (int)fused256, (int)fused);
    if (!direct && !fused256) hipLaunchKernelGGL(planes_to_act_kernel<T>, dim3(blocks), dim3(256), 0, s, L.in, L.act_in, npix, L.F, L.FP);
    unsigned short *x = L.act[0], *t = L.act[1], *u = L.act[2];
    hipError_t e;
    size_t li = 0;
    auto layer = [&](int idx) { return L.w + L.w_off[idx]; };
    auto layer4 = [&](int idx) -> const unsigned short* { return (L.w4 && L.w4_off[idx] != (size_t)-1) ? L.w4 + L.w4_off[idx] : nullptr; };
    auto shift = [&](int idx) { return L.shift + L.shift_off[idx]; };
    ConvArgs a;
    a.B = L.B;
    if (fused256) {
        static std::atomic<bool> attr_done{ false };
        if (!attr_done.load(std::memory_order_acquire)) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower2b_kernel<T, 256>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower2b_kernel<T, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            attr_done.store(true, std::memory_order_release);
        }
        if (split) {
            static std::atomic<bool> attr2s_done{ false };
            if (!attr2s_done.load(std::memory_order_acquire)) {
                if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower2s_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
                attr2s_done.store(true, std::memory_order_release);
            }
            // every polled word zero before every launch (a block of its own at the allocation's start, a multiple of 16 bytes)
            if ((e = hipMemsetAsync(L.xflag, 0, layers_xflag_bytes(pairs), s)) != hipSuccess) return e;
            Tower2sArgs t2;
            t2.planes = L.in; t2.F = L.F; t2.magic = (unsigned)((0x100000000ull + L.F - 1) / L.F);
            t2.w = L.w2b; t2.shift = shift(0); t2.out = x; t2.B = L.B; t2.R = L.R;
            t2.xbuf = L.xbuf; t2.xflag = L.xflag; t2.flags = L.flags; t2.npairs = pairs;
            static const int abl2s = getenv("KAMI_T2S_ABL") ? atoi(getenv("KAMI_T2S_ABL")) : 0;
            t2.abl = abl2s;
            hipLaunchKernelGGL((tower2s_kernel<T>), dim3(grid2s), dim3(384), 2 * NPIX * (256 * 2 + 16), s, t2);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        } else {
        Tower256Args t2;
        t2.planes = L.in; t2.F = L.F; t2.magic = (unsigned)((0x100000000ull + L.F - 1) / L.F);
        t2.w = L.CP == 256 ? L.w2b : layer4(0); t2.shift = shift(0); t2.out = x; t2.B = L.B; t2.R = L.R;
        if (L.CP == 256) hipLaunchKernelGGL((tower2b_kernel<T, 256>), dim3((L.B + 1) / 2), dim3(256), 2 * NPIX * (256 * 2 + 16), s, t2);
        else hipLaunchKernelGGL((tower2b_kernel<T, 128>), dim3((L.B + 1) / 2), dim3(256), 2 * NPIX * (128 * 2 + 16), s, t2);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        li = 1 + 2 * (size_t)L.R;
    }
    if (fused && !fused256) {
        static std::atomic<bool> attr_done{ false };
        if (!attr_done.load(std::memory_order_acquire)) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower128_kernel<T, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower128_kernel<T, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            attr_done.store(true, std::memory_order_release);
        }
        Tower128Args t8;
        t8.planes = direct ? L.in : nullptr; t8.F = L.F; t8.magic = (unsigned)((0x100000000ull + L.F - 1) / L.F);
        t8.in = L.act_in; t8.w = layer4(0); t8.shift = shift(0); t8.out = x; t8.B = L.B; t8.R = L.R;
        li = 1 + 2 * (size_t)L.R;
        // both heads inside the same launch (KAMI_T128_HEAD=0: policy_head4_kernel behind it, the A/B baseline)
        static const bool head_in = !(getenv("KAMI_T128_HEAD") && atoi(getenv("KAMI_T128_HEAD")) == 0);
        if (head_in && L.wh && L.CP == 128) {
            Head4Args& hd = t8.hd;
            hd.x = nullptr; hd.w = L.wh; hd.shift1 = shift(li); hd.bias2 = shift(li + 1); hd.policy = L.policy; hd.logits = L.want_logits ? L.logits : nullptr;
            hd.flags = L.flags; hd.B = L.B;
            hd.vw = L.vw; hd.vshift = L.vshift; hd.fcw = L.fcw; hd.fcb = L.fcb; hd.fc4 = L.fc4; hd.vfull = L.vfull;
            hipLaunchKernelGGL((tower128_kernel<T, true>), dim3((L.B + 3) / 4), dim3(256), 4 * CHUNKB + 4 * NPIX * (128 * 2 + 16), s, t8);
            return hipGetLastError();
        }
        hipLaunchKernelGGL((tower128_kernel<T, false>), dim3((L.B + 3) / 4), dim3(256), 4 * CHUNKB + 4 * NPIX * (128 * 2 + 16), s, t8);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    // stem                                                                  nn.cpp:62-65
    if (!fused && !fused256) {
    a.in = L.act_in; a.w = layer(li); a.w4 = layer4(li); a.shift = shift(li); a.skip = nullptr; a.out = x; a.Ci = L.FP; a.Co = L.CP; ++li;
    if ((e = launch_conv<T, 9, 0>(a, s)) != hipSuccess) return e;
    }
    for (int r = 0; r < ((fused || fused256) ? 0 : L.R); ++r) {           // nn.cpp:26-34
        a.in = x; a.w = layer(li); a.w4 = layer4(li); a.shift = shift(li); a.skip = nullptr; a.out = t; a.Ci = L.CP; a.Co = L.CP; ++li;
        if ((e = launch_conv<T, 9, 0>(a, s)) != hipSuccess) return e;
        a.in = t; a.w = layer(li); a.w4 = layer4(li); a.shift = shift(li); a.skip = x; a.out = u; ++li;
        if ((e = launch_conv<T, 9, 1>(a, s)) != hipSuccess) return e;
        unsigned short* tmp = x; x = u; u = tmp;
    }
    // policy head                                                           nn.cpp:72-80
    a.w4 = nullptr;
    if (L.wh && (L.CP == 128 || L.CP == 256)) {
        // conv + BN + ReLU, conv + bias and the softmax of four boards per workgroup in one launch
        Head4Args hd;
        hd.x = x; hd.w = L.wh; hd.shift1 = shift(li); hd.bias2 = shift(li + 1); hd.policy = L.policy; hd.logits = L.want_logits ? L.logits : nullptr;
        hd.flags = L.flags; hd.B = L.B;
        hd.vw = L.vw; hd.vshift = L.vshift; hd.fcw = L.fcw; hd.fcb = L.fcb; hd.fc4 = L.fc4; hd.vfull = L.vfull;
        li += 2;
        const int lds = 4 * CHUNKB + 4 * 64 * (128 * 2 + 16) + 4 * 64 * 4;
        static std::atomic<bool> attr_done{ false };
        if (!attr_done.load(std::memory_order_acquire)) {
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_head4_kernel<T, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&policy_head4_kernel<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
            attr_done.store(true, std::memory_order_release);
        }
        if (L.CP == 128) hipLaunchKernelGGL((policy_head4_kernel<T, 1>), dim3((L.B + 3) / 4), dim3(256), lds, s, hd);
        else hipLaunchKernelGGL((policy_head4_kernel<T, 2>), dim3((L.B + 3) / 4), dim3(256), lds, s, hd);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    } else {
    a.in = x; a.w = layer(li); a.shift = shift(li); a.skip = nullptr; a.out = L.pmid; a.Ci = L.CP; a.Co = KH_POLICY_MID; ++li;
    if ((e = launch_conv<T, 1, 0>(a, s)) != hipSuccess) return e;
    a.in = L.pmid; a.w = layer(li); a.shift = shift(li); a.out = L.logits; a.Ci = KH_POLICY_MID; a.Co = 128; ++li;
    if ((e = launch_conv<T, 1, 2>(a, s)) != hipSuccess) return e;
    launch_softmax4672(L.logits, L.policy, L.B, L.flags, s);                  // nn.cpp:80
    // value head                                                            nn.cpp:83-88
    int vb = (int)((npix + 255) / 256);
    if (vb > 4096) vb = 4096;
    hipLaunchKernelGGL(value_conv_kernel<T>, dim3(vb), dim3(256), 0, s, x, L.vw, L.vshift, L.v64, npix, L.CP);
    launch_value_fc(L.v64, L.fcw, L.fcb, L.vfull, L.B, L.flags, s);
    }
    return hipGetLastError();
}

}  // namespace lay

#ifdef KAMI_WIDE_DIAG
extern "C" int kh_debug_t128_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lay::g_t128_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int kh_debug_t2s_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lay::g_t2s_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int kh_debug_wide_stamps(unsigned long long* out, int n)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lay::g_wide_stamps), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
#endif

size_t layers_lds_bytes(int Ci) { return (size_t)lay::LDS_IMG + (size_t)2 * lay::NPIX * (Ci * 2 + 16); }

// sets the kernels' dynamic-LDS attribute outside any stream capture (kh_train records a step as a graph)
hipError_t conv_f32_raw_prepare()
{
    const void* fns[] = {
        reinterpret_cast<const void*>(&lay::conv_f32_kernel<9, 3>), reinterpret_cast<const void*>(&lay::conv_f32_kernel<9, 4>),
        reinterpret_cast<const void*>(&lay::conv_f32_kernel<1, 3>), reinterpret_cast<const void*>(&lay::conv_f32_kernel<1, 4>),
        reinterpret_cast<const void*>(&lay::conv_f32_small_kernel<9, 3>), reinterpret_cast<const void*>(&lay::conv_f32_small_kernel<9, 4>),
        reinterpret_cast<const void*>(&lay::conv_f32_small_kernel<1, 3>), reinterpret_cast<const void*>(&lay::conv_f32_small_kernel<1, 4>),
    };
    for (const void* f : fns) {
        const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// the exact-fp32 MFMA convolution for the trainer (train.hip): out = conv(in) (+ bias), raw or added to `out`
hipError_t launch_conv_f32_raw(const float* in, const float* packed_w, const float* bias, float* out, int B, int Ci, int Co, int taps,
                               bool accumulate, hipStream_t s)
{
    lay::ConvArgsF32 a;
    a.in = in; a.w = packed_w; a.shift = bias; a.skip = nullptr; a.out = out; a.B = B; a.Ci = Ci; a.Co = Co;
    if (taps == 9) return accumulate ? lay::launch_conv_f32<9, 4>(a, s) : lay::launch_conv_f32<9, 3>(a, s);
    return accumulate ? lay::launch_conv_f32<1, 4>(a, s) : lay::launch_conv_f32<1, 3>(a, s);
}

hipError_t launch_layers(int dtype, const LayersArgs& L, hipStream_t s)
{
    if (dtype == KH_F32) return lay::run_f32(L, s);
    return dtype == KH_BF16 ? lay::run<__bf16>(L, s) : lay::run<_Float16>(L, s);
}

}  // namespace kh
