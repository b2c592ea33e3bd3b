// tower_mfma.hip — the throughput path: the WHOLE network forward (kami/nn/nn.cpp:59-91) as ONE
// persistent gfx950 kernel.  bf16 or f16 operands, fp32 accumulation on the matrix cores
// (v_mfma_f32_32x32x16_{bf16,f16}).
//
// Why one kernel: at batch 512 the 6x64 net is only ~35 GFLOP — ~14 us at the MFMA peak — so a
// launch per layer (13 conv layers + heads, ~1.5 us per dependent boundary) would cost more than
// the arithmetic.  Instead one workgroup owns TW_NB boards for the whole forward pass:
//
//   HBM  --fp32 planes-->  LDS S (bf16, 10x12 zero-haloed pixel grid per board)
//   stem 3x3 conv S -> X, then per residual block  X -conv1-> T,  T -conv2(+X)-> X   (all in LDS)
//   heads: value 1x1 + FC + tanh (VALU), policy 1x1 -> P, policy 1x1 -> logits L, softmax -> HBM
//
// Activations never leave the CU.  Each 3x3 conv is an implicit GEMM  D[co][pixel] =
// sum_k W[co][k] * X[k][pixel], k = (tap, ci):  weights are the MFMA A operand, pre-packed on the
// host in exact fragment order; activations are the B operand, read straight from the pixel-major
// LDS image (one ds_read_b128 per lane per 16-channel k-step, tap shifts are immediates).  Output
// channels land 4-consecutive per lane, so the epilogue (ReLU, +skip, convert) writes 8-byte
// packed groups back to LDS.  BatchNorm (eval) is folded: scale into the weights before rounding,
// shift into the accumulator's initial value.
//
// Weights (~1 MB for 6x64, L2 resident) stream through a 6-slot LDS ring of 8 KB chunks by
// LDS-DMA (global_load_lds_dwordx4), one workgroup barrier per chunk, prefetched 5 chunks ahead
// behind a counted s_waitcnt vmcnt; the stream is cyclic over the layers so the prefetch runs
// across layer and board-group boundaries.
//
// LDS bank conflicts: pixel stride = 2*C + 16 bytes and row pitch 12, with the lane->pixel map
// PIXMAP chosen so that every 16-lane group of a ds_read_b128 touches 16 distinct 16-byte slots
// for every tap (MI355X_MICROARCH.md §LDS: groups {0-3,12-15,20-27}, {4-11,16-19,28-31}).
#include "tower_common.h"

#include <atomic>
#include "encode_square.h"

#include <cstdio>
#include <cstdlib>

// Timing-only ablations for tools/tower_ablate.py (diagnostic builds: -DKAMI_TOWER_ABL=<bits>; results are NOT the
// network's, the shipped library is built with 0): 1 no weight DMA inside the steps (the ring keeps its first chunks),
// 2 no per-step barrier either, 4 no policy store, 8 no plane loads, 16 no image write-back at the layer boundaries.
#ifndef KAMI_TOWER_ABL
#define KAMI_TOWER_ABL 0
#endif

// In-kernel phase stamps for tools/tower_stamps.py (diagnostic builds: -DKAMI_TOWER_STAMP=1): s_memtime at ~24 phase
// boundaries per wave, kept in LDS and written over the workgroup's first value row at the end (cdna_hip_programming.md
// §7: read the SHARES of such a build, never its run time).
#ifndef KAMI_TOWER_STAMP
#define KAMI_TOWER_STAMP 0
#endif
#if KAMI_TOWER_STAMP
#define TW_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[wave * 32 + (k)] = t_; } while (0)
#else
#define TW_STAMP(k) do { } while (0)
#endif

namespace kh {


// Ring protocol (D = RING_D slots, chunk c lives in slot c mod D).  Entering step(c) every wave
// holds chunk c's fragments in registers (prefetched during step c-1), chunk c+1 is the oldest
// DMA that may still be in flight and chunks c+2 .. c+D-2 are behind it.  step(c):
//   1. s_waitcnt vmcnt(2*(D-3))  -> this wave's pieces of chunk c+1 have landed
//   2. s_barrier                 -> everybody's have; everybody has finished the MFMAs of chunk c-1
//   3. refill slot (c-1) mod D with chunk c+D-1
// and returns the LDS offset of chunk c+1, which the caller reads into its other register set
// while the MFMAs of chunk c run.  A chunk is therefore requested D-2 steps before it is needed.
// VMX: vector-memory operations younger than the ring's that the wait must leave in flight (the
// stem's first steps run under the tail of the prologue's loads, see the kernel).
template <int VMX = 0>
__device__ __forceinline__ unsigned pipe_step(Pipe& p, int wave, int lane)
{
#if !(KAMI_TOWER_ABL & 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING_D - 3) + VMX) : "memory");
#endif
#if !(KAMI_TOWER_ABL & 2)
    asm volatile("s_barrier" ::: "memory");
#endif
    // Nothing moves across the step boundary: hipcc otherwise hoists the next step's MFMAs up to
    // their operand loads and the register double-buffering collapses into load->wait->MFMA.
    __builtin_amdgcn_sched_barrier(0);
#if !(KAMI_TOWER_ABL & 1)
    pipe_issue(p, wave, lane);
#endif
    p.cslot = (p.cslot + 1 == RING_D) ? 0 : p.cslot + 1;
    return p.ring + p.cslot * CHUNK;
}

struct NoHook { __device__ __forceinline__ void operator()() const {} };

// RELAX / VMX: the first RELAX steps wait with VMX extra operations allowed in flight; hook() runs
// between step HOOK_AT - 1 and step HOOK_AT.
// NREG: the first NREG k-steps take their activation fragments from `breg` (the wave's own packed
// output of the previous layer, see Packed) instead of the LDS image; CF as in b_offset.  With
// NREG > 0 nothing is read from the image before the first step's barrier, which then also orders
// the previous epilogue's image writes of all waves before the other taps' reads.
template <typename T, int TAPS, int KS, int MS, int PAR, int RELAX = 0, int VMX = 0, int HOOK_AT = -1,
          int NREG = 0, bool CF = false, int TAILV = 0, class Hook = NoHook>
__device__ __forceinline__ void gemm_layer(Pipe& p, const char* smem, int wave, int lane,
                                           unsigned b_base, int stride, f32x16 (&acc)[MS],
                                           typename Elem<T>::vec8 (&A)[2][8], const Hook& hook = Hook(),
                                           const typename Elem<T>::vec8* breg = nullptr)
{
    using V = typename Elem<T>::vec8;
    using S = LayerShape<TAPS, KS, MS>;
    constexpr int KPC = S::KPC;
    V B[2][KPC];
#pragma unroll
    for (int k = 0; k < KPC; ++k) {
        if (k < NREG) B[PAR][k] = breg[k];
        else B[PAR][k] = *reinterpret_cast<const V*>(smem + b_base + b_offset<TAPS, KS, CF>(k, stride));
    }
#pragma unroll
    for (int n = 0; n < S::NCH; ++n) {
        const int cur = (PAR + n) & 1, nxt = cur ^ 1;
        if (n == HOOK_AT) hook();
        const unsigned a_off = (n < RELAX ? pipe_step<VMX>(p, wave, lane) : pipe_step<>(p, wave, lane)) + lane * 16;
#pragma unroll
        for (int f = 0; f < 8; ++f) A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
        if (n + 1 < S::NCH) {
#pragma unroll
            for (int k = 0; k < KPC; ++k) {
                if ((n + 1) * KPC + k < NREG) B[nxt][k] = breg[(n + 1) * KPC + k];
                else B[nxt][k] = *reinterpret_cast<const V*>(smem + b_base + b_offset<TAPS, KS, CF>((n + 1) * KPC + k, stride));
            }
        }
        if (TAILV > 0 && n == S::NCH - 1) {
            // last chunk, tile-major: tile 0 is final four MFMAs early and the caller's epilogue of
            // it (TAILV vector ops per MFMA gap) runs under tile 1's MFMAs
#pragma unroll
            for (int ms = 0; ms < MS; ++ms)
#pragma unroll
                for (int k = 0; k < KPC; ++k) acc[ms] = Elem<T>::mfma(A[cur][k * MS + ms], B[cur][k], acc[ms]);
#pragma unroll
            for (int i = 0; i < KPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read (next layer's first weights)
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA tile 0
            }
#pragma unroll
            for (int i = 0; i < (MS - 1) * KPC; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA tile 1..
                if (i > 0) __builtin_amdgcn_sched_group_barrier(0x002, TAILV, 0);   // VALU: tile 0's epilogue
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < KPC; ++k)
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) acc[ms] = Elem<T>::mfma(A[cur][k * MS + ms], B[cur][k], acc[ms]);
        // Pin the interleave: the next chunk's operand reads go out two per MFMA from the top of
        // the step.  Left alone hipcc sinks them to the end of the step and their latency lands on
        // the next barrier; issued as one burst they measured 8 % slower than paced.
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        }
    }
}

// A chunk of zeros in the stream that only flips the register-set parity back to 0.
template <typename T, int PAR>
__device__ __forceinline__ void gemm_dummy(Pipe& p, const char* smem, int wave, int lane,
                                           typename Elem<T>::vec8 (&A)[2][8])
{
    using V = typename Elem<T>::vec8;
    const unsigned a_off = pipe_step(p, wave, lane) + lane * 16;
#pragma unroll
    for (int f = 0; f < 8; ++f) A[PAR ^ 1][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
}

// valuefc + tanh -> [B][256] (nn.cpp:86-88): thread j owns output j, its weight row sits in registers
template <bool LEGAL>
__device__ __forceinline__ void value_fc(const TowerArgs& a, const float4 (&fcw)[16], float fcbias, const float* v64,
                                         int b0, int tid, int lane)
{
    float s[TW_NB];
#pragma unroll
    for (int bb = 0; bb < TW_NB; ++bb) s[bb] = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float4 w = fcw[k];
#pragma unroll
        for (int bb = 0; bb < TW_NB; ++bb) {
            const float4 x = *reinterpret_cast<const float4*>(v64 + bb * 64 + k * 4);
            s[bb] = fmaf(x.x, w.x, s[bb]); s[bb] = fmaf(x.y, w.y, s[bb]);
            s[bb] = fmaf(x.z, w.z, s[bb]); s[bb] = fmaf(x.w, w.w, s[bb]);
        }
    }
    bool nan = false;
#pragma unroll
    for (int bb = 0; bb < TW_NB; ++bb) {
        if (b0 + bb < a.B) {
            const float r = tanhf(s[bb] + fcbias);
            nan |= (r != r);
            a.vfull[(size_t)(b0 + bb) * KH_VALUE_WIDTH + tid] = r;
            if (LEGAL && tid == 0) a.lg_values[b0 + bb] = r;               // column 0: the position's value
        }
    }
    if (__any(nan) && lane == 0) raise_flag<LEGAL>(a, 1);
}


// ---------------------------------------------------------------- the kernel
// KS_STEM = padded input planes / 16 (2 for F <= 32, 8 for F <= 128).  LEGAL: legal-move mode (TowerArgs::lg_*), its own
// instantiation so that the plain kernel's code does not change by a single instruction.
template <typename T, int KS_STEM, bool LEGAL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void tower_kernel(TowerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FP = KS_STEM * 16;
    constexpr int SSTR = FP * 2 + 16;
    constexpr int SBOARD = NPIX * SSTR;
    constexpr int ST_SIZE = st_size(FP);
    constexpr int LDS_PAR = LDS_ST + ST_SIZE;
    constexpr int LDS_L = LDS_X;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5;
    const int R = a.R, F = a.F;

    float* par = reinterpret_cast<float*>(smem + LDS_PAR);
    const float* shift3 = par;                              // [(1 + 2R)][64]
    const float* pshift1 = par + (1 + 2 * R) * TW_CP;       // [128]
    const float* pbias2 = pshift1 + KH_POLICY_MID;          // [128] (73 real)
    const float* vw = pbias2 + 128;                         // [64] valueconv weight * bn scale
    const float* vsh = vw + TW_CP;                          // [4]  folded valueconv/bn shift
    float* v64 = const_cast<float*>(vsh) + 4;               // [TW_NB][64] scratch
    float* red = v64 + TW_NB * 64;                          // [16] reduction scratch
#if KAMI_TOWER_STAMP
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + ((LDS_PAR + tower_par_floats(6) * 4 + 15) & ~15));
    if (lane == 0) stamps[wave * 32 + 30] = __builtin_amdgcn_s_memrealtime();
    TW_STAMP(0);
#endif

    // this wave's 32-pixel column tile: board wave>>1, rows 4*(wave&1)..+3
    const int wb = wave >> 1;
    const int lp = PIXMAP[lane & 31];
    const int py = 4 * (wave & 1) + (lp >> 3), px = lp & 7;
    const unsigned xin = LDS_X + wb * XBOARD + (py * PITCH + px) * XSTR + h * 16;       // tap (0,0) = pixel (y-1,x-1)
    const unsigned tin = LDS_ST + wb * XBOARD + (py * PITCH + px) * XSTR + h * 16;
    const unsigned sin = LDS_ST + wb * SBOARD + (py * PITCH + px) * SSTR + h * 16;
    const unsigned xout = LDS_X + wb * XBOARD + ((py + 1) * PITCH + px + 1) * XSTR;     // own pixel
    const unsigned tout = LDS_ST + wb * XBOARD + ((py + 1) * PITCH + px + 1) * XSTR;

    // Weight stream first: the ring fills while parameters and the first planes are fetched.
    Pipe pipe;
    pipe.stream = a.wstream; pipe.nch = a.nchunks; pipe.next = 0; pipe.islot = 0; pipe.cslot = 0;
    pipe.ring = LDS_RING;
#pragma unroll
    for (int i = 0; i < RING_D - 1; ++i) pipe_issue(pipe, wave, lane);

    // parameter block -> LDS
    for (int i = tid; i < a.npar; i += 256) par[i] = a.params[i];
    TW_STAMP(1);

    using V = typename Elem<T>::vec8;
    V A[2][8];                             // two register sets of weight fragments (current / next chunk)
    constexpr int P1 = LayerShape<9, KS_STEM, 2>::NCH & 1;   // register-set parity after the stem
    constexpr int CH = FP / 8;             // 8-channel (16-byte) chunks per pixel of S
    constexpr int NIT = TW_NB * 64 * CH / 256;              // (board, pixel, chunk) items per thread
    bool first = true;
    float4_u pl[2][4][2];                  // KS_STEM == 8: plane loads in flight, [half][item][+0 / +32 channels]

    const int ngroups = (a.B + TW_NB - 1) / TW_NB;
    for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
        const int b0 = grp * TW_NB;

        // ---- 1'. compact ingest: Env::observe (env.h:202-262) straight into S — one thread per (board,
        //          POV square) builds the 30 channel values from the 80-byte record; values are
        //          0/1/2/4/8, exact in bf16 and f16, so this equals encode -> planes -> convert bit for bit
        if (KS_STEM == 2 && a.boards) {
            const u32x4 z = { 0, 0, 0, 0 };
            for (int i = tid; i < TW_NB * NPIX; i += 256) {
                const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
                if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
                char* dx = smem + LDS_X + (i / NPIX) * XBOARD + pp * XSTR;
#pragma unroll
                for (int k = 0; k < XSTR / 16; ++k) *reinterpret_cast<u32x4*>(dx + k * 16) = z;
                char* ds = smem + LDS_ST + (i / NPIX) * SBOARD + pp * SSTR;
#pragma unroll
                for (int k = 0; k < SSTR / 16; ++k) *reinterpret_cast<u32x4*>(ds + k * 16) = z;
            }
            if (tid < TW_NB * 64) {
                const int bb = tid >> 6, p = tid & 63;
                float v[32];
#pragma unroll
                for (int k = 0; k < 32; ++k) v[k] = 0.0f;
                if (b0 + bb < a.B) {
                    float w[KH_NFEATURES];
                    encode_square(a.boards + (b0 + bb), p, w);
#pragma unroll
                    for (int k = 0; k < KH_NFEATURES; ++k) v[k] = w[k];
                }
                char* dst = smem + LDS_ST + bb * SBOARD + (((p >> 3) + 1) * PITCH + (p & 7) + 1) * SSTR;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    u32x4 o;
                    o.x = pack2<T>(v[8 * c + 0], v[8 * c + 1]); o.y = pack2<T>(v[8 * c + 2], v[8 * c + 3]);
                    o.z = pack2<T>(v[8 * c + 4], v[8 * c + 5]); o.w = pack2<T>(v[8 * c + 6], v[8 * c + 7]);
                    *reinterpret_cast<u32x4*>(dst + c * 16) = o;
                }
            }
        } else
        // ---- 1''. 33..128 planes: the stem runs as two 64-channel passes (the packer splits its
        //          weights the same way) and only the first half of the planes is waited for here; the
        //          second half lands under the first pass's steps and is converted by `ingest_h1`
        //          between two of them.  Item j of a thread = (board, pixel) tid/8 + 32 j, channels
        //          64 hh + 4 (tid%8) .. +3 and +32: every wave load covers 8 pixels x 128 contiguous
        //          bytes.  Loads are clamped into the row / batch and masked afterwards so that every
        //          wave issues exactly 8 loads per half (the counted vmcnt waits depend on it).
        if (KS_STEM == 8) {
            {
#pragma unroll
                for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int bp = (tid >> 3) + 32 * j;
                        const int brd = min(b0 + (bp >> 6), a.B - 1);
#if KAMI_TOWER_ABL & 8
                        const float* row = a.in + (size_t)(brd & 0) * F;      // every load hits the same line: no HBM traffic
#else
                        const float* row = a.in + ((size_t)brd * 64 + (bp & 63)) * F;
#endif
                        const int c = hh * 64 + 4 * (tid & 7);
                        // read-once stream: non-temporal, so that the XCD's L2 keeps the weight stream from one
                        // launch to the next (25 MB of planes and policy pass through 32 MB of L2 per launch)
                        pl[hh][j][0] = __builtin_nontemporal_load(reinterpret_cast<const float4_u*>(row + min(c, F - 4)));
                        pl[hh][j][1] = __builtin_nontemporal_load(reinterpret_cast<const float4_u*>(row + min(c + 32, F - 4)));
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // zero the halo pixels of X and S (the previous group's logits / policy image lived there)
            const u32x4 z = { 0, 0, 0, 0 };
            for (int i = tid; i < TW_NB * NPIX; i += 256) {
                const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
                if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
                char* dx = smem + LDS_X + (i / NPIX) * XBOARD + pp * XSTR;
#pragma unroll
                for (int k = 0; k < XSTR / 16; ++k) *reinterpret_cast<u32x4*>(dx + k * 16) = z;
                char* ds = smem + LDS_ST + (i / NPIX) * SBOARD + pp * SSTR;
#pragma unroll
                for (int k = 0; k < SSTR / 16; ++k) *reinterpret_cast<u32x4*>(ds + k * 16) = z;
            }
            ingest_half<T>(pl[0], 0, smem + LDS_ST, SSTR, SBOARD, b0, tid, lane, a);
        } else
        // ---- 1. planes fp32 [b][64][F] -> T in S (interior pixels, all FP channels); halos zeroed
        {
            // item i = tid + 256*j -> (board, pixel, 8-channel chunk); a wave covers 4 whole pixels
            float vin[NIT][8];
            const int c0 = (tid % CH) * 8;
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int i = tid + 256 * j;
                const int bb = i / (64 * CH), p = (i / CH) & 63;
                const float* src = a.in + ((size_t)(b0 + bb) * 64 + p) * F + c0;
                const bool live = (b0 + bb) < a.B;
                if (live && c0 + 8 <= F) {
                    const float4_u lo = *reinterpret_cast<const float4_u*>(src);
                    const float4_u hi = *reinterpret_cast<const float4_u*>(src + 4);
                    vin[j][0] = lo.x; vin[j][1] = lo.y; vin[j][2] = lo.z; vin[j][3] = lo.w;
                    vin[j][4] = hi.x; vin[j][5] = hi.y; vin[j][6] = hi.z; vin[j][7] = hi.w;
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) vin[j][k] = (live && c0 + k < F) ? src[k] : 0.0f;
                }
            }
            // zero the halo pixels of X and S (the previous group's logits / policy image lived there)
            const u32x4 z = { 0, 0, 0, 0 };
            for (int i = tid; i < TW_NB * NPIX; i += 256) {
                const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
                if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
                char* dx = smem + LDS_X + (i / NPIX) * XBOARD + pp * XSTR;
#pragma unroll
                for (int k = 0; k < XSTR / 16; ++k) *reinterpret_cast<u32x4*>(dx + k * 16) = z;
                char* ds = smem + LDS_ST + (i / NPIX) * SBOARD + pp * SSTR;
#pragma unroll
                for (int k = 0; k < SSTR / 16; ++k) *reinterpret_cast<u32x4*>(ds + k * 16) = z;
            }
            bool bad = false;
#pragma unroll
            for (int j = 0; j < NIT; ++j) {
                const int i = tid + 256 * j;
                const int bb = i / (64 * CH), p = (i / CH) & 63;
                u32x4 o;
                unsigned w[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float lo = vin[j][2 * k], hi = vin[j][2 * k + 1];
                    bad = bad || ((__float_as_uint(lo) & 0x7f800000u) == 0x7f800000u) || ((__float_as_uint(hi) & 0x7f800000u) == 0x7f800000u);
                    w[k] = pack2<T>(lo, hi);
                }
                o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
                *reinterpret_cast<u32x4*>(smem + LDS_ST + bb * SBOARD + (((p >> 3) + 1) * PITCH + (p & 7) + 1) * SSTR + (tid % CH) * 16) = o;
            }
            // a NaN/Inf plane value makes the reference's policy NaN (nn.cpp:176): same verdict here
            if (__any(bad) && lane == 0) raise_flag<LEGAL>(a, 0);
        }
        // vector-memory operations of this wave that are younger than the ring's and may still be in
        // flight past this point: the 8 loads of the second plane half
        constexpr int VMX = KS_STEM == 8 ? 8 : 0;
        TW_STAMP(2);
        if (first) {
            // first group only: chunk 0 of the stream has landed -> first register set
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RING_D - 2) + VMX) : "memory");
            first = false;
            lds_barrier();
#pragma unroll
            for (int f = 0; f < 8; ++f) A[0][f] = *reinterpret_cast<const V*>(smem + LDS_RING + lane * 16 + f * 1024);
        } else {
            lds_barrier();
        }

        // This wave's tile of the residual stream (its own 32 pixels x 64 channels), packed: the skip
        // operand, the centre-tap operand of the next conv and the input of both heads.  The LDS image
        // X carries the same values for the neighbouring pixels' taps.
        Packed<2> xk;
        f32x16 xf[2];          // the same tile in fp32: the skip operand and the value head's input
        TW_STAMP(3);
        // ---- 2. stem: conv1 + batchnorm1 + relu, S -> X                       nn.cpp:62-65
        {
            f32x16 acc[2];
            acc_init<2>(acc, shift3, h);
            if (KS_STEM == 8) {
                // pass 1 (planes 0..63): its first RING_D - 2 steps only need chunks that were requested
                // before the plane loads, so their waits leave the second half's loads in flight
                auto ingest_h1 = [&]() {
                    ingest_half<T>(pl[1], 1, smem + LDS_ST, SSTR, SBOARD, b0, tid, lane, a);
                };
                gemm_layer<T, 9, 4, 2, 0, RING_D - 2, VMX, RING_D - 2>(pipe, smem, wave, lane, sin, SSTR, acc, A, ingest_h1);
                gemm_layer<T, 9, 4, 2, 1>(pipe, smem, wave, lane, sin + 128, SSTR, acc, A);   // pass 2: planes 64..127
            } else {
                gemm_layer<T, 9, KS_STEM, 2, 0>(pipe, smem, wave, lane, sin, SSTR, acc, A);
            }
            TW_STAMP(4);
            epilogue_residual<T, false>(acc, xf, xk);
            store_packed<2>(xk, smem, xout, h);
            lds_barrier();
            // T shares LDS with S: clear T's halo before the tower reads through it
            const u32x4 z = { 0, 0, 0, 0 };
            for (int i = tid; i < TW_NB * NPIX; i += 256) {
                const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
                if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
                char* d = smem + LDS_ST + (i / NPIX) * XBOARD + pp * XSTR;
#pragma unroll
                for (int k = 0; k < XSTR / 16; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
            }
        }

        // ---- 3. residual tower: x = x + relu(bn2(conv2(relu(bn1(conv1 x)))))   nn.cpp:26-34
        // Layer boundaries: the epilogue packs the tile, writes it to the image and the next conv
        // starts on its centre tap from those registers; the first step's ring barrier doubles as the
        // image barrier for the other eight taps.
        TW_STAMP(5);
        f32x16 accn[2];         // next layer's accumulator start (its folded shifts), fetched a layer ahead
        acc_init<2>(accn, shift3 + TW_CP, h);
        for (int r = 0; r < R; ++r) {
            f32x16 acc[2];
            V bf[4];
            packed_fragments<T, 2>(xk, bf);
            acc[0] = accn[0]; acc[1] = accn[1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // own image writes done before the step barrier
            acc_init<2>(accn, shift3 + (2 + 2 * r) * TW_CP, h);
            gemm_layer<T, 9, TW_CP / 16, 2, P1, 0, 0, -1, 4, true, 6>(pipe, smem, wave, lane, xin, XSTR, acc, A, NoHook(), bf);
            TW_STAMP(6 + 2 * r);
            Packed<2> tk;
            epilogue_pack<T, 2>(acc, tk);
#if !(KAMI_TOWER_ABL & 16)
            store_packed<2>(tk, smem, tout, h);
#endif
            packed_fragments<T, 2>(tk, bf);
            acc[0] = accn[0]; acc[1] = accn[1];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            acc_init<2>(accn, shift3 + (3 + 2 * r) * TW_CP, h);         // (past the last block: the policy shifts, unused)
            gemm_layer<T, 9, TW_CP / 16, 2, P1 ^ 1, 0, 0, -1, 4, true, 11>(pipe, smem, wave, lane, tin, XSTR, acc, A, NoHook(), bf);
            TW_STAMP(7 + 2 * r);
            epilogue_residual<T, true>(acc, xf, xk);
#if !(KAMI_TOWER_ABL & 16)
            store_packed<2>(xk, smem, xout, h);
#endif
        }

        // ---- 4a. value head, first half: valueconv + vbatchnorm + relu (nn.cpp:83-85) on the fp32
        //          tile: each lane holds 32 of its pixel's 64 channels, its partner lane (+-32) the rest.
        //          Four independent partial sums (a single fma chain is latency-bound with one wave per
        //          SIMD), halves joined by one v_permlane32_swap.  A NaN or Inf anywhere in the residual
        //          stream makes this lane's partial sum non-finite (x*w, w finite): that is the poisoned-
        //          stream detector for the ReLUs' NaN squashing (see relu_nan).
        {
            f32x2 s01 = { 0.0f, 0.0f }, s23 = { 0.0f, 0.0f };      // register pairs: v_pk_fma_f32 on consecutive registers
#pragma unroll
            for (int ms = 0; ms < 2; ++ms)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    using f32x4 = float __attribute__((ext_vector_type(4)));
                    const f32x4 w = *reinterpret_cast<const f32x4*>(vw + ms * 32 + 8 * g + 4 * h);
                    const f32x2 x01 = { xf[ms][4 * g + 0], xf[ms][4 * g + 1] }, x23 = { xf[ms][4 * g + 2], xf[ms][4 * g + 3] };
                    s01 = x01 * w.xy + s01;
                    s23 = x23 * w.zw + s23;
                }
            const float part = (s01.x + s01.y) + (s23.x + s23.y);
            const unsigned pu = __float_as_uint(part);
            if ((pu & 0x7f800000u) == 0x7f800000u) raise_flag<LEGAL>(a, 0);
            const auto sw = __builtin_amdgcn_permlane32_swap(pu, pu, false, false);   // {lanes 0-31 twice, lanes 32-63 twice}
            const float sv = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            if (h == 0) v64[wb * 64 + py * 8 + px] = relu_nan(sv + vsh[0]);
        }

        // valuefc row of this thread (output j = tid): requested here, used after the softmax.  17 loads
        // younger than the ring's: the next RING_D - 2 steps (two of 4b, two of 4c) leave them in flight.
        TW_STAMP(18);
        float4 fcw[16];
#pragma unroll
        for (int k = 0; k < 16; ++k)      // scalar base + one lane offset: no per-load address registers
            fcw[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.fcw4) + (size_t)k * KH_VALUE_WIDTH * 16 + (unsigned)tid * 16u);
        const float fcbias = a.fcb[tid];
        __builtin_amdgcn_sched_barrier(0);

        // ---- 4b. policy head: policyconv + pbatchnorm + relu (nn.cpp:72-74), 1x1: the operand is the
        //          wave's own tile, and so is policyconv2's — neither touches an LDS image
        Packed<4> pk;
        {
            f32x16 acc[4];
            V bf[4];
            packed_fragments<T, 2>(xk, bf);
            acc_init<4>(acc, pshift1, h);
            gemm_layer<T, 1, TW_CP / 16, 4, P1, 2, 17, -1, 4>(pipe, smem, wave, lane, 0, 0, acc, A, NoHook(), bf);
            TW_STAMP(19);
            epilogue_pack<T, 4>(acc, pk);
        }
        // ---- 4c. policyconv2 (+bias): -> logits L[board][pixel*73 + plane]      nn.cpp:75-79
        {
            f32x16 acc[4];                  // 73 planes padded to 128 rows: whole 2-k-step chunks
            V bf[8];
            packed_fragments<T, 4>(pk, bf);
            acc_init<4>(acc, pbias2, h);
            gemm_layer<T, 1, KH_POLICY_MID / 16, 4, P1, RING_D - 4, 17, -1, 8>(pipe, smem, wave, lane, 0, 0, acc, A, NoHook(), bf);
            if (P1) gemm_dummy<T, 1>(pipe, smem, wave, lane, A);      // stream parity back to 0 for the next group
            float* lrow = reinterpret_cast<float*>(smem + LDS_L + wb * LBOARD) + (py * 8 + px) * KH_POLICY_PLANES;
            {    // raw logits to LDS, softmax below
#pragma unroll
                for (int ms = 0; ms < 3; ++ms)      // planes >= 96 are padding
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int plane = ms * 32 + 8 * g + 4 * h + i;
                            if (plane < KH_POLICY_PLANES) lrow[plane] = acc[ms][4 * g + i];
                        }
                lds_barrier();
            }
        }
        TW_STAMP(20);

        // ---- 4d. softmax over all 4672 logits of a board (nn.cpp:80): 128 threads per board, one LDS
        //          pass, each thread keeps its <= 10 float4 in registers
        {
            const int bb = tid >> 7, tt = tid & 127;
            const bool live = (b0 + bb) < a.B;
            const float4* L4 = reinterpret_cast<const float4*>(smem + LDS_L + bb * LBOARD);
            constexpr int NQ = KH_PSIZE / 4;               // 1168 float4 = 9 * 128 + 16
            float4 v[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                const int q = tt + 128 * k;
                v[k] = (q < NQ) ? L4[q] : make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
            }
            if (a.logits && live) {
                float4* lo = reinterpret_cast<float4*>(a.logits + (size_t)(b0 + bb) * KH_PSIZE);
#pragma unroll
                for (int k = 0; k < 10; ++k)
                    if (tt + 128 * k < NQ) lo[tt + 128 * k] = v[k];
            }
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < 10; ++k) m = fmaxf(fmaxf(m, fmaxf(v[k].x, v[k].y)), fmaxf(v[k].z, v[k].w));
            m = wave_max_f(m);
            if (lane == 0) red[wave] = m;
            lds_barrier();
            m = fmaxf(red[bb * 2], red[bb * 2 + 1]);
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 10; ++k) {
                v[k].x = __expf(v[k].x - m); v[k].y = __expf(v[k].y - m);
                v[k].z = __expf(v[k].z - m); v[k].w = __expf(v[k].w - m);
                s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            }
            s = wave_sum_f(s);
            if (lane == 0) red[4 + wave] = s;
            lds_barrier();
            const float inv = 1.0f / (red[4 + bb * 2] + red[4 + bb * 2 + 1]);
            bool nan = false;
            if (LEGAL) {
                // legal-move mode: the board's first wave does what gather_legal_kernel does with the stored row — p[a] =
                // exp(l[a] - m) * inv recomputed from the logits in LDS (the same operations on the same values: the same
                // bits), summed and renormalised in that kernel's order
                nan = inv != inv;                           // a NaN logit makes the sum, hence every entry, NaN
                if (live && (wave & 1) == 0) {
                    const float* Lb = reinterpret_cast<const float*>(smem + LDS_L + bb * LBOARD);
                    const int lo = a.lg_offsets[b0 + bb], hi = a.lg_offsets[b0 + bb + 1];
                    float sum = 0.0f;
                    // (__fmul_rn: the row's entries are rounded products; no contraction into the sum)
                    for (int k = lo + lane; k < hi; k += 64) {
                        const int ac = a.lg_actions[k];
                        sum += (ac >= 0 && ac < KH_PSIZE) ? __fmul_rn(__expf(Lb[ac] - m), inv) : 0.0f;
                    }
                    sum = wave_sum_f(sum);
                    const float rn = sum > 0.0f ? 1.0f / sum : 0.0f;
                    for (int k = lo + lane; k < hi; k += 64) {
                        const int ac = a.lg_actions[k];
                        a.lg_priors[k] = (ac >= 0 && ac < KH_PSIZE) ? __fmul_rn(__fmul_rn(__expf(Lb[ac] - m), inv), rn) : 0.0f;
                    }
                }
            } else if (live) {
                float4* po = reinterpret_cast<float4*>(a.policy + (size_t)(b0 + bb) * KH_PSIZE);
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    float4 o;
                    o.x = v[k].x * inv; o.y = v[k].y * inv; o.z = v[k].z * inv; o.w = v[k].w * inv;
                    nan |= (o.x != o.x) | (o.y != o.y) | (o.z != o.z) | (o.w != o.w);
                    if (tt + 128 * k < NQ) {
                        using f4 = float __attribute__((ext_vector_type(4)));
                        const f4 ov = { o.x, o.y, o.z, o.w };
#if !(KAMI_TOWER_ABL & 4)
                        __builtin_nontemporal_store(ov, reinterpret_cast<f4*>(po) + tt + 128 * k);
#endif
                    }
                }
            }
            if (__any(nan) && lane == 0) raise_flag<LEGAL>(a, 0);
        }

        TW_STAMP(21);
        // ---- 4e. value head, second half: valuefc + tanh -> [B][256]            nn.cpp:86-88
        value_fc<LEGAL>(a, fcw, fcbias, v64, b0, tid, lane);
        lds_barrier();      // L / v64 are dead; the next group may overwrite them
        TW_STAMP(22);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the prefetch ring before exit
#if KAMI_TOWER_STAMP
    TW_STAMP(23);
    if (lane == 0) stamps[wave * 32 + 31] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane < 32 && (int)blockIdx.x * TW_NB < a.B)      // over the workgroup's first value row (256 floats = 4 waves x 32 stamps)
        reinterpret_cast<unsigned long long*>(a.vfull + (size_t)blockIdx.x * TW_NB * KH_VALUE_WIDTH)[wave * 32 + lane] = stamps[wave * 32 + lane];
#endif
}

template <typename T, int KS_STEM, bool LEGAL = false> static hipError_t launch(const TowerArgs& a, int grid, hipStream_t s)
{
    constexpr int FP = KS_STEM * 16;
    const int lds = LDS_ST + st_size(FP) + tower_par_floats(a.R) * 4 + (KAMI_TOWER_STAMP ? 16 + 4 * 32 * 8 : 0);
    static std::atomic<bool> attr_done{ false };      // engines are called from many host threads; setting it twice is harmless
    if (!attr_done.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_kernel<T, KS_STEM, LEGAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((tower_kernel<T, KS_STEM, LEGAL>), dim3(grid), dim3(256), lds, s, a);
    return hipGetLastError();
}

int tower_lds_bytes(int FP, int R)
{
    return LDS_ST + st_size(FP) + tower_par_floats(R) * 4;
}

#ifndef KAMI_TOWER_DEFAULT_V
#define KAMI_TOWER_DEFAULT_V 8
#endif

int tower_variant()
{
    static const int variant = [] { const char* v = getenv("KAMI_TOWER_V"); return v && *v && atoi(v) == 4 ? 4 : (v && *v && atoi(v) == 8 ? 8 : KAMI_TOWER_DEFAULT_V); }();
    return variant;
}

hipError_t launch_tower(int dtype, int FP, const TowerArgs& a, int num_cus, hipStream_t s)
{
    if (tower_variant() == 8) return launch_tower8(dtype, FP, a, num_cus, s);
    const int ngroups = (a.B + TW_NB - 1) / TW_NB;
    const int grid = ngroups < num_cus ? ngroups : num_cus;      // one workgroup per CU (LDS-bound residency)
    if (a.lg_offsets) {                      // legal-move mode: compact records (F <= 32) only
        if (FP != 32 || !a.lg_actions || !a.lg_priors || !a.lg_values || !a.lg_flags) return hipErrorInvalidValue;
        return dtype == KH_BF16 ? launch<__bf16, 2, true>(a, grid, s) : launch<_Float16, 2, true>(a, grid, s);
    }
    if (dtype == KH_BF16) return FP == 32 ? launch<__bf16, 2>(a, grid, s) : launch<__bf16, 8>(a, grid, s);
    return FP == 32 ? launch<_Float16, 2>(a, grid, s) : launch<_Float16, 8>(a, grid, s);
}

}  // namespace kh
