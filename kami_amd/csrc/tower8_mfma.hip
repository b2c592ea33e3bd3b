// tower8_mfma.hip — the whole network forward (kami/nn/nn.cpp:59-91) as one persistent gfx950 kernel with
// SPECIALISED waves: 512 threads per workgroup = 4 compute waves (one per SIMD: the MFMAs, the epilogues, the
// softmax) + 4 helper waves (one per SIMD beside them: the weight stream's LDS-DMA, the plane ingest, the value FC).
//
// Why: tower_mfma.hip's four waves do everything themselves, and with ONE wave per SIMD every instruction that is
// not an MFMA and does not fit a 24-cycle MFMA gap stops the matrix pipe.  Measured on that kernel (same-process A/B,
// profiles/r03_tower_ablations_v1.txt; phase stamps, profiles/r03_tower_stamps_v1.txt): the two LDS-DMA pieces per
// step cost 3.8 us of 36, the plane ingest keeps the matrix pipe idle for the first 9 500 cycles of 72 400, the heads'
// VALU work and the value FC for most of the last 11 700.  A helper wave's VMEM / VALU / LDS instructions issue
// beside the compute wave's MFMAs (different instruction types of different waves issue in the same cycle), so here
// the compute waves' stream is: barrier, operand reads, MFMAs, epilogue.
//
// Same data layout, same weight stream, same arithmetic in the same order as tower_mfma.hip (geometry and helpers:
// tower_common.h): the convolutions' results are bit-identical to that kernel's (tools/tower_ablate.py).
//
// Barriers: s_barrier counts all 8 waves, so both roles execute exactly the same number of them per board group —
// every barrier below is tagged [B..] in both paths.
#include "tower_common.h"

#include <atomic>
#include <cstdlib>

#ifndef KAMI_TOWER_STAMP
#define KAMI_TOWER_STAMP 0
#endif
#if KAMI_TOWER_STAMP
#define T8_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) stamps[wave * 32 + (k)] = t_; } while (0)
#else
#define T8_STAMP(k) do { } while (0)
#endif

// Timing-only ablations (diagnostic builds, -DKAMI_TOWER8_ABL=<bits>; results are NOT the network's): 1 no LDS-DMA in the
// steps (the ring keeps its first chunks), 2 no activation reads in the regular steps, 4 no weight reads in them.
#ifndef KAMI_TOWER8_ABL
#define KAMI_TOWER8_ABL 0
#endif

namespace kh {

// Output stores are WRITE-THROUGH (sc1): the policy rows are 9.6 MB per batch-512 launch, written in the kernel's last
// phase; with plain or non-temporal stores they sit dirty in the XCDs' L2s when the kernel ends and the end-of-kernel
// release writes them back before the next launch may start — 1.7 us of every 32 (same-device A/B of the store's cache
// policy: nt 31.92, plain 31.96, sc1 30.22, sc0 sc1 30.21, sc1 nt 30.32 us; profiles/r03_tower8_store_policy_ab.txt).
using f32x4_t = float __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store_wt(f32x4_t* p, f32x4_t v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void store_wt(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }

// ---------------------------------------------------------------- ring protocol, two roles
// Chunk c lives in slot c mod RING_D.  Step c of a compute wave multiplies chunk c (in registers since step c - 1)
// and reads chunk c + 1 into its other register set.  Before barrier c a helper wave has waited for its two pieces of
// chunk c + 2 (T8_LAG = 1: one chunk more than the reads that follow barrier c need), so a compute wave may read
// chunk c + 1 BEFORE barrier c — the first step of a layer starts on operands it already holds and meets the barrier
// in its middle (see boundary()).  After barrier c the helper refills slot (c - 1) mod RING_D — whose chunk every
// compute wave finished reading two steps ago — with chunk c + RING_D - 1: requested RING_D - 3 steps before its
// barrier.  Barrier c may sit anywhere between the reads of chunk c (step c - 1) and those of chunk c + 2 (step c + 1).
constexpr int T8_LAG = 1;
// (Tried and dropped: the stream waves bringing the weights global -> registers -> ds_write_b128 instead of LDS-DMA — six
//  chunks deep in registers, compiler-counted waits: bit-identical, 35.4 us against the LDS-DMA's 32.5 on the same device,
//  profiles/r03_tower8_wreg_ab.txt.  Two waves' wide LDS stores run at half the store path's rate.)
struct CPipe { unsigned ring; int cslot; };

__device__ __forceinline__ unsigned cpipe_advance(CPipe& p)      // next chunk's LDS offset, no barrier
{
    p.cslot = (p.cslot + 1 == RING_D) ? 0 : p.cslot + 1;
    return p.ring + p.cslot * CHUNK;
}

__device__ __forceinline__ unsigned cpipe_step(CPipe& p)
{
    asm volatile("s_barrier" ::: "memory");
    // nothing moves across the step boundary (see tower_mfma.hip: the register double buffer would collapse)
    __builtin_amdgcn_sched_barrier(0);
    p.cslot = (p.cslot + 1 == RING_D) ? 0 : p.cslot + 1;
    return p.ring + p.cslot * CHUNK;
}

// A stream wave's half of a chunk: four 1 KB LDS-DMA pieces (tower_common.h's pipe_issue moves a quarter).
__device__ __forceinline__ void pipe_issue4(Pipe& p, int hw, int lane)
{
    const char* sbase = p.stream + (size_t)p.next * CHUNK + hw * 4096;      // wave-uniform -> SGPR pair
    const unsigned dst = p.ring + p.islot * CHUNK + hw * 4096;
    const unsigned voff = lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
    p.next = (p.next + 1 == p.nch) ? 0 : p.next + 1;
    p.islot = (p.islot + 1 == RING_D) ? 0 : p.islot + 1;
}

// VMX: vector-memory operations of this stream wave that are YOUNGER than the ring's and may stay in flight
template <int VMX>
__device__ __forceinline__ void hpipe_step(Pipe& p, int hw, int lane)
{
#if KAMI_TOWER8_ABL & 1
    asm volatile("s_barrier" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (RING_D - 3 - T8_LAG) + VMX) : "memory");
    asm volatile("s_barrier" ::: "memory");
    pipe_issue4(p, hw, lane);
#endif
}

// ---------------------------------------------------------------- implicit-GEMM layer, compute-wave side
// tower_mfma.hip's gemm_layer without the stream's issue side (template parameters as there).
template <typename T, int TAPS, int KS, int MS, int PAR, int NREG = 0, bool CF = false, int TAILV = 0>
__device__ __forceinline__ void gemm8_layer(CPipe& p, const char* smem, int lane, unsigned b_base, int stride,
                                            f32x16 (&acc)[MS], typename Elem<T>::vec8 (&A)[2][8],
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
        const unsigned a_off = cpipe_step(p) + lane * 16;
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
            // last chunk, tile-major: tile 0 is final four MFMAs early, its epilogue runs under tile 1's MFMAs
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
        // the next chunk's operand reads go out two per MFMA from the top of the step
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        }
    }
}

// A chunk of zeros in the stream that only flips the register-set parity back to 0.
template <typename T, int PAR>
__device__ __forceinline__ void gemm8_dummy(CPipe& p, const char* smem, int lane, typename Elem<T>::vec8 (&A)[2][8])
{
    using V = typename Elem<T>::vec8;
    const unsigned a_off = cpipe_step(p) + lane * 16;
#pragma unroll
    for (int f = 0; f < 8; ++f) A[PAR ^ 1][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
}


// ---------------------------------------------------------------- steps N0 .. N1-1 of a layer, operands carried by the caller
// gemm8_layer's loop body with the register sets owned by the caller, so that a layer can be cut at its first and last
// chunk (boundary() below).  Step n multiplies chunk n (A[cur], B[cur], cur = (PAR + n) & 1) and reads chunk n + 1.
// BARMID: step N0's barrier sits in the MIDDLE of the step (after four MFMAs) instead of at its top — used for the step
// that follows a boundary(), whose own barrier comes late in the layer's first chunk.
// k-step -> byte offset of its activation fragment from b_base.  BM_RASTER / BM_CENTRE: tower_common.h's b_offset without /
// with the centre tap first; BM_HALVES / BM_QUARTERS: the 33..128-plane stem walked as two passes of 64 channels
// (tower_mfma.hip's order) / four passes of 32 (this kernel's: the planes arrive and are converted quarter by quarter),
// the order the packer lays the stream's weights out in.
enum { BM_RASTER = 0, BM_CENTRE = 1, BM_HALVES = 2, BM_QUARTERS = 3 };
template <int TAPS, int KS, int BM>
__device__ __forceinline__ constexpr unsigned b_off8(int kk, int stride)
{
    if constexpr (BM == BM_HALVES) {
        const int pass = kk / (TAPS * KS / 2), k = kk % (TAPS * KS / 2);
        return b_offset<TAPS, KS / 2, false>(k, stride) + pass * (KS / 2) * 32;
    }
    if constexpr (BM == BM_QUARTERS) {
        const int pass = kk / (TAPS * KS / 4), k = kk % (TAPS * KS / 4);
        return b_offset<TAPS, KS / 4, false>(k, stride) + pass * (KS / 4) * 32;
    }
    return b_offset<TAPS, KS, BM == BM_CENTRE>(kk, stride);
}

template <typename T, int TAPS, int KS, int MS, int PAR, int NREG, int BM, int N0, int N1, bool BARMID = false>
__device__ __forceinline__ void gemm8_steps(CPipe& p, const char* smem, int lane, unsigned b_base, int stride,
                                            f32x16 (&acc)[MS], typename Elem<T>::vec8 (&A)[2][8],
                                            typename Elem<T>::vec8 (&B)[2][8 / MS],
                                            const typename Elem<T>::vec8* breg = nullptr)
{
    using V = typename Elem<T>::vec8;
    using S = LayerShape<TAPS, KS, MS>;
    constexpr int KPC = S::KPC;
#pragma unroll
    for (int n = N0; n < N1; ++n) {
        const int cur = (PAR + n) & 1, nxt = cur ^ 1;
        if (BARMID && n == N0) {
            static_assert(!BARMID || (MS == 2 && KPC == 4), "mid-step barrier: 2 x 4 chunk");
            // chunk n + 1's weights landed before the PREVIOUS barrier (T8_LAG): read them first, meet the barrier
            // after four MFMAs, read the activations behind it
            const unsigned a_off = cpipe_advance(p) + lane * 16;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) {
                    const int i = 2 * k + ms;
                    A[nxt][2 * i] = *reinterpret_cast<const V*>(smem + a_off + (2 * i) * 1024);
                    A[nxt][2 * i + 1] = *reinterpret_cast<const V*>(smem + a_off + (2 * i + 1) * 1024);
                    acc[ms] = Elem<T>::mfma(A[cur][k * MS + ms], B[cur][k], acc[ms]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_barrier" ::: "memory");
#pragma unroll
            for (int k = 2; k < 4; ++k) {
#pragma unroll
                for (int ms = 0; ms < 2; ++ms) {
                    if (k == 2 && n + 1 < S::NCH) {
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int kk = 2 * ms + j;
                            if ((n + 1) * KPC + kk < NREG) B[nxt][kk] = breg[(n + 1) * KPC + kk];
                            else B[nxt][kk] = *reinterpret_cast<const V*>(smem + b_base + b_off8<TAPS, KS, BM>((n + 1) * KPC + kk, stride));
                        }
                    }
                    acc[ms] = Elem<T>::mfma(A[cur][k * MS + ms], B[cur][k], acc[ms]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            continue;
        }
        const unsigned a_off = cpipe_step(p) + lane * 16;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            if ((KAMI_TOWER8_ABL & 4) && f > 0) A[nxt][f] = A[nxt][0];
            else A[nxt][f] = *reinterpret_cast<const V*>(smem + a_off + f * 1024);
        }
        if (n + 1 < S::NCH) {
#pragma unroll
            for (int k = 0; k < KPC; ++k) {
                if ((n + 1) * KPC + k < NREG) B[nxt][k] = breg[(n + 1) * KPC + k];
                else if ((KAMI_TOWER8_ABL & 2) && k > 0) B[nxt][k] = B[nxt][0];
                else B[nxt][k] = *reinterpret_cast<const V*>(smem + b_base + b_off8<TAPS, KS, BM>((n + 1) * KPC + k, stride));
            }
        }
#pragma unroll
        for (int k = 0; k < KPC; ++k)
#pragma unroll
            for (int ms = 0; ms < MS; ++ms) acc[ms] = Elem<T>::mfma(A[cur][k * MS + ms], B[cur][k], acc[ms]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // DS read
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // MFMA
        }
    }
}

// ---------------------------------------------------------------- layer boundary
// What ends layer L (a 2 x 4 chunk: stem or 3x3) and starts the 3x3 layer L + 1, as ONE scheduled sequence:
//   L's last chunk tile-major (tile 0's sums final four MFMAs early) | L + 1's first chunk — the centre tap, whose
//   operand is this wave's own output, straight from the epilogue's registers — with L's epilogue spread over the
//   gaps of both, the image write-back for the neighbours' taps behind it, and L + 1's barrier only before the first
//   reads of that image (chunk 1's activations).
// The matrix pipe therefore never waits for "epilogue -> LDS -> barrier -> read": measured on tower8 v1 that wait
// was ~435 cycles per conv1 -> conv2 boundary and ~775 per conv2 -> conv1 (profiles/r03_tower8_v1_ab_stamps.txt).
// Every micro-block below is fenced (sched_barrier): source order IS issue order.
//   EPI_PACK   t = relu(acc), rounded to T first   (nn.cpp:30)
//   EPI_RESID  x = x + relu(acc)                   (nn.cpp:31; with x = 0 also the stem's relu, nn.cpp:65)
// Same operations on the same values as epilogue_pack / epilogue_residual: the same bits.
enum { EPI_STEM = 0, EPI_PACK = 1, EPI_RESID = 2 };

template <typename T, int EPI, int G>
__device__ __forceinline__ u32x2 epi_group(const f32x16& acc, f32x16& xf)
{
    u32x2 o;
    if (EPI == EPI_PACK) {
        o.x = relu_pk(pack2<T>(acc[4 * G + 0], acc[4 * G + 1]));
        o.y = relu_pk(pack2<T>(acc[4 * G + 2], acc[4 * G + 3]));
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = relu_nan(acc[4 * G + i]);
            xf[4 * G + i] = EPI == EPI_RESID ? xf[4 * G + i] + t : t;
        }
        o.x = pack2<T>(xf[4 * G + 0], xf[4 * G + 1]);
        o.y = pack2<T>(xf[4 * G + 2], xf[4 * G + 3]);
    }
    return o;
}

#define T8_FENCE() __builtin_amdgcn_sched_barrier(0)

// CURP: register set that holds L's last chunk (A[CURP], B[CURP]); on exit A[CURP] / B[CURP] hold L + 1's chunk 1 and
// nacc L + 1's sums after its chunk 0.  out_pix: LDS offset of the lane's own pixel (+ 8 h: its channel group) in the
// image L + 1 reads; nb_base: L + 1's b_base; nshift: LDS offset of L + 1's folded shifts (+ 16 h).  All three are
// per-lane bases that every access below extends by an immediate only.
template <typename T, int EPI, int CURP>
__device__ __forceinline__ void boundary(CPipe& p, char* smem, int lane, f32x16 (&acc)[2], f32x16 (&nacc)[2],
                                         unsigned nshift, typename Elem<T>::vec8 (&A)[2][8],
                                         typename Elem<T>::vec8 (&B)[2][4], f32x16 (&xf)[2], unsigned out_pix, unsigned nb_base)
{
    using V = typename Elem<T>::vec8;
    constexpr int cur = CURP, nxt = CURP ^ 1;
    u32x2 o[2][4];
#define T8_LDN(ms, g) do { const float4 s_ = *reinterpret_cast<const float4*>(smem + nshift + ((ms) * 32 + 8 * (g)) * 4); \
        nacc[ms][4 * (g) + 0] = s_.x; nacc[ms][4 * (g) + 1] = s_.y; nacc[ms][4 * (g) + 2] = s_.z; nacc[ms][4 * (g) + 3] = s_.w; } while (0)
#define T8_ST(t, g) (*reinterpret_cast<u32x2*>(smem + out_pix + ((t) * 32 + 8 * (g)) * 2) = o[t][g])
#define T8_FRAG(ks) __builtin_bit_cast(V, (u32x4){ o[(ks) >> 1][2 * ((ks) & 1)].x, o[(ks) >> 1][2 * ((ks) & 1)].y, \
                                                   o[(ks) >> 1][2 * ((ks) & 1) + 1].x, o[(ks) >> 1][2 * ((ks) & 1) + 1].y })
    const unsigned a_off = cpipe_step(p) + lane * 16;                                   // [barrier: L's last step]
#define T8_RDA(set, f, off) (A[set][f] = *reinterpret_cast<const V*>(smem + (off) + (f) * 1024))
    // ---- L's last chunk, tile 0; L + 1's shifts become its accumulators' initial values
    T8_LDN(0, 0); T8_LDN(0, 1); acc[0] = Elem<T>::mfma(A[cur][0], B[cur][0], acc[0]); T8_FENCE();
    T8_LDN(0, 2); T8_LDN(0, 3); acc[0] = Elem<T>::mfma(A[cur][2], B[cur][1], acc[0]); T8_FENCE();
    T8_LDN(1, 0); T8_LDN(1, 1); acc[0] = Elem<T>::mfma(A[cur][4], B[cur][2], acc[0]); T8_FENCE();
    T8_LDN(1, 2); T8_LDN(1, 3); acc[0] = Elem<T>::mfma(A[cur][6], B[cur][3], acc[0]); T8_FENCE();
    // ---- tile 1; L + 1's first weights; tile 0's epilogue once its sums are final
    T8_RDA(nxt, 0, a_off); T8_RDA(nxt, 1, a_off); acc[1] = Elem<T>::mfma(A[cur][1], B[cur][0], acc[1]); T8_FENCE();
    T8_RDA(nxt, 2, a_off); T8_RDA(nxt, 3, a_off); acc[1] = Elem<T>::mfma(A[cur][3], B[cur][1], acc[1]); T8_FENCE();
    T8_RDA(nxt, 4, a_off); T8_RDA(nxt, 5, a_off);
    o[0][0] = epi_group<T, EPI, 0>(acc[0], xf[0]); o[0][1] = epi_group<T, EPI, 1>(acc[0], xf[0]);
    acc[1] = Elem<T>::mfma(A[cur][5], B[cur][2], acc[1]); T8_FENCE();
    T8_RDA(nxt, 6, a_off); T8_RDA(nxt, 7, a_off);
    o[0][2] = epi_group<T, EPI, 2>(acc[0], xf[0]); o[0][3] = epi_group<T, EPI, 3>(acc[0], xf[0]);
    acc[1] = Elem<T>::mfma(A[cur][7], B[cur][3], acc[1]); T8_FENCE();
    // ---- L + 1, chunk 0: centre tap, k-steps 0..3 = the wave's own channels 0..15, .., 48..63
    const unsigned a2 = cpipe_advance(p) + lane * 16;             // chunk 1's weights: landed before the barrier above
    {
        const V f0 = T8_FRAG(0);
        T8_ST(0, 0); T8_ST(0, 1); T8_RDA(cur, 0, a2);
        nacc[0] = Elem<T>::mfma(A[nxt][0], f0, nacc[0]); T8_FENCE();
        T8_ST(0, 2); T8_ST(0, 3); T8_RDA(cur, 1, a2);
        nacc[1] = Elem<T>::mfma(A[nxt][1], f0, nacc[1]); T8_FENCE();
    }
    {
        const V f1 = T8_FRAG(1);
        T8_RDA(cur, 2, a2); T8_RDA(cur, 3, a2);
        nacc[0] = Elem<T>::mfma(A[nxt][2], f1, nacc[0]); T8_FENCE();
        T8_RDA(cur, 4, a2); T8_RDA(cur, 5, a2);
        o[1][0] = epi_group<T, EPI, 0>(acc[1], xf[1]); o[1][1] = epi_group<T, EPI, 1>(acc[1], xf[1]);
        nacc[1] = Elem<T>::mfma(A[nxt][3], f1, nacc[1]); T8_FENCE();
    }
    {
        const V f2 = T8_FRAG(2);
        o[1][2] = epi_group<T, EPI, 2>(acc[1], xf[1]); o[1][3] = epi_group<T, EPI, 3>(acc[1], xf[1]);
        T8_ST(1, 0); T8_ST(1, 1); T8_ST(1, 2); T8_ST(1, 3);
        nacc[0] = Elem<T>::mfma(A[nxt][4], f2, nacc[0]); T8_FENCE();
        T8_RDA(cur, 6, a2); T8_RDA(cur, 7, a2);
        nacc[1] = Elem<T>::mfma(A[nxt][5], f2, nacc[1]); T8_FENCE();
    }
    {
        const V f3 = T8_FRAG(3);
        // the image writes are older than the two reads above: done once at most two LDS operations are outstanding;
        // then every wave's are, and chunk 1's activations (the window's first tap) may be read      [barrier: L + 1's first step]
        asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int k = 0; k < 4; ++k) B[cur][k] = *reinterpret_cast<const V*>(smem + nb_base + b_offset<9, TW_CP / 16, true>(4 + k, XSTR));
        nacc[0] = Elem<T>::mfma(A[nxt][6], f3, nacc[0]); T8_FENCE();
        nacc[1] = Elem<T>::mfma(A[nxt][7], f3, nacc[1]); T8_FENCE();
    }
#undef T8_LDN
#undef T8_ST
}

// The last 3x3 layer's (or, without blocks, the stem's) end in front of the heads: last chunk tile-major, the whole
// epilogue, no image (both heads are 1x1 and read the wave's own tile).  A[CURP ^ 1] receives policyconv's first chunk.
template <typename T, int EPI, int CURP>
__device__ __forceinline__ void tower_tail(CPipe& p, const char* smem, int lane, f32x16 (&acc)[2], typename Elem<T>::vec8 (&A)[2][8],
                                           typename Elem<T>::vec8 (&B)[2][4], f32x16 (&xf)[2], Packed<2>& xk)
{
    using V = typename Elem<T>::vec8;
    constexpr int cur = CURP, nxt = CURP ^ 1;
    const unsigned a_off = cpipe_step(p) + lane * 16;
    T8_RDA(nxt, 0, a_off); T8_RDA(nxt, 1, a_off); acc[0] = Elem<T>::mfma(A[cur][0], B[cur][0], acc[0]); T8_FENCE();
    T8_RDA(nxt, 2, a_off); T8_RDA(nxt, 3, a_off); acc[0] = Elem<T>::mfma(A[cur][2], B[cur][1], acc[0]); T8_FENCE();
    T8_RDA(nxt, 4, a_off); T8_RDA(nxt, 5, a_off); acc[0] = Elem<T>::mfma(A[cur][4], B[cur][2], acc[0]); T8_FENCE();
    T8_RDA(nxt, 6, a_off); T8_RDA(nxt, 7, a_off); acc[0] = Elem<T>::mfma(A[cur][6], B[cur][3], acc[0]); T8_FENCE();
    acc[1] = Elem<T>::mfma(A[cur][1], B[cur][0], acc[1]); T8_FENCE();
    acc[1] = Elem<T>::mfma(A[cur][3], B[cur][1], acc[1]); T8_FENCE();
    xk.o[0][0] = epi_group<T, EPI, 0>(acc[0], xf[0]); xk.o[0][1] = epi_group<T, EPI, 1>(acc[0], xf[0]);
    acc[1] = Elem<T>::mfma(A[cur][5], B[cur][2], acc[1]); T8_FENCE();
    xk.o[0][2] = epi_group<T, EPI, 2>(acc[0], xf[0]); xk.o[0][3] = epi_group<T, EPI, 3>(acc[0], xf[0]);
    acc[1] = Elem<T>::mfma(A[cur][7], B[cur][3], acc[1]); T8_FENCE();
    xk.o[1][0] = epi_group<T, EPI, 0>(acc[1], xf[1]); xk.o[1][1] = epi_group<T, EPI, 1>(acc[1], xf[1]);
    xk.o[1][2] = epi_group<T, EPI, 2>(acc[1], xf[1]); xk.o[1][3] = epi_group<T, EPI, 3>(acc[1], xf[1]);
}
#undef T8_RDA

// zero the halo pixels of the two images of `stride` bytes per pixel at `base` (256 threads)
template <int STRIDE>
__device__ __forceinline__ void zero_halo(char* base, int board_bytes, int t)
{
    const u32x4 z = { 0, 0, 0, 0 };
    for (int i = t; i < TW_NB * NPIX; i += 256) {
        const int pp = i % NPIX, yy = pp / PITCH, xx = pp % PITCH;
        if (yy >= 1 && yy <= 8 && xx >= 1 && xx <= 8) continue;
        char* d = base + (i / NPIX) * board_bytes + pp * STRIDE;
#pragma unroll
        for (int k = 0; k < STRIDE / 16; ++k) *reinterpret_cast<u32x4*>(d + k * 16) = z;
    }
}

// The value head's second half on the four helper waves: valuefc + tanh -> [B][256] (nn.cpp:86-88), thread j = output j.
// The row is requested behind [BL], waits in registers while the compute waves reduce their logits ([BS1], [BS2]) and
// is used while they scale and store the policy rows: off the workgroup's tail.  v64: the value conv's [TW_NB][64] in LDS.
#if KAMI_TOWER_STAMP
#define FC_STAMP(k) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); if (lane == 0) fc_stamps[k] = t_; } while (0)
#define FC_STAMP_ARG , stamps + wave * 32
#else
#define FC_STAMP(k) do { } while (0)
#define FC_STAMP_ARG
#endif
template <bool LEGAL>
__device__ __forceinline__ void helper_value_fc(const TowerArgs& a, const float* v64, int b0, int j, int lane, unsigned long long* fc_stamps = nullptr)
{
    asm volatile("s_barrier" ::: "memory");                                                 // [BL]
    unsigned voff = (unsigned)j * 16u;        // (opaque: the optimiser would hoist 16 address pairs out of the group loop and spill them)
    asm volatile("" : "+v"(voff));
    float4 fcw[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
        fcw[k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.fcw4) + (size_t)k * KH_VALUE_WIDTH * 16 + voff);
    const float fcbias = a.fcb[j];
    __builtin_amdgcn_sched_barrier(0);
    FC_STAMP(10);
    asm volatile("s_barrier" ::: "memory");                                                 // [BS1]
    FC_STAMP(11);
    asm volatile("s_barrier" ::: "memory");                                                 // [BS2]
    FC_STAMP(12);
    // (the sums between [BS1] and [BS2] instead of behind [BS2]: no difference, 30.58 / 30.62 us on one device)
    float s[TW_NB];
#pragma unroll
    for (int bb = 0; bb < TW_NB; ++bb) s[bb] = 0.0f;
    static_assert(TW_NB == 2, "value FC: two boards per workgroup");
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        // four k-groups at a time: the reads' base is re-defined behind the previous four groups' sums, so that the 32
        // broadcast reads of v64 are not all hoisted to the top (128 registers, and spills beside the stream's buffers)
        const float* vb = v64 + (k & ~3) * 4;
        if ((k & 3) == 0) asm volatile("" : "+v"(vb), "+v"(s[0]), "+v"(s[1]));
        const float4 w = fcw[k];
#pragma unroll
        for (int bb = 0; bb < TW_NB; ++bb) {
            const float4 x = *reinterpret_cast<const float4*>(vb + bb * 64 + (k & 3) * 4);
            s[bb] = fmaf(x.x, w.x, s[bb]); s[bb] = fmaf(x.y, w.y, s[bb]);
            s[bb] = fmaf(x.z, w.z, s[bb]); s[bb] = fmaf(x.w, w.w, s[bb]);
        }
    }
    FC_STAMP(13);
    bool nan = false;
#pragma unroll
    for (int bb = 0; bb < TW_NB; ++bb) {
        if (b0 + bb < a.B) {
            const float r = tanhf(s[bb] + fcbias);        // (1 - 2 / (1 + e^2x) on v_exp / v_rcp instead: no change, 32.05 vs 32.10 us — the
                                                          //  2 000 clocks between the sums and the end are the stores queueing behind the policy rows')
            nan |= (r != r);
#if defined(T8_VFULL_PLAIN)
            a.vfull[(size_t)(b0 + bb) * KH_VALUE_WIDTH + j] = r;
#else
            store_wt(a.vfull + (size_t)(b0 + bb) * KH_VALUE_WIDTH + j, r);
#endif
            if (LEGAL && j == 0) a.lg_values[b0 + bb] = r;               // column 0: the position's value
        }
    }
    if (__any(nan) && lane == 0) raise_flag<LEGAL>(a, 1);
}

// ---------------------------------------------------------------- the kernel
// KS_STEM = padded input planes / 16 (2 for F <= 32, 8 for F <= 128); LEGAL: legal-move mode (TowerArgs::lg_*).
template <typename T, int KS_STEM, bool LEGAL = false>
__global__ __launch_bounds__(512) void tower8_kernel(TowerArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int FP = KS_STEM * 16;
    constexpr int SSTR = FP * 2 + 16;
    constexpr int SBOARD = NPIX * SSTR;
    constexpr int ST_SIZE = st_size(FP);
    constexpr int LDS_PAR = LDS_ST + ST_SIZE;
    constexpr int LDS_L = LDS_X;
    using V = typename Elem<T>::vec8;
    // steps (= chunks = barriers) per board group, phase by phase: both roles walk exactly these
    constexpr int NSTEM = LayerShape<9, KS_STEM, 2>::NCH;     // (33..128 planes: 2 x 9 chunks, the two 64-channel passes)
    constexpr int P1 = NSTEM & 1;                            // register-set parity after the stem
    constexpr int NLAYER = LayerShape<9, TW_CP / 16, 2>::NCH;                                   // 9 per 3x3 layer
    constexpr int NPOL = LayerShape<1, TW_CP / 16, 4>::NCH + LayerShape<1, KH_POLICY_MID / 16, 4>::NCH + (P1 ? 1 : 0);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = tid & 255;              // thread index within the role
    const int cw = wave & 3;               // wave index within the role
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
    T8_STAMP(0);
#endif
    const int ngroups = (a.B + TW_NB - 1) / TW_NB;

    if (wave >= 6) {
        // =====================================================================================  plane waves (2)
        // The input planes -> T in the S image, the T image's zero halo, the value head's FC.  These two waves issue no
        // LDS-DMA, so the compiler's own counted waits on their loads are exact: a quarter is converted as soon as
        // ITS loads have landed, whatever is still in flight behind them.
        const int pt = tid & 127;
        for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
            const int b0 = grp * TW_NB;
            if (KS_STEM == 8) {
                // ---- 33..128 planes, four quarters of 32: the stem walks the quarters in this order (BM_QUARTERS) and
                //      starts when the FIRST has been converted; quarter q is converted, two items at a time between the
                //      stem's barriers, before the barrier that precedes the first read of its chunks.  Item j of a thread =
                //      (board, pixel) (pt + 128 j) / 8, channels 32 q + 4 (pt % 8) .. +3: a wave load covers 8 pixels x
                //      128 contiguous bytes.  Loads are clamped into the row / batch and masked afterwards.
                float4_u pl[4][8];
                const int lc = pt & 7;
                const int nb = min(TW_NB, a.B - b0);                 // live boards of this group; a dead one re-reads the last live one
                const char* gbase = reinterpret_cast<const char*>(a.in + (size_t)b0 * 64 * F);     // wave-uniform
                unsigned voff[8];                                    // byte offset of item j's pixel row from gbase
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int ip = (pt >> 3) + 16 * j;
                    voff[j] = (unsigned)((min(ip >> 6, nb - 1) * 64 + (ip & 63)) * F) * 4u;
                }
                // S offset of item 0's pixel, + this lane's 8 bytes; item j adds (j >> 2) boards and 2 (j & 3) pixel rows
                const unsigned sdst = LDS_ST + ((((pt >> 6) + 1) * PITCH + ((pt >> 3) & 7) + 1) * SSTR) + 8 * lc;
                unsigned badmax = 0u;                                // max over every plane value's bits without the sign: >= 0x7f800000 iff one is NaN or Inf
                auto request = [&](int q) {
                    // read-once stream: non-temporal, so that the XCD's L2 keeps the weight stream
                    const unsigned cc = (unsigned)min(32 * q + 4 * lc, F - 4) * 4u;
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        pl[q][j] = __builtin_nontemporal_load(reinterpret_cast<const float4_u*>(gbase + voff[j] + cc));
                };
                auto convert = [&](int q, int j) {
                    float x[4] = { pl[q][j].x, pl[q][j].y, pl[q][j].z, pl[q][j].w };
                    if (32 * q + 32 > F) {                           // (wave-uniform) a quarter that holds the row's ragged end
                        const int c = 32 * q + 4 * lc;
                        const int sh = c - min(c, F - 4);            // the load was moved back by sh channels; channels >= F are zero
                        float y[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            y[k] = sh == 0 ? x[k] : 0.0f;
#pragma unroll
                            for (int m = 1; m < 4; ++m)
                                if (k + m < 4) y[k] = sh == m ? x[k + m] : y[k];
                        }
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[k] = y[k];
                    }
                    // (integer max on the bits: an x * 0 sum is folded away by the optimiser, four v_cmp_class + s_or per item
                    //  made the plane waves late at the stem's barriers — +2 000 cycles on the stem)
                    badmax = max(max(badmax, __float_as_uint(x[0]) & 0x7fffffffu), __float_as_uint(x[1]) & 0x7fffffffu);
                    badmax = max(max(badmax, __float_as_uint(x[2]) & 0x7fffffffu), __float_as_uint(x[3]) & 0x7fffffffu);
                    u32x2 o;
                    o.x = pack2<T>(x[0], x[1]); o.y = pack2<T>(x[2], x[3]);
                    *reinterpret_cast<u32x2*>(smem + sdst + (j >> 2) * SBOARD + (j & 3) * 2 * PITCH * SSTR + 64 * q) = o;
                };
                // everything requested up front: the issue stalls at the memory pipeline's depth (two quarters), but all four
                // quarters have landed ~900 cycles after the last request and the pipeline is then free for the stem's
                // weights (two quarters in flight and the others requested between the steps: 0.4 % slower, same device)
                request(0); request(1); request(2); request(3);
                T8_STAMP(6);
#pragma unroll
                for (int j = 0; j < 8; ++j) convert(0, j);
                T8_STAMP(8);
                lds_barrier();                                                              // [B0]
                T8_STAMP(2);
                // quarter q's first chunk is chunk (18 q + 3) / 4 = 0, 4, 9, 13: read behind that step's barrier
                //   slot (between barriers)   s0 s1 s2 s3 | s4 s5 s6 s7 s8 | s9 s10 s11 s12 | ...
                //   items converted           q1: 2 2 2 2 | q2: 2 2 2 1 1  | q3: 2  2   2   2
#pragma unroll
                for (int i = 0; i < NSTEM; ++i) {
                    if (i < 4) { convert(1, 2 * i); convert(1, 2 * i + 1); }
                    else if (i < 7) { convert(2, 2 * (i - 4)); convert(2, 2 * (i - 4) + 1); }
                    else if (i < 9) convert(2, i - 1);
                    else if (i < 13) { convert(3, 2 * (i - 9)); convert(3, 2 * (i - 9) + 1); }
                    if (i == 3 || i == 8 || i == 12) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // a quarter complete
                    asm volatile("s_barrier" ::: "memory");                                 // [stem step i]
                }
                const bool bad = badmax >= 0x7f800000u;
                // a NaN/Inf plane value makes the reference's policy NaN (nn.cpp:176): same verdict here
                if (__any(bad) && lane == 0) raise_flag<LEGAL>(a, 0);
            } else {
                if (a.boards) {
                    // ---- compact ingest: Env::observe (env.h:202-262) straight into S, one thread per (board, square)
                    const int bb = pt >> 6, p = pt & 63;
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
                } else {
                    // ---- planes fp32 [b][64][F], F <= 32 -> T in S (interior pixels, all FP channels)
                    constexpr int CH = FP / 8;                       // 8-channel (16-byte) chunks per pixel of S
                    constexpr int NIT = TW_NB * 64 * CH / 128;       // (board, pixel, chunk) items per thread
                    float vin[NIT][8];
                    const int c0 = (pt % CH) * 8;
#pragma unroll
                    for (int j = 0; j < NIT; ++j) {
                        const int i = pt + 128 * j;
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
                    bool bad = false;
#pragma unroll
                    for (int j = 0; j < NIT; ++j) {
                        const int i = pt + 128 * j;
                        const int bb = i / (64 * CH), p = (i / CH) & 63;
                        u32x4 o;
                        unsigned w[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float lo = vin[j][2 * k], hi = vin[j][2 * k + 1];
                            bad = bad || !__builtin_isfinite(lo) || !__builtin_isfinite(hi);
                            w[k] = pack2<T>(lo, hi);
                        }
                        o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
                        *reinterpret_cast<u32x4*>(smem + LDS_ST + bb * SBOARD + (((p >> 3) + 1) * PITCH + (p & 7) + 1) * SSTR + (pt % CH) * 16) = o;
                    }
                    if (__any(bad) && lane == 0) raise_flag<LEGAL>(a, 0);
                }
                lds_barrier();                                                              // [B0]
                T8_STAMP(2);
                for (int i = 0; i < NSTEM; ++i) asm volatile("s_barrier" ::: "memory");     // [stem steps]
            }
            T8_STAMP(5);
            const int nsteps = 2 * R * NLAYER + NPOL;
            for (int i = 0; i < nsteps; ++i) {
                asm volatile("s_barrier" ::: "memory");                                     // [tower + policy steps]
                if (i == 0 && R > 0) {
                    // behind the first tower step's barrier every compute wave's last read of S is complete: T, which
                    // shares LDS with S, gets its zero halo, nine steps before conv2 of the first block reads through it
                    for (int z = 0; z < 2; ++z) zero_halo<XSTR>(smem + LDS_ST, XBOARD, pt + 128 * z);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
            T8_STAMP(19);
            helper_value_fc<LEGAL>(a, v64, b0, tid & 255, lane FC_STAMP_ARG);                            // [BL] [BS1] [BS2] inside
            T8_STAMP(21);
            asm volatile("s_barrier" ::: "memory");                                         // [BE]
            T8_STAMP(22);
        }
    } else if (wave >= 4) {
        // =====================================================================================  stream waves (2)
        // The weight stream (each wave one half of every 8 KB chunk); behind a group's last step also its half of the value FC.
        const int hw = wave - 4;
        const int nsteps = NSTEM + 2 * R * NLAYER + NPOL;
        Pipe pipe;
        pipe.stream = a.wstream; pipe.nch = a.nchunks; pipe.next = 0; pipe.islot = 0; pipe.cslot = 0; pipe.ring = LDS_RING;
#pragma unroll
        for (int i = 0; i < RING_D - 1; ++i) pipe_issue4(pipe, hw, lane);

        for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
            // chunks 0 and 1 of this group have landed (chunks 2, 3, 4 may still be in flight)
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (RING_D - 3)) : "memory");                 // [B0]
            T8_STAMP(2);
            for (int i = 0; i < nsteps; ++i) hpipe_step<0>(pipe, hw, lane);
            T8_STAMP(19);
            // (the stream is at rest until the next group's first step: these loads and stores are the youngest
            //  operations, every ring piece older than them has long landed)
            helper_value_fc<LEGAL>(a, v64, grp * TW_NB, tid & 255, lane FC_STAMP_ARG);                   // [BL] [BS1] [BS2] inside
            T8_STAMP(21);
            asm volatile("s_barrier" ::: "memory");                                         // [BE]
            T8_STAMP(22);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // drain the prefetch ring before exit
    } else {
        // =====================================================================================  compute waves
        const int h = lane >> 5;
        // this wave's 32-pixel column tile: board cw>>1, rows 4*(cw&1)..+3
        const int wb = cw >> 1;
        const int lp = PIXMAP[lane & 31];
        const int py = 4 * (cw & 1) + (lp >> 3), px = lp & 7;
        const unsigned xin = LDS_X + wb * XBOARD + (py * PITCH + px) * XSTR + h * 16;       // tap (0,0) = pixel (y-1,x-1)
        const unsigned tin = LDS_ST + wb * XBOARD + (py * PITCH + px) * XSTR + h * 16;
        const unsigned sin = LDS_ST + wb * SBOARD + (py * PITCH + px) * SSTR + h * 16;
        const unsigned xout = LDS_X + wb * XBOARD + ((py + 1) * PITCH + px + 1) * XSTR;     // own pixel
        const unsigned tout = LDS_ST + wb * XBOARD + ((py + 1) * PITCH + px + 1) * XSTR;
        // per-lane bases of the boundary's accesses (opaque to the optimiser: it would otherwise fold each access's
        // constant into an address register of its own, dozens of them, and spill)
        unsigned xout_h = xout + 8 * h, tout_h = tout + 8 * h, sh0 = LDS_PAR + 16 * h;
        asm volatile("" : "+v"(xout_h), "+v"(tout_h), "+v"(sh0));

        CPipe pipe;
        pipe.ring = LDS_RING; pipe.cslot = 0;
        // parameter block -> LDS: all of a thread's float4 requested before the first is stored (npar is a multiple of 4;
        // a dependent load -> store round trip per iteration cost 5 000 cycles behind the plane waves' traffic)
        {
            const float4* src = reinterpret_cast<const float4*>(a.params);
            float4* dst = reinterpret_cast<float4*>(par);
            const int n4 = a.npar >> 2;
            for (int i0 = 0; i0 < n4; i0 += 4 * 256) {
                const float4 v0 = src[min(i0 + ct, n4 - 1)], v1 = src[min(i0 + 256 + ct, n4 - 1)];
                const float4 v2 = src[min(i0 + 512 + ct, n4 - 1)], v3 = src[min(i0 + 768 + ct, n4 - 1)];
                if (i0 + ct < n4) dst[i0 + ct] = v0;
                if (i0 + 256 + ct < n4) dst[i0 + 256 + ct] = v1;
                if (i0 + 512 + ct < n4) dst[i0 + 512 + ct] = v2;
                if (i0 + 768 + ct < n4) dst[i0 + 768 + ct] = v3;
            }
        }
        T8_STAMP(1);

        V A[2][8];                             // two register sets of weight fragments (current / next chunk)
        bool first = true;

        for (int grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {
            const int b0 = grp * TW_NB;
            // zero the halo pixels of X and S (the previous group's logits lived there); the helpers fill S's interior
            zero_halo<XSTR>(smem + LDS_X, XBOARD, ct);
            zero_halo<SSTR>(smem + LDS_ST, SBOARD, ct);
            T8_STAMP(2);
            lds_barrier();                                                                  // [B0]
            if (first) {
                first = false;
#pragma unroll
                for (int f = 0; f < 8; ++f) A[0][f] = *reinterpret_cast<const V*>(smem + LDS_RING + lane * 16 + f * 1024);
            }

            // This wave's tile of the residual stream (its own 32 pixels x 64 channels), packed: the skip operand,
            // the centre-tap operand of the next conv and the input of both heads; X carries it for the neighbours' taps.
            Packed<2> xk;
            f32x16 xf[2];          // the same tile in fp32: the skip operand and the value head's input
#pragma unroll
            for (int i = 0; i < 16; ++i) { xf[0][i] = 0.0f; xf[1][i] = 0.0f; }
            T8_STAMP(3);
            // ---- stem: conv1 + batchnorm1 + relu, S -> X (nn.cpp:62-65): all but its last chunk
            f32x16 accA[2], accB[2];       // the running layer's sums / the next layer's (alternating roles)
            V B[2][4];
            acc_init<2>(accA, shift3, h);
            {
                // 33..128 planes: four passes of 32 channels in one 18-chunk walk
                constexpr int BMS = KS_STEM == 8 ? BM_QUARTERS : BM_RASTER;
#pragma unroll
                for (int k = 0; k < 4; ++k) B[0][k] = *reinterpret_cast<const V*>(smem + sin + b_off8<9, KS_STEM, BMS>(k, SSTR));
                gemm8_steps<T, 9, KS_STEM, 2, 0, 0, BMS, 0, NSTEM - 1>(pipe, smem, lane, sin, SSTR, accA, A, B);
            }
            T8_STAMP(4);
            // (T shares LDS with S and the tower's image reads go through T's halo: the helper waves clear it behind the
            //  first tower step's barrier, after every wave's last read of S and long before conv2 of the first block.)
            constexpr int CP1 = P1 ^ 1;      // register set of the last chunk of the stem and of every conv2
            // ---- residual tower: x = x + relu(bn2(conv2(relu(bn1(conv1 x)))))   nn.cpp:26-34
            // The stem's end is a residual end with x = 0 (0 + relu(acc) = relu(acc) exactly): ONE boundary form at the
            // loop's top, one tail behind it, no branch inside.
            T8_STAMP(5);
            for (int r = 0; r < R; ++r) {
                boundary<T, EPI_RESID, CP1>(pipe, smem, lane, accA, accB, sh0 + (1 + 2 * r) * TW_CP * 4, A, B, xf, xout_h, xin);
                gemm8_steps<T, 9, TW_CP / 16, 2, P1, 4, BM_CENTRE, 1, 8, true>(pipe, smem, lane, xin, XSTR, accB, A, B);        // conv1
                T8_STAMP(6 + 2 * r);
                boundary<T, EPI_PACK, P1>(pipe, smem, lane, accB, accA, sh0 + (2 + 2 * r) * TW_CP * 4, A, B, xf, tout_h, tin);
                gemm8_steps<T, 9, TW_CP / 16, 2, P1 ^ 1, 4, BM_CENTRE, 1, 8, true>(pipe, smem, lane, tin, XSTR, accA, A, B);    // conv2
                T8_STAMP(7 + 2 * r);
            }
            tower_tail<T, EPI_RESID, CP1>(pipe, smem, lane, accA, A, B, xf, xk);

            // ---- value head, first half: valueconv + vbatchnorm + relu (nn.cpp:83-85) on the fp32 tile (see
            //      tower_mfma.hip 4a: four partial sums, halves joined by one v_permlane32_swap; a NaN or Inf anywhere
            //      in the residual stream makes the partial sum non-finite: the poisoned-stream detector)
            {
                f32x2 s01 = { 0.0f, 0.0f }, s23 = { 0.0f, 0.0f };
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
            T8_STAMP(18);

            // ---- policy head: policyconv + pbatchnorm + relu (nn.cpp:72-74), 1x1: the operand is the wave's own tile
            Packed<4> pk;
            {
                f32x16 acc[4];
                V bf[4];
                packed_fragments<T, 2>(xk, bf);
                acc_init<4>(acc, pshift1, h);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // v64 written before the next barrier
                gemm8_layer<T, 1, TW_CP / 16, 4, P1, 4>(pipe, smem, lane, 0, 0, acc, A, bf);
                T8_STAMP(19);
                epilogue_pack<T, 4>(acc, pk);
            }
            // ---- policyconv2 (+bias): -> logits L[board][pixel*73 + plane]      nn.cpp:75-79
            {
                f32x16 acc[4];                  // 73 planes padded to 128 rows: whole 2-k-step chunks
                V bf[8];
                packed_fragments<T, 4>(pk, bf);
                acc_init<4>(acc, pbias2, h);
                gemm8_layer<T, 1, KH_POLICY_MID / 16, 4, P1, 8>(pipe, smem, lane, 0, 0, acc, A, bf);
                if (P1) gemm8_dummy<T, 1>(pipe, smem, lane, A);      // stream parity back to 0 for the next group
                float* lrow = reinterpret_cast<float*>(smem + LDS_L + wb * LBOARD) + (py * 8 + px) * KH_POLICY_PLANES;
#pragma unroll
                for (int ms = 0; ms < 3; ++ms)      // planes >= 96 are padding
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int plane = ms * 32 + 8 * g + 4 * h + i;
                            if (plane < KH_POLICY_PLANES) lrow[plane] = acc[ms][4 * g + i];
                        }
                lds_barrier();                                                              // [BL]
            }
            T8_STAMP(20);

            // ---- softmax over all 4672 logits of a board (nn.cpp:80): 128 threads per board, one LDS pass, each
            //      thread keeps its <= 10 float4 in registers
            {
                const int bb = ct >> 7, tt = ct & 127;
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
                if (lane == 0) red[cw] = m;
                lds_barrier();                                                              // [BS1]
                m = fmaxf(red[bb * 2], red[bb * 2 + 1]);
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < 10; ++k) {
                    v[k].x = __expf(v[k].x - m); v[k].y = __expf(v[k].y - m);
                    v[k].z = __expf(v[k].z - m); v[k].w = __expf(v[k].w - m);
                    s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
                }
                s = wave_sum_f(s);
                if (lane == 0) red[4 + cw] = s;
                lds_barrier();                                                              // [BS2]
                const float inv = 1.0f / (red[4 + bb * 2] + red[4 + bb * 2 + 1]);
                bool nan = false;
                if (LEGAL) {
                    // legal-move mode: the board's first wave does what gather_legal_kernel does with the stored row
                    // (tower_mfma.hip 4d: the same operations on the same values in the same order: the same bits)
                    nan = inv != inv;                           // a NaN logit makes the sum, hence every entry, NaN
                    if (live && (cw & 1) == 0) {
                        const float* Lb = reinterpret_cast<const float*>(smem + LDS_L + bb * LBOARD);
                        const int lo = a.lg_offsets[b0 + bb], hi = a.lg_offsets[b0 + bb + 1];
                        float sum = 0.0f;
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
                            store_wt(reinterpret_cast<f4*>(po) + tt + 128 * k, ov);
                        }
                    }
                }
                if (__any(nan) && lane == 0) raise_flag<LEGAL>(a, 0);
            }
            T8_STAMP(21);
            lds_barrier();      // [BE] L / v64 are dead; the next group may overwrite them
            T8_STAMP(22);
        }
    }
#if KAMI_TOWER_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    T8_STAMP(23);
    if (lane == 0) stamps[wave * 32 + 31] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // compute waves over the workgroup's first value row, helper waves over its second (256 floats = 4 waves x 32 stamps)
    if (lane < 32 && (int)blockIdx.x * TW_NB + 1 < a.B)
        reinterpret_cast<unsigned long long*>(a.vfull + ((size_t)blockIdx.x * TW_NB + (wave >> 2)) * KH_VALUE_WIDTH)[cw * 32 + lane] = stamps[wave * 32 + lane];
#endif
}

template <typename T, int KS_STEM, bool LEGAL = false> static hipError_t launch8(const TowerArgs& a, int grid, hipStream_t s)
{
    constexpr int FP = KS_STEM * 16;
    const int lds = LDS_ST + st_size(FP) + tower_par_floats(a.R) * 4 + (KAMI_TOWER_STAMP ? 16 + 8 * 32 * 8 : 0);
    static std::atomic<bool> attr_done{ false };      // engines are called from many host threads; setting it twice is harmless
    if (!attr_done.load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower8_kernel<T, KS_STEM, LEGAL>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done.store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL((tower8_kernel<T, KS_STEM, LEGAL>), dim3(grid), dim3(512), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_tower8(int dtype, int FP, const TowerArgs& a, int num_cus, hipStream_t s)
{
    const int ngroups = (a.B + TW_NB - 1) / TW_NB;
    const int grid = ngroups < num_cus ? ngroups : num_cus;      // one workgroup per CU (LDS-bound residency)
    if (a.lg_offsets) {                      // legal-move mode: compact records (F <= 32) only
        if (FP != 32 || !a.lg_actions || !a.lg_priors || !a.lg_values || !a.lg_flags) return hipErrorInvalidValue;
        return dtype == KH_BF16 ? launch8<__bf16, 2, true>(a, grid, s) : launch8<_Float16, 2, true>(a, grid, s);
    }
    if (dtype == KH_BF16) return FP == 32 ? launch8<__bf16, 2>(a, grid, s) : launch8<__bf16, 8>(a, grid, s);
    return FP == 32 ? launch8<_Float16, 2>(a, grid, s) : launch8<_Float16, 8>(a, grid, s);
}

}  // namespace kh
