// tower_common.h — geometry, MFMA fragment helpers, the LDS-DMA weight ring and the epilogues shared by the two
// whole-network kernels (tower_mfma.hip: 4 waves per workgroup; tower8_mfma.hip: 4 compute + 4 helper waves).
#pragma once
#include "kh_internal.h"
#include "encode_square.h"

namespace kh {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using float4_u = float __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned 16-byte load

// ---------------------------------------------------------------- geometry (bytes unless noted)
constexpr int PITCH = 12;                       // pixels per padded row (8 + halo, padded for banking)
constexpr int NPIX = 10 * PITCH;                // padded pixels per board
constexpr int XSTR = TW_CP * 2 + 16;            // 144: pixel stride of the C-channel images
constexpr int XBOARD = NPIX * XSTR;             // 17 280
constexpr int PSTR = KH_POLICY_MID * 2 + 16;    // 272: pixel stride of the policy mid image
constexpr int PBOARD = 64 * PSTR;               // dense 64 pixels (1x1 conv needs no halo)
constexpr int LBOARD = KH_PSIZE * 4;            // logits fp32 per board
#ifndef KAMI_RING_D
#define KAMI_RING_D 6
#endif
constexpr int RING_D = KAMI_RING_D;      // (7 fits the LDS too: tried for one more step of prefetch lead, see DESIGN)
constexpr int CHUNK = 8192;                     // 8 fragments of 1 KB
// The DMA ring sits at LDS offset 0 so that its addresses fit M0's 16-bit LDS offset field.
constexpr int LDS_RING = 0;
constexpr int LDS_X = RING_D * CHUNK;           // 49 152
constexpr int LDS_ST = LDS_X + TW_NB * XBOARD;  // 83 712
constexpr int ST_MIN = TW_NB * LBOARD - TW_NB * XBOARD + TW_NB * PBOARD;   // L spills past X, P at the end

__host__ __device__ constexpr int st_size(int FP)
{
    int s = TW_NB * NPIX * (FP * 2 + 16);
    return s > ST_MIN ? s : ST_MIN;
}

// lane column r (0..31) of a 32-pixel MFMA tile -> local pixel (row 0..3)*8 + x.  Hardware lane
// groups {0-3,12-15,20-27} take rows 0 and 2, {4-11,16-19,28-31} rows 1 and 3: with PITCH 12 the
// padded pixel indices of each group are distinct mod 16.
static __device__ __constant__ const unsigned char PIXMAP[32] = {
    0, 1, 2, 3,                 /* lanes 0-3   : row 0, x 0-3 */
    8, 9, 10, 11, 12, 13, 14, 15,   /* lanes 4-11  : row 1, x 0-7 */
    4, 5, 6, 7,                 /* lanes 12-15 : row 0, x 4-7 */
    24, 25, 26, 27,             /* lanes 16-19 : row 3, x 0-3 */
    16, 17, 18, 19, 20, 21, 22, 23, /* lanes 20-27 : row 2, x 0-7 */
    28, 29, 30, 31              /* lanes 28-31 : row 3, x 4-7 */
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

// Two fp32 -> one dword of two T (round to nearest even): the vector convert lowers to ONE
// v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950 (scalar casts + shifts do not), and being
// compiler-visible it gets its MFMA-result read hazards padded like any other VALU op.
using f32x2 = float __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ unsigned pack2(float lo, float hi)
{
    using T2 = T __attribute__((ext_vector_type(2)));
    const f32x2 v = { lo, hi };
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, T2));
}
// ReLU on a packed pair of bf16 / f16: both formats keep the sign in bit 15, so a signed 16-bit
// max with 0 clears negatives (and -0) and leaves everything else: one v_pk_max_i16 per 2 values.
using short2_t = short __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned relu_pk(unsigned w)
{
    const short2_t z = { 0, 0 };
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2_t, w), z));
}
template <typename T> __device__ __forceinline__ float unpack_lo(unsigned u) { return (float)__builtin_bit_cast(T, (unsigned short)(u & 0xffffu)); }
template <typename T> __device__ __forceinline__ float unpack_hi(unsigned u) { return (float)__builtin_bit_cast(T, (unsigned short)(u >> 16)); }

// ReLU as one integer max.  It squashes negative-signed NaNs (torch::relu keeps NaN), so the kernel guards
// the reference's NaN contract (nn.cpp:176-180) elsewhere: non-finite input planes are flagged
// while they are converted, and the residual stream is checked for NaN/Inf after the tower.
// (fp32: signed-integer max with 0 on the bits — one v_max_i32, no canonicalising pre-op.)
__device__ __forceinline__ float relu_nan(float v) { const int i = __float_as_int(v); return __int_as_float(i > 0 ? i : 0); }

// ---------------------------------------------------------------- weight stream (LDS-DMA ring)
struct Pipe {
    const char* stream;     // packed fragments, nch chunks of 8 KB, cyclic
    int nch;
    int next;               // chunk index to issue next
    int islot, cslot;       // ring slots: next to fill / next to consume
    unsigned ring;          // LDS byte offset of the ring
};

// One wave's quarter of a chunk: two 1 KB LDS-DMA pieces (64 lanes x 16 B each -> LDS[M0 + inst
// offset + lane*16]).  Scalar base + one constant per-lane offset: no address VALU work per step;
// the instruction offset moves the global AND the LDS address (verified on gfx950,
// tools/glds_test.hip).  M0 is written in the same statement that reads it and restored
// (cdna_hip_programming.md §5.7).
__device__ __forceinline__ void pipe_issue(Pipe& p, int wave, int lane)
{
    const char* sbase = p.stream + (size_t)p.next * CHUNK + wave * 2048;      // wave-uniform -> SGPR pair
    const unsigned dst = p.ring + p.islot * CHUNK + wave * 2048;
    const unsigned voff = lane * 16;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
    p.next = (p.next + 1 == p.nch) ? 0 : p.next + 1;
    p.islot = (p.islot + 1 == RING_D) ? 0 : p.islot + 1;
}

__device__ __forceinline__ void lds_barrier()
{
    // all of this wave's LDS writes complete, then rendezvous (no vmcnt: DMA stays in flight)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---------------------------------------------------------------- implicit-GEMM layer
// D[ms] (32 channels x 32 pixels each) += W-fragments (ring) x activation fragments (LDS image).
// b_base: this lane's byte offset of pixel (y-1, x-1) of its column's pixel, chunk h;  TAPS = 9
// walks the 3x3 window (dy*PITCH + dx) * stride, TAPS = 1 stays put.  KS 16-channel k-steps per
// tap, MS 32-channel output tiles.  One 8-fragment chunk = 8/MS k-steps.  A[PAR] holds this
// layer's first chunk on entry; on exit A[PAR ^ (chunks & 1)] holds the next layer's.
template <int TAPS, int KS, int MS> struct LayerShape {
    static constexpr int KPC = 8 / MS;                       // k-steps per chunk
    static constexpr int TK = TAPS * KS;                     // real k-steps
    static constexpr int NCH = (TK + KPC - 1) / KPC;         // chunks (the packer zero-pads the tail)
};

// CF (centre first): the 3x3 window is walked centre tap first, then the other eight in raster
// order — the centre tap's operand is the wave's own previous output and comes from registers.
template <int TAPS, int KS, bool CF = false>
__device__ __forceinline__ constexpr unsigned b_offset(int kk, int stride)
{
    // k-steps past the real ones multiply zero weights: point them at tap 0 (always in bounds)
    const int k = kk < TAPS * KS ? kk : 0;
    const int ti = k / KS, ks = k % KS;
    const int tap = !CF ? ti : (ti == 0 ? 4 : (ti <= 4 ? ti - 1 : ti));
    return (unsigned)((TAPS == 9 ? ((tap / 3) * PITCH + (tap % 3)) * stride : 0) + ks * 32);
}

// accumulator initial value = folded BatchNorm shift of this lane's channels
template <int MS>
__device__ __forceinline__ void acc_init(f32x16 (&acc)[MS], const float* shift, int h)
{
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 s = *reinterpret_cast<const float4*>(shift + ms * 32 + 8 * g + 4 * h);
            acc[ms][4 * g + 0] = s.x; acc[ms][4 * g + 1] = s.y; acc[ms][4 * g + 2] = s.z; acc[ms][4 * g + 3] = s.w;
        }
}

// A wave's output tile after the epilogue, packed to T: o[ms][g] = channels 32 ms + 8 g + 4 h + 0..3
// of this lane's pixel (the MFMA C/D layout: 4 consecutive channels per register group).
template <int MS> struct Packed { u32x2 o[MS][4]; };

// ReLU and convert (conv1 of a block, policyconv): convert first, then ReLU on the packed pairs —
// half the VALU work of clamping the 32 fp32 values.
template <typename T, int MS>
__device__ __forceinline__ void epilogue_pack(const f32x16 (&acc)[MS], Packed<MS>& pk)
{
#pragma unroll
    for (int ms = 0; ms < MS; ++ms) {
        // all eight converts of a tile, then all eight clamps: left alone hipcc alternates them
        // through one temporary register and every instruction waits for the one before it
        unsigned c[8];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            c[2 * g] = pack2<T>(acc[ms][4 * g + 0], acc[ms][4 * g + 1]);
            c[2 * g + 1] = pack2<T>(acc[ms][4 * g + 2], acc[ms][4 * g + 3]);
        }
        asm volatile("" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]), "+v"(c[5]), "+v"(c[6]), "+v"(c[7]));
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pk.o[ms][g].x = relu_pk(c[2 * g]);
            pk.o[ms][g].y = relu_pk(c[2 * g + 1]);
        }
    }
}

// End of a residual block (and the stem): xf = (SKIP ? xf : 0) + relu(acc) in fp32 — the residual
// stream of the wave's own pixels stays in fp32 registers (nn.cpp:31 adds in fp32 too), only the
// copies that feed the MFMAs are rounded to T.
template <typename T, bool SKIP>
__device__ __forceinline__ void epilogue_residual(const f32x16 (&acc)[2], f32x16 (&xf)[2], Packed<2>& pk)
{
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
        f32x16 t;
#pragma unroll
        for (int i = 0; i < 16; ++i) t[i] = relu_nan(acc[ms][i]);
        xf[ms] = SKIP ? xf[ms] + t : t;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            pk.o[ms][g].x = pack2<T>(xf[ms][4 * g + 0], xf[ms][4 * g + 1]);
            pk.o[ms][g].y = pack2<T>(xf[ms][4 * g + 2], xf[ms][4 * g + 3]);
        }
    }
}

// 8-byte packed stores of the tile to an LDS image (for the neighbouring pixels' taps).
template <int MS>
__device__ __forceinline__ void store_packed(const Packed<MS>& pk, char* smem, unsigned out_pix, int h)
{
#pragma unroll
    for (int ms = 0; ms < MS; ++ms)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<u32x2*>(smem + out_pix + (ms * 32 + 8 * g + 4 * h) * 2) = pk.o[ms][g];
}

// The packed tile as MFMA B fragments: k-step ks, slot j of lane (r, h) = channel
// 32 (ks >> 1) + 8 (2 (ks & 1) + (j >> 2)) + 4 h + (j & 3) — the packer orders the matching
// weight fragments the same way (pack_layer, `perm`), so no data moves: two register pairs per k-step.
template <typename T, int MS>
__device__ __forceinline__ void packed_fragments(const Packed<MS>& pk, typename Elem<T>::vec8 (&b)[2 * MS])
{
    using V = typename Elem<T>::vec8;
#pragma unroll
    for (int ks = 0; ks < 2 * MS; ++ks) {
        const u32x2 lo = pk.o[ks >> 1][2 * (ks & 1)], hi = pk.o[ks >> 1][2 * (ks & 1) + 1];
        const u32x4 v = { lo.x, lo.y, hi.x, hi.y };
        b[ks] = __builtin_bit_cast(V, v);
    }
}

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


// a NaN verdict (nn.cpp:176-180): OR-ed into the engine's device flags; in legal-move mode also stored to the caller's block
template <bool LEGAL>
__device__ __forceinline__ void raise_flag(const TowerArgs& a, int which)
{
    atomicOr(&a.flags[which], 1);
    if (LEGAL) __builtin_nontemporal_store(1, a.lg_flags + which);
}

// Second stage of the 128-plane ingest: one half (64 channels) of this thread's four (board, pixel)
// items -> T in the S image; flags non-finite inputs like the reference's NaN check (nn.cpp:176).
template <typename T>
__device__ __forceinline__ void ingest_half(const float4_u (&v)[4][2], int hh, char* simg, int sstr, int sboard,
                                            int b0, int tid, int lane, const TowerArgs& a)
{
    const int F = a.F;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int bp = (tid >> 3) + 32 * j, bb = bp >> 6, p = bp & 63;
        const bool live = (b0 + bb) < a.B;
        char* dst = simg + bb * sboard + (((p >> 3) + 1) * PITCH + (p & 7) + 1) * sstr + hh * 128 + 8 * (tid & 7);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = hh * 64 + 32 * q + 4 * (tid & 7);
            const int sh = c - min(c, F - 4);         // the load was moved back by sh channels; channels >= F are zero
            float x[4] = { v[j][q].x, v[j][q].y, v[j][q].z, v[j][q].w };
            if (sh != 0) {
                float y[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    y[k] = 0.0f;
#pragma unroll
                    for (int m = 1; m < 4; ++m)
                        if (k + m < 4 && sh == m) y[k] = x[k + m];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = y[k];
            }
            u32x2 o = { 0u, 0u };
            if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) bad = bad || ((__float_as_uint(x[k]) & 0x7f800000u) == 0x7f800000u);
                o.x = pack2<T>(x[0], x[1]); o.y = pack2<T>(x[2], x[3]);
            }
            *reinterpret_cast<u32x2*>(dst + 64 * q) = o;
        }
    }
    if (__any(bad) && lane == 0) raise_flag<false>(a, 0);
}

}  // namespace kh
