// kh_api.hip — the C ABI of libkamihip.so (include/kami_hip.h): engine object, weight
// preparation / hot swap, workspace slots, host-buffer and device-buffer entry points.
//
// Boundary being replaced: class kami::NN (kami/nn/nn.h:40-73, kami/nn/nn.cpp:107-222) and
// Env::observe (kami/env.h:202-262).  There is no CPU fallback anywhere in this library:
// without a gfx950 device kh_create fails with KH_ERR_NO_DEVICE.
#include "kh_internal.h"
#include "torch_archive.h"

#include <sched.h>
#include <atomic>
#include <chrono>
#include <thread>
#include <cmath>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <algorithm>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return fail(KH_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                        __FILE__, __LINE__);                                                 \
    } while (0)

struct DevMem {
    void* p = nullptr;
    size_t bytes = 0;
    ~DevMem() { if (p) (void)hipFree(p); }
    int ensure(size_t n)
    {
        if (n <= bytes) return KH_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        HIPCHK(hipMalloc(&p, n));
        bytes = n;
        return KH_OK;
    }
    template <class T> T* as() const { return static_cast<T*>(p); }
};

// page-locked host staging: one DMA per direction instead of one driver-staged copy per argument
struct PinMem {
    void* p = nullptr;
    size_t bytes = 0;
    ~PinMem() { if (p) (void)hipHostFree(p); }
    int ensure(size_t n)
    {
        if (n <= bytes) return KH_OK;
        if (p) { (void)hipHostFree(p); p = nullptr; bytes = 0; }
        n += n / 2;                              // action counts vary from call to call
        HIPCHK(hipHostMalloc(&p, n, hipHostMallocDefault));
        bytes = n;
        return KH_OK;
    }
    char* at(size_t off) const { return static_cast<char*>(p) + off; }
};

// ------------------------------------------------------------------------------- weights
// Host view of the canonical blob (order documented at kh_weight_count in kami_hip.h).
struct ConvBN { const float *w, *b, *g, *be, *rm, *rv; };
struct HostNet {
    ConvBN stem;
    std::vector<ConvBN> res;
    ConvBN pconv;
    const float *p2w, *p2b;
    ConvBN vconv;
    const float *fcw, *fcb;
};

const float* take(const float*& p, size_t n) { const float* r = p; p += n; return r; }
void take_convbn(const float*& p, ConvBN& c, size_t wn, int co)
{
    c.w = take(p, wn); c.b = take(p, co);
    c.g = take(p, co); c.be = take(p, co); c.rm = take(p, co); c.rv = take(p, co);
}

HostNet parse_blob(const float* blob, int F, int C, int R)
{
    HostNet n;
    const float* p = blob;
    take_convbn(p, n.stem, (size_t)C * F * 9, C);
    n.res.resize(2 * R);
    for (auto& c : n.res) take_convbn(p, c, (size_t)C * C * 9, C);
    take_convbn(p, n.pconv, (size_t)KH_POLICY_MID * C, KH_POLICY_MID);
    n.p2w = take(p, (size_t)KH_POLICY_PLANES * KH_POLICY_MID);
    n.p2b = take(p, KH_POLICY_PLANES);
    take_convbn(p, n.vconv, (size_t)C, 1);
    n.fcw = take(p, (size_t)KH_VALUE_WIDTH * 64);
    n.fcb = take(p, KH_VALUE_WIDTH);
    return n;
}

// One immutable, device-resident parameter set.  kh_load_weights builds a new one and swaps
// the engine's shared_ptr; calls in flight keep the old set alive until they finish.
struct Weights {
    int generation = 0;
    std::vector<float> blob;             // host copy (kh_clone)
    DevMem simple;                       // fp32 [tap][ci][co] + scale/shift per layer
    std::vector<kh::SimpleLayer> layers; // stem, 2R tower convs, policyconv, policyconv2, valueconv
    const float *fcw = nullptr, *fcb = nullptr;
    // whole-network MFMA kernel (tower_mfma.hip): packed fragment stream + folded parameters
    DevMem tw_stream, tw_stream8, tw_par, tw_fc4;      // tw_stream8: the stream in tower8_kernel's stem order (empty: same as tw_stream)
    int tw_nchunks = 0, tw_npar = 0, tw_FP = 0;
    bool tw_ok = false;
    std::string tw_why;
    // per-layer MFMA path for wide nets (layers_mfma.hip)
    DevMem ly_w, ly_shift, ly_misc;      // ly_misc: vw[CP], fcw[256*64], fcb[256], fc4[16][256][4]
    DevMem ly_w4;                        // 3x3 layers once more, packed for conv4_mfma_kernel
    DevMem ly_w2b;                       // stem + tower packed for tower256_kernel (256-channel blocks)
    bool ly_w2b_ok = false;
    DevMem ly_wh;                        // policyconv + policyconv2 packed for policy_head4_kernel
    bool ly_wh_ok = false;
    std::vector<size_t> ly_w_off, ly_shift_off, ly_w4_off;
    int ly_FP = 0, ly_CP = 0;
    float ly_vshift = 0.0f;
    bool ly_ok = false;
};

// Eval-mode BatchNorm folded to an epilogue (scale, shift):
//   bn(conv + bias) = conv * s + ((bias - mean) * s + beta),  s = gamma / sqrt(var + 1e-5)
void fold_bn(const ConvBN& c, int co, float* scale, float* shift)
{
    for (int i = 0; i < co; ++i) {
        const float s = c.g[i] / sqrtf(c.rv[i] + 1e-5f);
        scale[i] = s;
        shift[i] = (c.b[i] - c.rm[i]) * s + c.be[i];
    }
}

uint16_t f2bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // keep NaN a NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                            // round to nearest even
    return (uint16_t)(u >> 16);
}
uint16_t f2f16(float f)
{
    _Float16 h = (_Float16)f;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}

// Append one layer's MFMA A-operand fragments (v_mfma_f32_32x32x16: lane l = (r = l & 31, h = l >> 5)
// holds W[co = ms*32 + r][k = 8h + j], j = 0..7) in consumption order tap -> k-step -> ms, BN scale
// folded in before rounding, zero-padded to (MS*32, KS*16) and to a whole number of 8-fragment chunks.
// ci0: first input channel of this pass (the 128-plane stem runs as two 64-channel passes).
// centre_first: 3x3 taps in the order 4,0,1,2,3,5,6,7,8.  perm: the first `perm` k-steps of the
// stream take their activations from the consumer's packed output registers (tower_mfma.hip,
// packed_fragments): slot (h, j) of k-step ks is input channel
// 32 (ks >> 1) + 8 (2 (ks & 1) + (j >> 2)) + 4 h + (j & 3) instead of 16 ks + 8 h + j.
void pack_layer(std::vector<uint16_t>& out, int dtype, const float* w, const float* scale, int Co, int Ci,
                int taps, int KS, int MS, int ci0 = 0, bool centre_first = false, int perm = 0, bool pad = true)
{
    int kstep = 0;
    for (int ti = 0; ti < taps; ++ti) {
        const int tap = !centre_first ? ti : (ti == 0 ? 4 : (ti <= 4 ? ti - 1 : ti));
        for (int ks = 0; ks < KS; ++ks, ++kstep)
            for (int ms = 0; ms < MS; ++ms)
                for (int l = 0; l < 64; ++l) {
                    const int r = l & 31, h = l >> 5;
                    for (int j = 0; j < 8; ++j) {
                        const int co = ms * 32 + r;
                        const int ci = ci0 + (kstep < perm ? 32 * (ks >> 1) + 8 * (2 * (ks & 1) + (j >> 2)) + 4 * h + (j & 3)
                                                           : ks * 16 + 8 * h + j);
                        float v = 0.0f;
                        if (co < Co && ci < Ci) v = w[((size_t)co * Ci + ci) * taps + tap] * (scale ? scale[co] : 1.0f);
                        out.push_back(dtype == KH_BF16 ? f2bf16(v) : f2f16(v));
                    }
                }
    }
    while (pad && out.size() % 4096) out.push_back(0);
}

// Fragments of one layer for layers_mfma.hip, BN scale folded in: 8 KB chunks of 64 input channels x 64
// output channels, [Co/64][Ci/64][tap][4][2][lane][8] — 64-channel slices of the reduction outermost, so that
// the kernel's variants (whole image staged at once, or in passes of 64 / 128 channels) all walk the same
// order and agree bit for bit.
void pack_layer_generic(uint16_t* o, int dtype, const float* w, const float* scale,
                        int Co, int Ci, int taps, int CoP, int CiP)
{
    // (runs at every weight install, the trainer's included: written by index into a slice sized CoP * CiP * taps)
    const bool bf = dtype == KH_BF16;
    for (int cb = 0; cb < CoP / 64; ++cb)
        for (int slice = 0; slice < CiP / 64; ++slice)
            for (int tap = 0; tap < taps; ++tap)
                for (int kk = 0; kk < 4; ++kk)
                    for (int ms = 0; ms < 2; ++ms)
                        for (int l = 0; l < 64; ++l) {
                            const int r = l & 31, h = l >> 5, ks = slice * 4 + kk;
                            const int co = cb * 64 + ms * 32 + r, ci0 = ks * 16 + 8 * h;
                            const float sc = scale ? (co < Co ? scale[co] : 0.0f) : 1.0f;
                            const float* src = w + ((size_t)co * Ci + ci0) * taps + tap;
                            for (int j = 0; j < 8; ++j) {
                                const float v = (co < Co && ci0 + j < Ci) ? src[(size_t)j * taps] * sc : 0.0f;
                                *o++ = bf ? f2bf16(v) : f2f16(v);
                            }
                        }
}

// The same fragments for conv4_mfma_kernel (four boards x 128 output channels per workgroup): 8 KB chunks of 32 input
// channels x 128 output channels, [Co/128][Ci/64][tap][half][ks2][ms 0..3][lane][8] — the reduction walks in the same
// order as above (64-channel slices, then taps, then k-steps), so both kernels produce the same bits.
void pack_layer_wide128(uint16_t* o, int dtype, const float* w, const float* scale,
                        int Co, int Ci, int taps, int CoP, int CiP, int CBC = 128)
{
    // CBC: output channels per block — 128 (conv4_mfma_kernel, tower128_kernel, policy_head4_kernel) or 256
    // (tower256_kernel: eight row tiles per k-step, one k-step per 8 KB chunk)
    const bool bf = dtype == KH_BF16;
    for (int cb = 0; cb < CoP / CBC; ++cb)
        for (int slice = 0; slice < CiP / 64; ++slice)
            for (int tap = 0; tap < taps; ++tap)
                for (int kk = 0; kk < 4; ++kk)                  // kk = 2 * half + ks2
                    for (int ms = 0; ms < CBC / 32; ++ms)
                        for (int l = 0; l < 64; ++l) {
                            const int r = l & 31, h = l >> 5, ks = slice * 4 + kk;
                            const int co = cb * CBC + ms * 32 + r, ci0 = ks * 16 + 8 * h;
                            const float sc = scale ? (co < Co ? scale[co] : 0.0f) : 1.0f;
                            const float* src = w + ((size_t)co * Ci + ci0) * taps + tap;
                            for (int j = 0; j < 8; ++j) {
                                const float v = (co < Co && ci0 + j < Ci) ? src[(size_t)j * taps] * sc : 0.0f;
                                *o++ = bf ? f2bf16(v) : f2f16(v);
                            }
                        }
}

// fp32 fragments for conv_f32_kernel: [Co/64][Ci slices of <= 128][tap][slice/8][2][lane][4]; lane (r, h) holds
// W[co = ms*32 + r][ci = 8j + 4h + 0..3]  (one slice up to 128 input channels: the image of a slice is what fits LDS)
void pack_layer_f32(float* o, const float* w, const float* scale, int Co, int Ci, int taps, int CoP, int CiP)
{
    for (int cb = 0; cb < CoP / 64; ++cb)
      for (int c_lo = 0; c_lo < CiP; c_lo += 128)
        for (int tap = 0; tap < taps; ++tap)
            for (int j = c_lo / 8; j < (c_lo + 128 < CiP ? c_lo + 128 : CiP) / 8; ++j)
                for (int ms = 0; ms < 2; ++ms)
                    for (int l = 0; l < 64; ++l) {
                        const int r = l & 31, h = l >> 5;
                        for (int i = 0; i < 4; ++i) {
                            const int co = cb * 64 + ms * 32 + r, ci = j * 8 + 4 * h + i;
                            *o++ = (co < Co && ci < Ci) ? w[((size_t)co * Ci + ci) * taps + tap] * (scale ? scale[co] : 1.0f) : 0.0f;
                        }
                    }
}

int build_layers(Weights& W, const HostNet& n, int dtype, int F, int C, int R)
{
    const bool f32 = dtype == KH_F32;
    // bf16/f16: input channels in multiples of 64 (an 8 KB weight chunk = 4 k-steps of one tap)
    const int FP = f32 ? (F + 7) / 8 * 8 : (F + 63) / 64 * 64, CP = (C + 63) / 64 * 64;
    // LDS image of two boards: 2 x 120 x (Ci * elem + 16) bytes must fit 160 KB
    if (CP > 256 || FP > 256) return KH_OK;      // not covered: ly_ok stays false
    // Every layer's fragments are packed by its own job into its own slice: the jobs run on a few host threads (this is
    // on the trainer's path too — kh_train installs its result here — and a 20x256 net is 48 M fragments' worth).
    std::vector<uint16_t> w, w4, wh, w2b;
    std::vector<float> wf;
    std::vector<float> shift;
    struct Job { int kind; size_t off; const float* wt; std::vector<float> sc; int Co, Ci, taps, CoP, CiP; };   // kind 0 generic, 1 wide128, 2 f32, 3 head
    std::vector<Job> jobs;
    size_t nw = 0, nw4 = 0, nwf = 0, nwh = 0, nw2b = 0;
    const bool want2b = !f32 && CP == 256 && FP == 128;      // tower256_kernel's shape
    std::vector<float> sc(256), sh(256);
    auto add = [&](const float* wt, const ConvBN* bn, const float* bias, int Co, int Ci, int taps, int CoP, int CiP) {
        W.ly_shift_off.push_back(shift.size());
        W.ly_w4_off.push_back((size_t)-1);
        if (bn) fold_bn(*bn, Co, sc.data(), sh.data());
        else for (int i = 0; i < Co; ++i) { sc[i] = 1.0f; sh[i] = bias[i]; }
        const std::vector<float> scv(sc.begin(), sc.begin() + Co);
        const size_t n = (size_t)CoP * CiP * taps;
        if (f32) { W.ly_w_off.push_back(nwf); jobs.push_back({ 2, nwf, wt, scv, Co, Ci, taps, CoP, CiP }); nwf += n; }
        else {
            W.ly_w_off.push_back(nw); jobs.push_back({ 0, nw, wt, scv, Co, Ci, taps, CoP, CiP }); nw += n;
            if (taps == 9 && CoP % 128 == 0 && (CiP == 128 || CiP == 256)) {     // conv4_mfma_kernel's shapes
                W.ly_w4_off.back() = nw4; jobs.push_back({ 1, nw4, wt, scv, Co, Ci, taps, CoP, CiP }); nw4 += n;
            }
            if (want2b && taps == 9) { jobs.push_back({ 4, nw2b, wt, scv, Co, Ci, taps, CoP, CiP }); nw2b += n; }
        }
        for (int i = 0; i < CoP; ++i) shift.push_back(i < Co ? sh[i] : 0.0f);
    };
    add(n.stem.w, &n.stem, nullptr, C, F, 9, CP, FP);
    for (int i = 0; i < 2 * R; ++i) add(n.res[i].w, &n.res[i], nullptr, C, C, 9, CP, CP);
    add(n.pconv.w, &n.pconv, nullptr, KH_POLICY_MID, C, 1, KH_POLICY_MID, CP);
    add(n.p2w, nullptr, n.p2b, KH_POLICY_PLANES, KH_POLICY_MID, 1, 128, KH_POLICY_MID);
    if (!f32 && (CP == 128 || CP == 256)) {          // policy_head4_kernel's shapes: policyconv then policyconv2
        fold_bn(n.pconv, KH_POLICY_MID, sc.data(), sh.data());
        jobs.push_back({ 3, nwh, n.pconv.w, std::vector<float>(sc.begin(), sc.begin() + KH_POLICY_MID), KH_POLICY_MID, C, 1, KH_POLICY_MID, CP });
        nwh += (size_t)KH_POLICY_MID * CP;
        jobs.push_back({ 3, nwh, n.p2w, std::vector<float>(), KH_POLICY_PLANES, KH_POLICY_MID, 1, 128, KH_POLICY_MID });
        nwh += (size_t)128 * KH_POLICY_MID;
    }
    w.resize(nw); w4.resize(nw4); wf.resize(nwf); wh.resize(nwh); w2b.resize(nw2b);
    {
        std::atomic<size_t> next{ 0 };
        auto run = [&]() {
            for (size_t j; (j = next.fetch_add(1)) < jobs.size();) {
                const Job& jb = jobs[j];
                const float* scp = jb.sc.empty() ? nullptr : jb.sc.data();
                if (jb.kind == 0) pack_layer_generic(w.data() + jb.off, dtype, jb.wt, scp, jb.Co, jb.Ci, jb.taps, jb.CoP, jb.CiP);
                else if (jb.kind == 1) pack_layer_wide128(w4.data() + jb.off, dtype, jb.wt, scp, jb.Co, jb.Ci, jb.taps, jb.CoP, jb.CiP);
                else if (jb.kind == 2) pack_layer_f32(wf.data() + jb.off, jb.wt, scp, jb.Co, jb.Ci, jb.taps, jb.CoP, jb.CiP);
                else if (jb.kind == 4) pack_layer_wide128(w2b.data() + jb.off, dtype, jb.wt, scp, jb.Co, jb.Ci, jb.taps, jb.CoP, jb.CiP, 256);
                else pack_layer_wide128(wh.data() + jb.off, dtype, jb.wt, scp, jb.Co, jb.Ci, jb.taps, jb.CoP, jb.CiP);
            }
        };
        const int nt = (int)std::min<size_t>(8, jobs.size());
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(run);
        run();
        for (auto& t : th) t.join();
    }
    std::vector<float> misc((size_t)CP + KH_VALUE_WIDTH * 64 + KH_VALUE_WIDTH + (size_t)KH_VALUE_WIDTH * 64, 0.0f);      // ... + fc4
    float vs, vsh;
    fold_bn(n.vconv, 1, &vs, &vsh);
    for (int i = 0; i < C; ++i) misc[i] = n.vconv.w[i] * vs;
    memcpy(misc.data() + CP, n.fcw, sizeof(float) * KH_VALUE_WIDTH * 64);
    memcpy(misc.data() + CP + (size_t)KH_VALUE_WIDTH * 64, n.fcb, sizeof(float) * KH_VALUE_WIDTH);
    {
        // valuefc.weight once more as [k / 4][output][4]: 64 lanes that take 64 consecutive outputs read 1 KB in one piece per
        // k-group (policy_head4_kernel / tower128_kernel's value FC; from the [256][64] rows every lane's 16 bytes were a
        // cache line of their own: 24 000 clocks of a 48 000-clock head)
        float* fc4 = misc.data() + CP + (size_t)KH_VALUE_WIDTH * 64 + KH_VALUE_WIDTH;
        for (int j = 0; j < KH_VALUE_WIDTH; ++j)
            for (int k = 0; k < 64; ++k) fc4[((size_t)(k / 4) * KH_VALUE_WIDTH + j) * 4 + (k & 3)] = n.fcw[(size_t)j * 64 + k];
    }
    W.ly_vshift = vsh; W.ly_FP = FP; W.ly_CP = CP;
    const void* wsrc = f32 ? (const void*)wf.data() : (const void*)w.data();
    const size_t wbytes = f32 ? wf.size() * 4 : w.size() * 2;
    if (W.ly_w.ensure(wbytes) || W.ly_shift.ensure(shift.size() * 4) || W.ly_misc.ensure(misc.size() * 4)) return KH_ERR_HIP;
    HIPCHK(hipMemcpy(W.ly_w.p, wsrc, wbytes, hipMemcpyHostToDevice));
    if (!w4.empty()) {
        if (W.ly_w4.ensure(w4.size() * 2)) return KH_ERR_HIP;
        HIPCHK(hipMemcpy(W.ly_w4.p, w4.data(), w4.size() * 2, hipMemcpyHostToDevice));
    }
    if (!w2b.empty()) {
        if (W.ly_w2b.ensure(w2b.size() * 2)) return KH_ERR_HIP;
        HIPCHK(hipMemcpy(W.ly_w2b.p, w2b.data(), w2b.size() * 2, hipMemcpyHostToDevice));
        W.ly_w2b_ok = true;
    }
    if (!wh.empty()) {
        if (W.ly_wh.ensure(wh.size() * 2)) return KH_ERR_HIP;
        HIPCHK(hipMemcpy(W.ly_wh.p, wh.data(), wh.size() * 2, hipMemcpyHostToDevice));
        W.ly_wh_ok = true;
    }
    HIPCHK(hipMemcpy(W.ly_shift.p, shift.data(), shift.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(W.ly_misc.p, misc.data(), misc.size() * 4, hipMemcpyHostToDevice));
    W.ly_ok = true;
    return KH_OK;
}

int build_tower(Weights& W, const HostNet& n, int dtype, int F, int C, int R)
{
    using namespace kh;
    if (C > TW_CP) { W.tw_why = "filters > 64 not supported by the MFMA tower kernel yet"; return KH_OK; }
    if (F > 128) { W.tw_why = "features > 128 not supported by the MFMA tower kernel yet"; return KH_OK; }
    const int FP = F <= 32 ? 32 : 128;
    if (tower_lds_bytes(FP, R) > 160 * 1024) { W.tw_why = "too many residual blocks for the LDS parameter area"; return KH_OK; }
    std::vector<float> sc(128), sh(128);
    std::vector<uint16_t> stream, stem8;
    std::vector<float> par((size_t)tower_par_copy_floats(R), 0.0f);
    fold_bn(n.stem, C, sc.data(), sh.data());
    if (FP == 128) {        // two 64-plane passes: the second half of the planes is still arriving during the first
        pack_layer(stream, dtype, n.stem.w, sc.data(), C, F, 9, 4, 2, 0);
        pack_layer(stream, dtype, n.stem.w, sc.data(), C, F, 9, 4, 2, 64);
        // tower8_kernel: four 32-plane passes in one unpadded run of 72 k-steps = the same 18 chunks
        for (int q = 0; q < 4; ++q) pack_layer(stem8, dtype, n.stem.w, sc.data(), C, F, 9, 2, 2, 32 * q, false, 0, false);
    } else {
        pack_layer(stream, dtype, n.stem.w, sc.data(), C, F, 9, FP / 16, 2);
    }
    memcpy(par.data(), sh.data(), sizeof(float) * C);
    for (int i = 0; i < 2 * R; ++i) {
        fold_bn(n.res[i], C, sc.data(), sh.data());
        pack_layer(stream, dtype, n.res[i].w, sc.data(), C, C, 9, TW_CP / 16, 2, 0, true, TW_CP / 16);
        memcpy(par.data() + (size_t)(1 + i) * TW_CP, sh.data(), sizeof(float) * C);
    }
    float* pshift1 = par.data() + (size_t)(1 + 2 * R) * TW_CP;
    fold_bn(n.pconv, KH_POLICY_MID, sc.data(), pshift1);
    pack_layer(stream, dtype, n.pconv.w, sc.data(), KH_POLICY_MID, C, 1, TW_CP / 16, 4, 0, false, TW_CP / 16);
    float* pbias2 = pshift1 + KH_POLICY_MID;
    memcpy(pbias2, n.p2b, sizeof(float) * KH_POLICY_PLANES);
    pack_layer(stream, dtype, n.p2w, nullptr, KH_POLICY_PLANES, KH_POLICY_MID, 1, KH_POLICY_MID / 16, 4, 0, false, KH_POLICY_MID / 16);
    if (((stream.size() / 4096) & 1) != 0) stream.resize(stream.size() + 4096, 0);   // parity chunk (see gemm_dummy)
    float* vw = pbias2 + 128;
    float vs, vsh;
    fold_bn(n.vconv, 1, &vs, &vsh);
    for (int i = 0; i < C; ++i) vw[i] = n.vconv.w[i] * vs;
    vw[TW_CP] = vsh;
    // valuefc.weight [256][64] -> [k/4][j][4] so that thread j reads coalesced float4
    std::vector<float> fc4((size_t)KH_VALUE_WIDTH * 64 + KH_VALUE_WIDTH);
    for (int j = 0; j < KH_VALUE_WIDTH; ++j)
        for (int k = 0; k < 64; ++k) fc4[((size_t)(k / 4) * KH_VALUE_WIDTH + j) * 4 + (k & 3)] = n.fcw[(size_t)j * 64 + k];
    memcpy(fc4.data() + (size_t)KH_VALUE_WIDTH * 64, n.fcb, sizeof(float) * KH_VALUE_WIDTH);

    W.tw_nchunks = (int)(stream.size() / 4096);
    W.tw_npar = (int)par.size();
    W.tw_FP = FP;
    int rc = 0;
    rc |= W.tw_stream.ensure(stream.size() * 2);
    rc |= W.tw_par.ensure(par.size() * 4);
    rc |= W.tw_fc4.ensure(fc4.size() * 4);
    if (rc) return KH_ERR_HIP;
    HIPCHK(hipMemcpy(W.tw_stream.p, stream.data(), stream.size() * 2, hipMemcpyHostToDevice));
    if (!stem8.empty()) {
        if (stem8.size() != (size_t)18 * 4096 || stem8.size() > stream.size()) return fail(KH_ERR_INVALID, "internal: stem stream size");
        memcpy(stream.data(), stem8.data(), stem8.size() * 2);          // everything behind the stem is the same
        if (W.tw_stream8.ensure(stream.size() * 2)) return KH_ERR_HIP;
        HIPCHK(hipMemcpy(W.tw_stream8.p, stream.data(), stream.size() * 2, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(W.tw_par.p, par.data(), par.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(W.tw_fc4.p, fc4.data(), fc4.size() * 4, hipMemcpyHostToDevice));
    W.tw_ok = true;
    return KH_OK;
}

int build_simple(Weights& W, const HostNet& n, int F, int C, int R)
{
    struct Plan { const float* w; int Ci, Co, taps, relu; const ConvBN* bn; const float* bias; };
    std::vector<Plan> plan;
    plan.push_back({ n.stem.w, F, C, 9, 1, &n.stem, nullptr });
    for (int i = 0; i < 2 * R; ++i) plan.push_back({ n.res[i].w, C, C, 9, 1, &n.res[i], nullptr });
    plan.push_back({ n.pconv.w, C, KH_POLICY_MID, 1, 1, &n.pconv, nullptr });
    plan.push_back({ n.p2w, KH_POLICY_MID, KH_POLICY_PLANES, 1, 0, nullptr, n.p2b });
    plan.push_back({ n.vconv.w, C, 1, 1, 1, &n.vconv, nullptr });

    size_t total = 0;
    for (auto& p : plan) total += (size_t)p.taps * p.Ci * p.Co + 2 * (size_t)p.Co;
    total += (size_t)KH_VALUE_WIDTH * 64 + KH_VALUE_WIDTH;
    std::vector<float> host(total);
    int rc = W.simple.ensure(total * sizeof(float));
    if (rc) return rc;
    float* dbase = W.simple.as<float>();
    size_t off = 0;
    for (auto& p : plan) {
        kh::SimpleLayer L;
        L.Ci = p.Ci; L.Co = p.Co; L.taps = p.taps; L.relu = p.relu;
        float* wt = host.data() + off;
        // libtorch [Co][Ci][kh][kw] -> [tap][Ci][Co]
        for (int co = 0; co < p.Co; ++co)
            for (int ci = 0; ci < p.Ci; ++ci)
                for (int k = 0; k < p.taps; ++k)
                    wt[((size_t)k * p.Ci + ci) * p.Co + co] = p.w[((size_t)co * p.Ci + ci) * p.taps + k];
        L.wt = dbase + off;
        off += (size_t)p.taps * p.Ci * p.Co;
        float* sc = host.data() + off;
        float* sh = sc + p.Co;
        if (p.bn) fold_bn(*p.bn, p.Co, sc, sh);
        else for (int i = 0; i < p.Co; ++i) { sc[i] = 1.0f; sh[i] = p.bias[i]; }
        L.scale = dbase + off; L.shift = dbase + off + p.Co;
        off += 2 * (size_t)p.Co;
        W.layers.push_back(L);
    }
    memcpy(host.data() + off, n.fcw, sizeof(float) * KH_VALUE_WIDTH * 64);
    W.fcw = dbase + off; off += (size_t)KH_VALUE_WIDTH * 64;
    memcpy(host.data() + off, n.fcb, sizeof(float) * KH_VALUE_WIDTH);
    W.fcb = dbase + off; off += KH_VALUE_WIDTH;
    HIPCHK(hipMemcpy(dbase, host.data(), total * sizeof(float), hipMemcpyHostToDevice));
    return KH_OK;
}

// ------------------------------------------------------------------------------- slots
// Per-call workspace: stream + device scratch.  kh_infer may be called concurrently from many
// host threads on one engine (nn.cpp:166 takes a shared lock); each call owns one slot.
struct Slot {
    hipStream_t stream = nullptr;
    int cap = 0;                 // boards the scratch is sized for
    DevMem in, x, t, u, ph, logits, policy, v64, vfull, flags, boards, planes, offs, acts, priors, actin, xchg;
    DevMem pack_in, pack_out;    // legal-move host path: arguments / results packed for one copy each way
    PinMem hin, hout;
    hipStream_t stream2 = nullptr;   // registered caller buffers: chunks alternate between the two streams
    bool busy = false;
    bool flags_clean = false;    // device NaN flags known to be zero
    // device-pointer API: the scratch above is shared by every caller stream, so a call on another stream
    // than the previous one first waits for that call's last kernel (event recorded behind it)
    hipEvent_t scratch_done = nullptr;
    hipStream_t scratch_stream = nullptr;
    bool scratch_pending = false;
};

struct Coalescer;       // the submit / wait queue, below

struct TrainCache {
    DevMem params, grads, work, dx, dp, dv, dloss;
    PinMem pin, pin_params;                     // batch staging; the parameter blob on its way up (a pageable source made the
                                                // upload take 0.1 ms or 10-27 ms from call to call: the runtime pins it on the fly)
    std::weak_ptr<Weights> on_device;           // the weights whose blob `params` holds right now (the previous call's result):
                                                // training them again needs no upload at all
    hipStream_t st = nullptr;
    hipGraph_t g = nullptr;
    hipGraphExec_t x = nullptr;
    int B = 0;
    float lr = 0.0f;
    bool valu = false, graph_tried = false;
    void drop_graph()
    {
        if (x) (void)hipGraphExecDestroy(x);
        if (g) (void)hipGraphDestroy(g);
        x = nullptr; g = nullptr; graph_tried = false;
    }
    ~TrainCache()
    {
        drop_graph();
        if (st) (void)hipStreamDestroy(st);
    }
};

}  // namespace

struct kh_engine {
    kh_config cfg;
    int num_cus = 256;
    bool f32_simple = false;     // KAMI_F32_SIMPLE=1: dtype f32 always runs forward_simple.hip
    int small_max = 128;         // kh_infer up to this batch AND up to 768 KB of planes takes the zero-copy path
                                 // (KAMI_SMALL_MAX; 0: never): measured 1.6x per thread at batch 16, even at 2 MB of planes
    std::mutex wmu;
    std::shared_ptr<Weights> weights;
    std::mutex smu;
    std::condition_variable scv;
    std::vector<std::unique_ptr<Slot>> slots;
    std::unique_ptr<Slot> devslot;           // scratch for the device-pointer API
    std::mutex dmu;
    Coalescer* co = nullptr;          // submit / wait queue (created on first use)
    std::atomic<Coalescer*> co_ready{ nullptr };     // the same pointer once the dispatcher runs: submitters skip co_mu
    std::atomic<bool> has_weights{ false };
    std::mutex co_mu;
    std::atomic<int> small_calls{ 0 };       // synchronous small-batch calls currently inside the engine
    std::atomic<int> co_target{ 0 }, co_wait_us{ 0 }, co_callers{ 0 };
    // kh_train's workspace, staging, stream and recorded step: kept from call to call (selfplay.cpp:266 trains again and
    // again with the same batch size and learning rate; allocating 0.1-2 GB and instantiating a ~270-node graph per
    // call cost more than a dozen SGD steps)
    std::mutex train_mu;
    TrainCache* train = nullptr;
    // caller buffers registered with kh_pin_buffer: [base, base + bytes)
    std::mutex pin_mu;
    std::vector<std::pair<const char*, size_t>> pinned;
};

namespace {

constexpr int MAX_SLOTS = 32;

int set_device(kh_engine* e) { HIPCHK(hipSetDevice(e->cfg.device)); return KH_OK; }

int slot_ensure(kh_engine* e, Slot& s, int batch, bool host_io)
{
    if (!s.stream) HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    if (batch <= s.cap) return KH_OK;
    const size_t B = batch, C = e->cfg.filters, F = e->cfg.features;
    int rc = 0;
    if (host_io) {
        rc |= s.in.ensure(B * 64 * F * 4);
        rc |= s.policy.ensure(B * KH_PSIZE * 4);
        rc |= s.vfull.ensure(B * KH_VALUE_WIDTH * 4);
        rc |= s.boards.ensure(B * sizeof(kh_board));
        rc |= s.planes.ensure(B * 64 * KH_NFEATURES * 4);
    }
    rc |= s.x.ensure(B * 64 * C * 4);
    rc |= s.t.ensure(B * 64 * C * 4);
    rc |= s.u.ensure(B * 64 * C * 4);
    rc |= s.ph.ensure(B * 64 * KH_POLICY_MID * 4);
    rc |= s.logits.ensure(B * KH_PSIZE * 4);
    rc |= s.v64.ensure(B * 64 * 4);
    rc |= s.flags.ensure(16);
    if (rc) return KH_ERR_HIP;
    s.cap = batch;
    return KH_OK;
}

struct SlotLease {
    kh_engine* e;
    Slot* s = nullptr;
    explicit SlotLease(kh_engine* e_) : e(e_)
    {
        std::unique_lock<std::mutex> lk(e->smu);
        for (;;) {
            for (auto& p : e->slots)
                if (!p->busy) { s = p.get(); break; }
            if (!s && (int)e->slots.size() < MAX_SLOTS) {
                e->slots.emplace_back(new Slot());
                s = e->slots.back().get();
            }
            if (s) { s->busy = true; return; }
            e->scv.wait(lk);
        }
    }
    ~SlotLease()
    {
        { std::lock_guard<std::mutex> lk(e->smu); s->busy = false; }
        e->scv.notify_one();
    }
};

std::shared_ptr<Weights> current_weights(kh_engine* e)
{
    std::lock_guard<std::mutex> lk(e->wmu);
    return e->weights;
}

// The forward pass on device buffers: exact-order fp32 path (forward_simple.hip).
int forward_simple(kh_engine* e, const Weights& W, Slot& s, const float* d_in, int B,
                   float* d_policy, float* d_vfull, float* d_logits_out)
{
    const int R = e->cfg.residuals;
    hipStream_t st = s.stream;
    float *x = s.x.as<float>(), *t = s.t.as<float>(), *u = s.u.as<float>();
    int* flags = s.flags.as<int>();
    HIPCHK(hipMemsetAsync(flags, 0, 16, st));
    size_t li = 0;
    kh::launch_simple_conv(W.layers[li++], d_in, nullptr, x, B, st);            // nn.cpp:62-65
    for (int r = 0; r < R; ++r) {                                              // nn.cpp:26-34
        kh::launch_simple_conv(W.layers[li++], x, nullptr, t, B, st);
        kh::launch_simple_conv(W.layers[li++], t, x, u, B, st);
        float* tmp = x; x = u; u = tmp;
    }
    float* logits = d_logits_out ? d_logits_out : s.logits.as<float>();
    kh::launch_simple_conv(W.layers[li++], x, nullptr, s.ph.as<float>(), B, st);       // nn.cpp:72-74
    kh::launch_simple_conv(W.layers[li++], s.ph.as<float>(), nullptr, logits, B, st);  // nn.cpp:75-79
    kh::launch_softmax4672(logits, d_policy, B, flags, st);                            // nn.cpp:80
    kh::launch_simple_conv(W.layers[li++], x, nullptr, s.v64.as<float>(), B, st);      // nn.cpp:83-85
    kh::launch_value_fc(s.v64.as<float>(), W.fcw, W.fcb, d_vfull, B, flags, st);        // nn.cpp:86-88
    HIPCHK(hipGetLastError());
    return KH_OK;
}

// The throughput path: one persistent kernel for the whole forward pass (tower_mfma.hip).
struct LegalDev { const int32_t* offsets; const int32_t* actions; float* priors; float* values; int* flags; };

// the weight stream in the running tower kernel's stem order (kh_internal.h: tower_variant)
static const char* tower_stream(const Weights& W)
{
    return kh::tower_variant() == 8 && W.tw_stream8.p ? W.tw_stream8.as<char>() : W.tw_stream.as<char>();
}

int forward_tower(kh_engine* e, const Weights& W, Slot& s, const float* d_in, int B,
                  float* d_policy, float* d_vfull, float* d_logits_out, const kh_board* d_boards = nullptr,
                  const LegalDev* lg = nullptr)
{
    if (!W.tw_ok) return fail(KH_ERR_INVALID, "bf16/f16 path unavailable for this configuration: %s", W.tw_why.c_str());
    if (!d_boards && (reinterpret_cast<uintptr_t>(d_in) & 15)) return fail(KH_ERR_INVALID, "input planes must be 16-byte aligned");
    hipStream_t st = s.stream;
    int* flags = s.flags.as<int>();
    if (!s.flags_clean) {               // NaN flags are only ever OR-ed by the kernel: clear on demand,
        HIPCHK(hipMemsetAsync(flags, 0, 16, st));   // not per launch (a memset node costs a launch boundary)
        s.flags_clean = true;
    }
    kh::TowerArgs a;
    a.in = d_in; a.boards = d_boards; a.B = B; a.F = e->cfg.features; a.R = e->cfg.residuals;
    a.wstream = tower_stream(W); a.nchunks = W.tw_nchunks;
    a.params = W.tw_par.as<float>(); a.npar = W.tw_npar;
    a.fcw4 = W.tw_fc4.as<float>(); a.fcb = W.tw_fc4.as<float>() + (size_t)KH_VALUE_WIDTH * 64;
    a.policy = d_policy; a.vfull = d_vfull; a.logits = d_logits_out; a.flags = flags;
    if (lg) { a.lg_offsets = lg->offsets; a.lg_actions = lg->actions; a.lg_priors = lg->priors; a.lg_values = lg->values; a.lg_flags = lg->flags; }
    HIPCHK(kh::launch_tower(e->cfg.dtype, W.tw_FP, a, e->num_cus, st));
    return KH_OK;
}

// Wide nets in bf16 / f16: one MFMA launch per layer (layers_mfma.hip), then the shared softmax / FC kernels.
int forward_layers(kh_engine* e, const Weights& W, Slot& s, const float* d_in, int B,
                   float* d_policy, float* d_vfull, float* d_logits_out)
{
    const size_t nb = B, eb = e->cfg.dtype == KH_F32 ? 4 : 2;
    int rc = 0;
    rc |= s.actin.ensure(nb * 64 * W.ly_FP * eb);
    rc |= s.x.ensure(nb * 64 * W.ly_CP * eb);
    rc |= s.t.ensure(nb * 64 * W.ly_CP * eb);
    rc |= s.u.ensure(nb * 64 * W.ly_CP * eb);
    rc |= s.ph.ensure(nb * 64 * KH_POLICY_MID * eb);
    rc |= s.logits.ensure(nb * KH_PSIZE * 4);
    rc |= s.v64.ensure(nb * 64 * 4);
    rc |= s.flags.ensure(16);
    if (rc) return KH_ERR_HIP;
    hipStream_t st = s.stream;
    int* flags = s.flags.as<int>();
    if (!s.flags_clean) {               // as in forward_tower: the kernels only ever OR into the flags, whoever reads a raised
        HIPCHK(hipMemsetAsync(flags, 0, 16, st));       // one marks the slot dirty — no memset node per forward
        s.flags_clean = true;
    }
    kh::LayersArgs L;
    L.in = d_in; L.B = B; L.F = e->cfg.features; L.FP = W.ly_FP; L.CP = W.ly_CP; L.R = e->cfg.residuals;
    L.act_in = s.actin.as<unsigned short>();
    L.act[0] = s.x.as<unsigned short>(); L.act[1] = s.t.as<unsigned short>(); L.act[2] = s.u.as<unsigned short>();
    L.pmid = s.ph.as<unsigned short>();
    L.logits = d_logits_out ? d_logits_out : s.logits.as<float>();
    L.v64 = s.v64.as<float>();
    L.w = W.ly_w.as<unsigned short>(); L.w_off = W.ly_w_off.data();
    L.w4 = W.ly_w4.as<unsigned short>(); L.w4_off = W.ly_w4_off.data();
    L.shift = W.ly_shift.as<float>(); L.shift_off = W.ly_shift_off.data();
    L.vw = W.ly_misc.as<float>(); L.vshift = W.ly_vshift;
    L.wh = W.ly_wh_ok ? W.ly_wh.as<unsigned short>() : nullptr;
    L.w2b = W.ly_w2b_ok ? W.ly_w2b.as<unsigned short>() : nullptr;
    L.policy = d_policy; L.flags = flags; L.want_logits = d_logits_out != nullptr;
    L.fcw = W.ly_misc.as<float>() + W.ly_CP; L.fcb = L.fcw + (size_t)KH_VALUE_WIDTH * 64; L.vfull = d_vfull;
    L.fc4 = L.fcb + KH_VALUE_WIDTH;
    L.num_cus = e->num_cus;
    if (W.ly_CP == 256 && W.ly_w2b_ok && e->cfg.dtype != KH_F32 && 16 * (((B + 1) / 2 + 7) / 8) <= e->num_cus) {
        // tower2s_kernel's exchange area (two workgroups per board pair at batches that leave half the chip idle)
        const int pairs_cap = e->num_cus / 2;
        if (s.xchg.ensure(kh::layers_xchg_bytes(pairs_cap)) == KH_OK) {
            L.xflag = s.xchg.as<unsigned>();
            L.xbuf = reinterpret_cast<unsigned short*>(static_cast<char*>(s.xchg.p) + kh::layers_xflag_bytes(pairs_cap));
            L.x_pairs = pairs_cap;
        }
    }
    HIPCHK(kh::launch_layers(e->cfg.dtype, L, st));              // tower, policy head + softmax (nn.cpp:72-80), value head (nn.cpp:83-88)
    HIPCHK(hipGetLastError());
    return KH_OK;
}

int forward_dispatch(kh_engine* e, const Weights& W, Slot& s, const float* d_in, int B,
                     float* d_policy, float* d_vfull, float* d_logits_out)
{
    switch (e->cfg.dtype) {
    case KH_F32:
        // exact-f32 MFMA path (layers_mfma.hip) when the shape is covered; plain VALU kernels otherwise
        // (KAMI_F32_SIMPLE=1 forces the latter: it is the order-exact anchor used by the tests)
        if (W.ly_ok && !e->f32_simple) return forward_layers(e, W, s, d_in, B, d_policy, d_vfull, d_logits_out);
        return forward_simple(e, W, s, d_in, B, d_policy, d_vfull, d_logits_out);
    case KH_BF16:
    case KH_F16:
        if (!W.tw_ok && W.ly_ok) return forward_layers(e, W, s, d_in, B, d_policy, d_vfull, d_logits_out);
        return forward_tower(e, W, s, d_in, B, d_policy, d_vfull, d_logits_out);
    default: return fail(KH_ERR_INVALID, "bad dtype %d", e->cfg.dtype);
    }
}

int check_cfg(const kh_config* c)
{
    if (!c) return fail(KH_ERR_INVALID, "null config");
    if (c->width != KH_WIDTH || c->height != KH_HEIGHT)
        return fail(KH_ERR_INVALID, "only 8x8 boards are supported (got %dx%d)", c->width, c->height);
    if (c->psize != KH_PSIZE) return fail(KH_ERR_INVALID, "psize must be %d", KH_PSIZE);
    if (c->features < 1 || c->features > 4096) return fail(KH_ERR_INVALID, "bad features %d", c->features);
    if (c->filters < 1 || c->filters > 1024) return fail(KH_ERR_INVALID, "bad filters %d", c->filters);
    if (c->residuals < 0 || c->residuals > 256) return fail(KH_ERR_INVALID, "bad residuals %d", c->residuals);
    if (c->dtype != KH_F32 && c->dtype != KH_BF16 && c->dtype != KH_F16)
        return fail(KH_ERR_INVALID, "bad dtype %d", c->dtype);
    if (c->value_mode != KH_VALUE_REFERENCE_FLAT && c->value_mode != KH_VALUE_PER_SAMPLE0)
        return fail(KH_ERR_INVALID, "bad value_mode %d", c->value_mode);
    return KH_OK;
}

// Env::observe can run inside the forward kernel: bf16 / f16 tower kernel with the encoder's 30 planes
bool fused_ingest(const kh_engine* e, const Weights& W)
{
    return e->cfg.dtype != KH_F32 && W.tw_ok && W.tw_FP == 32 && e->cfg.features == KH_NFEATURES;
}

struct LegalIO { const int32_t* offsets; const int32_t* actions; float* priors; };

bool is_pinned(kh_engine* e, const void* p, size_t bytes)
{
    std::lock_guard<std::mutex> lk(e->pin_mu);
    const char* c = static_cast<const char*>(p);
    for (auto& r : e->pinned)
        if (c >= r.first && c + bytes <= r.first + r.second) return true;
    return false;
}

// kh_infer with REGISTERED caller buffers (kh_pin_buffer) on the whole-network kernel: the copies are plain DMA out of /
// into the caller's pages, so the call is cut into four chunks that alternate between two streams — chunk k + 1's upload
// runs under chunk k's kernel and policy download.  PCIe is what bounds this ABI (30 464 B in, 18 692 B out per
// evaluation at 119 planes); pageable buffers cost the runtime a pin / unpin of the caller's pages per call on top.
int infer_host_pinned(kh_engine* e, const Weights& W, Slot& s, const float* input, int batch, float* policy, float* value)
{
    const size_t F = e->cfg.features;
    if (!s.stream2) HIPCHK(hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking));
    hipStream_t st[2] = { s.stream, s.stream2 };
    int* flags = s.flags.as<int>();
    if (!s.flags_clean) {
        HIPCHK(hipMemsetAsync(flags, 0, 16, st[0]));
        HIPCHK(hipStreamSynchronize(st[0]));
        s.flags_clean = true;
    }
    const int NCK = batch >= 256 ? 4 : (batch >= 64 ? 2 : 1);
    const int per = ((batch + NCK - 1) / NCK + 1) & ~1;            // whole board pairs per chunk
    float* d_in = s.in.as<float>();
    float* d_pol = s.policy.as<float>();
    float* d_vf = s.vfull.as<float>();
    for (int k = 0, lo = 0; lo < batch; ++k, lo += per) {
        const int n = std::min(per, batch - lo);
        hipStream_t q = st[k & 1];
        HIPCHK(hipMemcpyAsync(d_in + (size_t)lo * 64 * F, input + (size_t)lo * 64 * F, (size_t)n * 64 * F * 4, hipMemcpyHostToDevice, q));
        kh::TowerArgs a;
        a.in = d_in + (size_t)lo * 64 * F; a.boards = nullptr; a.B = n; a.F = e->cfg.features; a.R = e->cfg.residuals;
        a.wstream = tower_stream(W); a.nchunks = W.tw_nchunks;
        a.params = W.tw_par.as<float>(); a.npar = W.tw_npar;
        a.fcw4 = W.tw_fc4.as<float>(); a.fcb = W.tw_fc4.as<float>() + (size_t)KH_VALUE_WIDTH * 64;
        a.policy = d_pol + (size_t)lo * KH_PSIZE; a.vfull = d_vf + (size_t)lo * KH_VALUE_WIDTH; a.logits = nullptr; a.flags = flags;
        HIPCHK(kh::launch_tower(e->cfg.dtype, W.tw_FP, a, e->num_cus, q));
        HIPCHK(hipMemcpyAsync(policy + (size_t)lo * KH_PSIZE, d_pol + (size_t)lo * KH_PSIZE, (size_t)n * KH_PSIZE * 4, hipMemcpyDeviceToHost, q));
    }
    HIPCHK(hipStreamSynchronize(st[1]));
    // value (nn.cpp:186: the first `batch` floats of the flattened [batch,256] tensor, or column 0) and the NaN flags
    if (e->cfg.value_mode == KH_VALUE_REFERENCE_FLAT) HIPCHK(hipMemcpyAsync(value, d_vf, (size_t)batch * 4, hipMemcpyDeviceToHost, st[0]));
    else HIPCHK(hipMemcpy2DAsync(value, 4, d_vf, KH_VALUE_WIDTH * 4, 4, batch, hipMemcpyDeviceToHost, st[0]));
    int fl[4] = { 0, 0, 0, 0 };
    HIPCHK(hipMemcpyAsync(fl, flags, 16, hipMemcpyDeviceToHost, st[0]));
    HIPCHK(hipStreamSynchronize(st[0]));
    if (fl[0] | fl[1]) s.flags_clean = false;
    if (fl[0]) return fail(KH_ERR_NAN_POLICY, "inference policy output contains NaN");   // nn.cpp:176-177
    if (fl[1]) return fail(KH_ERR_NAN_VALUE, "inference value output contains NaN");     // nn.cpp:179-180
    return KH_OK;
}

// kh_infer at SMALL batches (kami's default selfplay_batch is 16): a call is latency, not bandwidth — four runtime
// copies out of / into pageable memory and their synchronisation cost more than the kernel.  So: the planes are copied by
// the CPU into the slot's page-locked block, the kernel reads them from there and writes the policy rows into the
// page-locked output block ITSELF (as the queue's launches do with records and priors), a second tiny launch puts the
// values and NaN flags behind them, completion is polled, and the CPU copies the rows out.  No copy engine, one wait.
int infer_host_small(kh_engine* e, const Weights& W, Slot& s, const float* input, int batch, float* policy, float* value)
{
    const size_t B = batch, F = e->cfg.features;
    const size_t in_bytes = B * 64 * F * 4, pol_bytes = B * KH_PSIZE * 4;
    const size_t o_val = (pol_bytes + 15) & ~(size_t)15, o_flags = o_val + ((B * 4 + 15) & ~(size_t)15);
    if (s.hin.ensure(in_bytes) || s.hout.ensure(o_flags + 16)) return KH_ERR_HIP;
    memcpy(s.hin.p, input, in_bytes);
    int rc = forward_tower(e, W, s, static_cast<const float*>(s.hin.p), batch, reinterpret_cast<float*>(s.hout.at(0)), s.vfull.as<float>(), nullptr);
    if (rc) return rc;
    hipStream_t st = s.stream;
    const int vstride = e->cfg.value_mode == KH_VALUE_REFERENCE_FLAT ? 1 : KH_VALUE_WIDTH;     // nn.cpp:186 / the value column
    int* fl = reinterpret_cast<int*>(s.hout.at(o_flags));
    kh::launch_gather_legal(nullptr, nullptr, nullptr, nullptr, batch, st, s.vfull.as<float>(), vstride, reinterpret_cast<float*>(s.hout.at(o_val)),
                            s.flags.as<int>(), fl);
    HIPCHK(hipGetLastError());
    for (int k = 0;; ++k) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) break;
        if (q != hipErrorNotReady) return fail(KH_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
        if ((k & 15) == 15) sched_yield();
    }
    memcpy(policy, s.hout.at(0), pol_bytes);                        // nn.cpp:173,185
    memcpy(value, s.hout.at(o_val), B * 4);
    if (fl[0] | fl[1]) s.flags_clean = false;
    if (fl[0]) return fail(KH_ERR_NAN_POLICY, "inference policy output contains NaN");   // nn.cpp:176-177
    if (fl[1]) return fail(KH_ERR_NAN_VALUE, "inference value output contains NaN");     // nn.cpp:179-180
    return KH_OK;
}

int infer_host(kh_engine* e, const float* input, const kh_board* boards, int batch,
               float* policy, float* value, float* value_full, float* logits, const LegalIO* legal = nullptr)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    if (batch < 1) return fail(KH_ERR_INVALID, "batch must be >= 1 (got %d)", batch);
    if ((!input && !boards) || (!policy && !legal)) return fail(KH_ERR_INVALID, "null buffer");
    int nact = 0;
    if (legal) {
        if (!legal->offsets || !legal->actions || !legal->priors) return fail(KH_ERR_INVALID, "null legal-move buffer");
        if (legal->offsets[0] != 0) return fail(KH_ERR_INVALID, "action_offsets[0] must be 0");
        for (int i = 0; i < batch; ++i)
            if (legal->offsets[i + 1] < legal->offsets[i]) return fail(KH_ERR_INVALID, "action_offsets must be non-decreasing");
        nact = legal->offsets[batch];
    }
    std::shared_ptr<Weights> W = current_weights(e);
    if (!W) return fail(KH_ERR_NO_WEIGHTS, "kh_infer before kh_load_weights");
    int rc = set_device(e);
    if (rc) return rc;
    SlotLease lease(e);
    Slot& s = *lease.s;
    if ((rc = slot_ensure(e, s, batch, true))) return rc;
    const size_t B = batch, F = e->cfg.features;
    hipStream_t st = s.stream;
    if (input && policy && value && !legal && !logits && !value_full && e->cfg.dtype != KH_F32 && W->tw_ok &&
        (reinterpret_cast<uintptr_t>(input) & 15) == 0 && is_pinned(e, input, B * 64 * F * 4) && is_pinned(e, policy, B * KH_PSIZE * 4))
        return infer_host_pinned(e, *W, s, input, batch, policy, value);
    if (input && policy && value && !legal && !logits && !value_full && e->cfg.dtype != KH_F32 && W->tw_ok && batch <= e->small_max &&
        B * 64 * F * 4 <= 768 * 1024)
        return infer_host_small(e, *W, s, input, batch, policy, value);
    const float* d_in;
    const bool fused = boards && fused_ingest(e, *W) && !logits;
    // The search's call (records or planes in, legal priors + one value per position out): everything
    // but the planes travels as ONE page-locked block per direction.
    const bool packed = legal && value && !policy && !logits && !value_full;
    auto up16 = [](size_t n) { return (n + 15) & ~(size_t)15; };
    const size_t in_boards = 0, in_offs = boards ? up16(B * sizeof(kh_board)) : 0, in_acts = in_offs + up16((B + 1) * 4),
                 in_total = in_acts + up16((size_t)nact * 4);
    const size_t out_priors = 0, out_values = up16((size_t)nact * 4), out_flags = out_values + up16(B * 4), out_total = out_flags + 16;
    const kh_board* d_boards = s.boards.as<kh_board>();
    if (packed) {
        if (s.hin.ensure(in_total) || s.hout.ensure(out_total)) return KH_ERR_HIP;
        if (s.pack_in.ensure(s.hin.bytes) || s.pack_out.ensure(s.hout.bytes)) return KH_ERR_HIP;
        if (boards) memcpy(s.hin.at(in_boards), boards, B * sizeof(kh_board));
        memcpy(s.hin.at(in_offs), legal->offsets, (B + 1) * 4);
        memcpy(s.hin.at(in_acts), legal->actions, (size_t)nact * 4);
        if (fused && e->cfg.value_mode == KH_VALUE_PER_SAMPLE0) {
            // ONE launch, no copy engine: the kernel reads records / offsets / actions from the page-locked block and writes
            // the legal priors, the values (column 0) and the NaN flags into the other one itself (tower_kernel's
            // legal-move mode; what the queue's launches do); completion is polled
            int* fl = reinterpret_cast<int*>(s.hout.at(out_flags));
            fl[0] = fl[1] = 0;
            const LegalDev lg{ reinterpret_cast<const int32_t*>(s.hin.at(in_offs)), reinterpret_cast<const int32_t*>(s.hin.at(in_acts)),
                               reinterpret_cast<float*>(s.hout.at(out_priors)), reinterpret_cast<float*>(s.hout.at(out_values)), fl };
            rc = forward_tower(e, *W, s, nullptr, batch, s.policy.as<float>(), s.vfull.as<float>(), nullptr,
                               reinterpret_cast<const kh_board*>(s.hin.at(in_boards)), &lg);
            if (rc) return rc;
            for (int k = 0;; ++k) {
                const hipError_t q = hipStreamQuery(st);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) return fail(KH_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
                if ((k & 15) == 15) sched_yield();
            }
            memcpy(legal->priors, s.hout.at(out_priors), (size_t)nact * 4);
            memcpy(value, s.hout.at(out_values), B * 4);
            if (fl[0] | fl[1]) s.flags_clean = false;
            if (fl[0]) return fail(KH_ERR_NAN_POLICY, "inference policy output contains NaN");   // nn.cpp:176-177
            if (fl[1]) return fail(KH_ERR_NAN_VALUE, "inference value output contains NaN");     // nn.cpp:179-180
            return KH_OK;
        }
        HIPCHK(hipMemcpyAsync(s.pack_in.p, s.hin.p, in_total, hipMemcpyHostToDevice, st));
        d_boards = reinterpret_cast<const kh_board*>(s.pack_in.as<char>() + in_boards);
    }
    if (boards) {
        if (!packed) HIPCHK(hipMemcpyAsync(s.boards.p, boards, B * sizeof(kh_board), hipMemcpyHostToDevice, st));
        if (!fused) kh::launch_encode_f32(d_boards, batch, s.planes.as<float>(), st);
        d_in = s.planes.as<float>();
    } else {
        HIPCHK(hipMemcpyAsync(s.in.p, input, B * 64 * F * 4, hipMemcpyHostToDevice, st));   // nn.cpp:160
        d_in = s.in.as<float>();
    }
    float* d_logits = logits ? s.logits.as<float>() : nullptr;
    if (fused) rc = forward_tower(e, *W, s, nullptr, batch, s.policy.as<float>(), s.vfull.as<float>(), nullptr, d_boards);
    else rc = forward_dispatch(e, *W, s, d_in, batch, s.policy.as<float>(), s.vfull.as<float>(), d_logits);
    if (rc) return rc;
    int flags[4] = { 0, 0, 0, 0 };
    if (packed) {
        const char* pin = s.pack_in.as<char>();
        char* pout = s.pack_out.as<char>();
        // nn.cpp:186 hands back the first `batch` floats of the flattened [batch,256] tensor; the fixed mode the value column
        const int vstride = e->cfg.value_mode == KH_VALUE_REFERENCE_FLAT ? 1 : KH_VALUE_WIDTH;
        kh::launch_gather_legal(s.policy.as<float>(), reinterpret_cast<const int32_t*>(pin + in_offs),
                                reinterpret_cast<const int32_t*>(pin + in_acts), reinterpret_cast<float*>(pout + out_priors), batch, st,
                                s.vfull.as<float>(), vstride, reinterpret_cast<float*>(pout + out_values), s.flags.as<int>(),
                                reinterpret_cast<int*>(pout + out_flags));
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(s.hout.p, s.pack_out.p, out_total, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        memcpy(legal->priors, s.hout.at(out_priors), (size_t)nact * 4);
        memcpy(value, s.hout.at(out_values), B * 4);
        memcpy(flags, s.hout.at(out_flags), 16);
        if (flags[0] | flags[1]) s.flags_clean = false;
        if (flags[0]) return fail(KH_ERR_NAN_POLICY, "inference policy output contains NaN");   // nn.cpp:176-177
        if (flags[1]) return fail(KH_ERR_NAN_VALUE, "inference value output contains NaN");     // nn.cpp:179-180
        return KH_OK;
    }
    if (legal && nact > 0) {
        if (s.offs.ensure((B + 1) * 4) || s.acts.ensure((size_t)nact * 4) || s.priors.ensure((size_t)nact * 4)) return KH_ERR_HIP;
        HIPCHK(hipMemcpyAsync(s.offs.p, legal->offsets, (B + 1) * 4, hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(s.acts.p, legal->actions, (size_t)nact * 4, hipMemcpyHostToDevice, st));
        kh::launch_gather_legal(s.policy.as<float>(), s.offs.as<int32_t>(), s.acts.as<int32_t>(), s.priors.as<float>(), batch, st);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(legal->priors, s.priors.p, (size_t)nact * 4, hipMemcpyDeviceToHost, st));
    }
    if (policy) HIPCHK(hipMemcpyAsync(policy, s.policy.p, B * KH_PSIZE * 4, hipMemcpyDeviceToHost, st));  // nn.cpp:173,185
    if (logits) HIPCHK(hipMemcpyAsync(logits, s.logits.p, B * KH_PSIZE * 4, hipMemcpyDeviceToHost, st));
    if (value_full)
        HIPCHK(hipMemcpyAsync(value_full, s.vfull.p, B * KH_VALUE_WIDTH * 4, hipMemcpyDeviceToHost, st));
    if (value) {
        if (e->cfg.value_mode == KH_VALUE_REFERENCE_FLAT)
            // nn.cpp:186: the first `batch` floats of the flattened [batch,256] tensor
            HIPCHK(hipMemcpyAsync(value, s.vfull.p, B * 4, hipMemcpyDeviceToHost, st));
        else
            HIPCHK(hipMemcpy2DAsync(value, 4, s.vfull.p, KH_VALUE_WIDTH * 4, 4, B, hipMemcpyDeviceToHost, st));
    }
    HIPCHK(hipMemcpyAsync(flags, s.flags.p, 16, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    if (flags[0] | flags[1]) s.flags_clean = false;
    if (flags[0]) return fail(KH_ERR_NAN_POLICY, "inference policy output contains NaN");   // nn.cpp:176-177
    if (flags[1]) return fail(KH_ERR_NAN_VALUE, "inference value output contains NaN");     // nn.cpp:179-180
    return KH_OK;
}


// ------------------------------------------------------------------------------- coalescing queue
// SURVEY §8(b), threading row: "per-thread stream + staging slot, OR internal queue that coalesces callers into
// bigger batches".  The slots above are the first; this is the second.  Callers hand over small batches
// (kh_submit_* returns a ticket at once, kh_wait blocks for it; the synchronous entry points use the same queue when
// other small calls are in flight), each caller copies its own rows into the open batch's merge buffers, and ONE
// dispatcher thread turns whatever has accumulated into a launch on one of four streams without waiting for it (one
// pinned block each way), polls the launches' completion words and publishes the results: every waiter copies its own
// rows out of the block.  While launches are on the device the next batch fills up, so the batch size adapts to the
// load; kh_set_coalesce adds a target size and a bounded wait for callers that know how much will be in flight (the
// self-play pool).
//
// Never blocks a submitter on its own outstanding work: tickets are a fixed pool (exhaustion -> KH_ERR_INVALID),
// merge buffers come back when their launch has completed and its rows have been fetched — by the waiters, or by the
// dispatcher for tickets nobody waits on — so the only wait inside kh_submit_* is for launches that are on the device.
constexpr int CO_ROWS = 1024;                   // boards per coalesced launch (merge buffer capacity)
constexpr int CO_ACTS = CO_ROWS * 48;           // legal actions per coalesced launch
constexpr int CO_SMALL_LEGAL = 512;             // a submission larger than this takes the direct path
constexpr int CO_SMALL_PLANES = 128;
constexpr int CO_BUFFERS = 12;                  // up to max_inflight on the device, one filling, the rest waiting for their callers to fetch

// The queue's lock.  Its critical sections are a few hundred nanoseconds (reserve rows, look at the batches) and a dozen
// workers hit it within the same microsecond when a launch hands their tickets back: with std::mutex the losers sleep
// on the futex and are woken one after the other (2-3 us each: measured as 33-37 us of "filling" per launch with 14
// workers), so it spins.  Sleeping paths (idle lane, buffers all on the device, waiters past their spin time) go
// through std::condition_variable_any, which takes any lock type.
struct SpinLock {
    // test-and-test-and-set.  (A ticket lock — FIFO hand-over, waiters on a plain load — was measured: the same with two sets
    // per worker, worse with four: 3.3-3.5 -> 2.8-3.0 M/s; a FIFO queue turns one descheduled waiter into everybody's wait.)
    std::atomic<int> held{ 0 };
    void lock()
    {
        for (int k = 0;; ++k) {
            if (!held.exchange(1, std::memory_order_acquire)) return;
            while (held.load(std::memory_order_relaxed)) {
                __builtin_ia32_pause();
                if ((++k & 1023) == 0) sched_yield();       // more threads than cores: the holder may need this one
            }
        }
    }
    bool try_lock() { return !held.exchange(1, std::memory_order_acquire); }
    void unlock() { held.store(0, std::memory_order_release); }
};

// (A waiter polls `state` in a tight loop, the dispatcher writes the other fields right before it publishes: the polled word
//  has a cache line of its own, and no two tickets share one — a dozen pollers on the lines the dispatcher was writing made
//  its hand-back 10-18 us per launch with three sets per worker.)
struct alignas(64) CoTicket {
    alignas(64) std::atomic<int> state{ 0 };    // 0 free, 1 queued, 2 done: results in the caller's buffers (waiters spin on it, then
                                                // sleep), 4 results in the batch's page-locked block, 5 somebody is copying them out
    uint32_t serial = 0;                        // (same line as `state`: written at submit time only, read by every poll)
    alignas(64) struct CoBatch* from = nullptr; // state 4 / 5: the batch that holds this ticket's rows
    int status = KH_OK;
    std::string err;
    int kind = 0, row0 = 0, rows = 0, act0 = 0, nact = 0;
    // the caller's buffers (valid until kh_wait returns): inputs for the rare per-ticket re-run, outputs for the scatter
    const kh_board* boards = nullptr; const float* planes = nullptr;
    const int32_t *offsets = nullptr, *actions = nullptr;
    float *priors = nullptr, *value = nullptr, *policy = nullptr;
};

struct CoBatch {
    int state = 0;                              // 0 free, 1 open, 2 sealed (the dispatcher owns it; after completion until the last
                                                // ticket's rows have been copied out)
    int kind = 0;                               // 0: records + legal actions -> priors; 1: planes -> full policy rows
    int rows = 0, nact = 0;
    std::atomic<int> copying{ 0 };              // submitters that have reserved rows and are still copying them in
    std::atomic<int> readers{ 0 };              // tickets whose rows are still in the block (state 4 / 5)
    unsigned* done = nullptr;                   // word in pin_out that signal_kernel sets to `serial` behind the launch
    unsigned serial = 0;
    bool full = false;
    std::chrono::steady_clock::time_point first, last;     // first / latest submission into this batch
    std::vector<CoTicket*> tickets;
    // kind 0 merges straight into page-locked memory that the kernels read and write THEMSELVES (no copy engine on
    // the path: 100 KB each way per launch is latency, not bandwidth): boards | offsets | actions in, priors | values
    // | NaN flags out
    PinMem pin_in, pin_out;
    kh_board* boards = nullptr;
    int32_t *offsets = nullptr, *actions = nullptr;
    float *priors = nullptr, *values = nullptr;
    int* flags_out = nullptr;
    Slot* lane = nullptr;                       // stream + device scratch of the launch this buffer is on (Coalescer::lanes)
    std::shared_ptr<Weights> W;                 // the weights a launch that is on the device runs on
    std::chrono::steady_clock::time_point t_seal, t_run, t_launched;
    std::vector<float> planes, vfull, policy;   // kind 1 / reference value copy-out: plain host staging for infer_host
};

struct Coalescer {
    kh_engine* e;
    SpinLock mu;
    std::condition_variable_any cv_lane, cv_done, cv_space;
    CoTicket tickets[KH_MAX_OUTSTANDING];
    CoBatch batches[CO_BUFFERS];
    std::atomic<uint64_t> free_mask{ ~0ull };   // bit i: ticket i is free (a submit takes the lowest under mu, a wait gives its own
                                                // back without the lock); KH_MAX_OUTSTANDING == 64 == its width
    std::atomic<unsigned> submits{ 0 };         // bumped by every submission: the dispatcher scans the batches (under mu) only when
                                                // it has changed, a deadline is due or a launch slot has come back
    std::atomic<bool> asleep{ false };          // the dispatcher sleeps on cv_lane: only then does a submitter notify it
    std::thread dispatcher;
    // Launches on the device at once: each on a lane = a stream of its own + device scratch.  FOUR streams, created one
    // after the other, because that is how many hardware queues the runtime spreads streams over: with a stream per
    // merge buffer (12) launches that were "in flight together" shared a queue and ran one behind the other (engine
    // call 80-110 us with four in flight against 38-41 with two).  KAMI_CO_INFLIGHT lowers it.
    static constexpr int MAX_LANES = 4;
    Slot lanes[MAX_LANES];
    bool lane_busy[MAX_LANES] = { false, false, false, false };
    int max_inflight = MAX_LANES;
    int sleepers = 0;
    int spin_us = 1000;                         // kh_wait spins this long on its ticket before it sleeps (KAMI_WAIT_SPIN_US): a
                                                // sleeper costs the dispatcher a futex wake per launch and itself 10-50 us, and one
                                                // slow cycle (> 150 us, round 2's value) used to tip a pool into that regime for good
    bool stop = false;
    int64_t launches = 0, rows_launched = 0;
    // KAMI_CO_TRACE=1: where a coalesced launch's time goes (printed when the engine is destroyed)
    bool trace = false;
    double us_fill = 0, us_copywait = 0, us_launch = 0, us_run = 0, us_finish = 0;
};

// records + legal actions -> priors + one value per position, straight out of / into the batch's page-locked blocks:
// forward kernel(s) and the gather kernel on the buffer's own stream.  Launch only: the dispatcher polls the stream
// (hipStreamQuery) and calls co_finish_legal when it has drained — the launch's latency is what every waiting caller pays.
int co_launch_legal(kh_engine* e, CoBatch& b)
{
    const int B = b.rows;
    b.W = current_weights(e);                   // kept until the launch has completed
    if (!b.W) return fail(KH_ERR_NO_WEIGHTS, "kh_infer before kh_load_weights");
    const Weights& W = *b.W;
    int rc = set_device(e);
    if (rc) return rc;
    Slot& s = *b.lane;
    if ((rc = slot_ensure(e, s, CO_ROWS, true))) return rc;        // sized once for the largest merged launch: no allocation (= device sync) mid-run
    hipStream_t st = s.stream;
    if (fused_ingest(e, W)) {
        // one launch: records in, legal priors + values + NaN flags out, all through the batch's page-locked blocks
        b.flags_out[0] = b.flags_out[1] = 0;
        const LegalDev lg{ b.offsets, b.actions, b.priors, b.values, b.flags_out };
        rc = forward_tower(e, W, s, nullptr, B, s.policy.as<float>(), s.vfull.as<float>(), nullptr, b.boards, &lg);
        if (rc) return rc;
    } else {
        kh::launch_encode_f32(b.boards, B, s.planes.as<float>(), st);
        rc = forward_dispatch(e, W, s, s.planes.as<float>(), B, s.policy.as<float>(), s.vfull.as<float>(), nullptr);
        if (rc) return rc;
        kh::launch_gather_legal(s.policy.as<float>(), b.offsets, b.actions, b.priors, B, st, s.vfull.as<float>(), KH_VALUE_WIDTH, b.values,
                                s.flags.as<int>(), b.flags_out);
    }
    // (the forward kernel's LAST workgroup writing the word itself was tried: its results then have to be system-scope
    //  stores, ~8 000 four-byte PCIe writes per launch instead of L2-combined lines — engine call 43-46 us against 38-40)
    if (++b.serial == 0) b.serial = 1;
    kh::launch_signal(b.done, b.serial, st);
    HIPCHK(hipGetLastError());
    return KH_OK;
}

int co_finish_legal(CoBatch& b)
{
    b.W.reset();
    if (b.flags_out[0] | b.flags_out[1]) b.lane->flags_clean = false;
    if (b.flags_out[0]) return fail(KH_ERR_NAN_POLICY, "inference policy output contains NaN");   // nn.cpp:176-177
    if (b.flags_out[1]) return fail(KH_ERR_NAN_VALUE, "inference value output contains NaN");     // nn.cpp:179-180
    return KH_OK;
}

// Starts a sealed batch.  Returns true when it is now on the device (records + legal actions, per-sample values: the
// self-play path) and the dispatcher has to poll for it; false when it ran synchronously (plane submissions and the
// reference's flattened value tensor go through infer_host: uploads, downloads and a stream wait) and `rc` is final.
bool co_start(kh_engine* e, CoBatch& b, int& rc)
{
    const int B = b.rows;
    const bool flat = e->cfg.value_mode == KH_VALUE_REFERENCE_FLAT;
    if (b.kind == 0 && !flat) {
        rc = co_launch_legal(e, b);
        if (rc) b.W.reset();
        return rc == KH_OK;
    }
    if (b.kind == 0) {
        // nn.cpp:186 hands back the first `batch` floats of the caller's OWN flattened [batch,256] tensor: take the
        // whole tensor and cut each caller's slice out of it below
        LegalIO l{ b.offsets, b.actions, b.priors };
        b.vfull.resize((size_t)B * KH_VALUE_WIDTH);
        rc = infer_host(e, nullptr, b.boards, B, nullptr, nullptr, b.vfull.data(), nullptr, &l);
    } else {
        b.policy.resize((size_t)B * KH_PSIZE);
        b.vfull.resize((size_t)B * KH_VALUE_WIDTH);
        rc = infer_host(e, b.planes.data(), nullptr, B, b.policy.data(), nullptr, b.vfull.data(), nullptr);
    }
    return false;
}

// One ticket's rows out of a finished records-and-legal-actions batch's page-locked block (per-sample values)
inline void co_fetch(CoTicket* t)
{
    const CoBatch* b = t->from;
    if (t->nact) memcpy(t->priors, b->priors + t->act0, (size_t)t->nact * 4);
    memcpy(t->value, b->values + t->row0, (size_t)t->rows * 4);
}

// the last ticket's rows are out: the buffer can be filled again
void co_release(Coalescer* c, CoBatch* b)
{
    std::lock_guard<SpinLock> lk(c->mu);
    b->tickets.clear();
    b->rows = b->nact = 0; b->full = false; b->state = 0;
    c->cv_space.notify_all();
}

// a ticket in state 4: whoever wins the claim copies its rows out (its waiter, or the dispatcher going round)
inline bool co_claim_fetch(Coalescer* c, CoTicket* t)
{
    int expect = 4;
    if (!t->state.compare_exchange_strong(expect, 5, std::memory_order_acquire)) return false;
    CoBatch* b = t->from;
    co_fetch(t);
    t->state.store(2, std::memory_order_release);
    if (b->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) co_release(c, b);
    return true;
}

// results (or the error) of a finished batch into every caller's own buffers, by the dispatcher alone: the synchronous
// kinds, errors, and the rare per-ticket re-run
void co_deliver(kh_engine* e, CoBatch& b, int rc)
{
    const bool flat = e->cfg.value_mode == KH_VALUE_REFERENCE_FLAT;
    const std::string err = rc ? g_err : std::string();
    for (CoTicket* t : b.tickets) {
        if (rc == KH_ERR_NAN_POLICY || rc == KH_ERR_NAN_VALUE) {
            // a NaN somewhere in the merged batch: the reference's exception belongs to the caller whose rows hold it.
            // Rare: run every ticket of this batch on its own (its inputs are still the caller's to keep until kh_wait).
            if (t->kind == 0) {
                LegalIO l{ t->offsets, t->actions, t->priors };
                t->status = infer_host(e, nullptr, t->boards, t->rows, nullptr, t->value, nullptr, nullptr, &l);
            } else {
                t->status = infer_host(e, t->planes, nullptr, t->rows, t->policy, t->value, nullptr, nullptr);
            }
            t->err = t->status ? g_err : std::string();
            continue;
        }
        t->status = rc; t->err = err;
        if (rc) continue;
        if (t->kind == 0) {
            if (t->nact) memcpy(t->priors, b.priors + t->act0, (size_t)t->nact * 4);
        } else {
            memcpy(t->policy, b.policy.data() + (size_t)t->row0 * KH_PSIZE, (size_t)t->rows * KH_PSIZE * 4);
        }
        if (b.kind == 0 && !flat) memcpy(t->value, b.values + t->row0, (size_t)t->rows * 4);
        else if (flat) memcpy(t->value, b.vfull.data() + (size_t)t->row0 * KH_VALUE_WIDTH, (size_t)t->rows * 4);    // rows <= 256 here
        else for (int i = 0; i < t->rows; ++i) t->value[i] = b.vfull[(size_t)(t->row0 + i) * KH_VALUE_WIDTH];
    }
}

// ONE dispatcher thread per engine: seals a batch when the rule says so, launches it WITHOUT waiting for it, polls the
// launches that are on the device (up to `max_inflight`, each on its buffer's own stream) and hands results back.
// Round 2 had two lanes that each blocked on their launch: two spinning threads of the 16 the search needs, and a batch
// that became ready while both were busy waited out a whole engine call.
// Completion: signal_kernel's word in the batch's page-locked block (hipStreamQuery only as the safety net that notices
// a failed stream).  Hand-back: a finished batch's tickets go to state 4 at once and every waiter copies its OWN rows
// out of the block (a dozen callers in parallel: the block was written over PCIe, every line is a DRAM miss — one
// thread copying 30 KB took 6-8 us of every caller's time); the dispatcher goes round the tickets nobody has claimed
// yet, so a buffer comes back whether or not its callers are waiting.
void co_dispatch(Coalescer* c)
{
    kh_engine* e = c->e;
    CoBatch* fly[CO_BUFFERS];
    CoBatch* drain[CO_BUFFERS];                 // finished batches whose tickets may still be in state 4
    int drain_age[CO_BUFFERS];
    int nfly = 0, ndrain = 0;
    auto us = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::micro>(d).count(); };
    auto complete = [&](CoBatch* b, int rc, bool in_block) {
        const auto t_ran = std::chrono::steady_clock::now();
        CoTicket* mine[KH_MAX_OUTSTANDING];
        const int nt = (int)b->tickets.size();
        for (int i = 0; i < nt; ++i) mine[i] = b->tickets[i];
        const int rows = b->rows;
        if (in_block && rc == KH_OK) {
            b->readers.store(nt, std::memory_order_relaxed);
            for (int i = 0; i < nt; ++i) { mine[i]->status = KH_OK; mine[i]->err.clear(); mine[i]->from = b; }
            for (int i = 0; i < nt; ++i) mine[i]->state.store(4, std::memory_order_release);
        } else {
            co_deliver(e, *b, rc);
            for (int i = 0; i < nt; ++i) mine[i]->state.store(2, std::memory_order_release);
        }
        bool wake;
        {
            std::lock_guard<SpinLock> lk(c->mu);
            c->launches += 1; c->rows_launched += rows;
            wake = c->sleepers > 0;
        }
        if (wake) c->cv_done.notify_all();          // (sleepers re-check their ticket under the lock: states were stored before it)
        if (in_block && rc == KH_OK) {                          // its waiters fetch their rows; the dispatcher sweeps up later
            int i = 0;
            while (i < ndrain && drain[i] != b) ++i;            // (still listed from its previous launch: released since)
            if (i == ndrain) ++ndrain;
            drain[i] = b; drain_age[i] = 0;
        }
        else co_release(c, b);
        if (c->trace) {
            c->us_fill += us(b->t_seal - b->first); c->us_copywait += us(b->t_run - b->t_seal); c->us_launch += us(b->t_launched - b->t_run);
            c->us_run += us(t_ran - b->t_run); c->us_finish += us(std::chrono::steady_clock::now() - t_ran);
        }
    };
    // The batches are looked at (under the queue's lock) only when something can have changed: a submission since the last
    // look, a deadline of an open batch, a launch slot that has come back, a settings change / stop (they bump `submits`
    // too).  Looking every time round made the dispatcher the 15th contender for a lock 14 workers submit through.
    unsigned seen = c->submits.load(std::memory_order_acquire) - 1;
    bool open_any = false, slot_back = false;
    auto next_due = std::chrono::steady_clock::time_point::max();
    for (unsigned spin = 0;; ++spin) {
        CoBatch* take = nullptr;
        const unsigned subs = c->submits.load(std::memory_order_acquire);
        if (subs != seen || slot_back || (open_any && std::chrono::steady_clock::now() >= next_due) ||
            (!open_any && nfly == 0 && ndrain == 0)) {
            std::unique_lock<SpinLock> lk(c->mu);
            seen = c->submits.load(std::memory_order_acquire);
            slot_back = false;
            open_any = false;
            next_due = std::chrono::steady_clock::time_point::max();
            const int target = e->co_target.load(), wait_us = e->co_wait_us.load(), callers = e->co_callers.load();
            for (auto& b : c->batches) {
                if (b.state != 1 || b.rows == 0) continue;
                open_any = true;
                if (nfly >= c->max_inflight) { next_due = std::chrono::steady_clock::time_point::max(); break; }   // (a slot coming back re-opens the question)
                // immediate mode (no target): whatever has accumulated goes at once;
                // target mode: wait for `target` rows or `callers` submissions — but no longer than wait_us after the
                // batch's first submission, and not once the burst of submissions has ended (nothing added for
                // wait_us / 8: callers that keep a fixed number of positions in flight rarely hit the target exactly —
                // terminal leaves need no evaluation)
                bool ready = b.full || target <= 0 || b.rows >= target || (callers > 0 && (int)b.tickets.size() >= callers);
                if (!ready) {
                    const auto due = std::min(b.first + std::chrono::microseconds(wait_us), b.last + std::chrono::microseconds(wait_us / 8 + 1));
                    ready = std::chrono::steady_clock::now() >= due;
                    if (!ready) next_due = std::min(next_due, due);
                }
                if (ready) { take = &b; break; }
            }
            if (!take && !open_any && nfly == 0 && ndrain == 0) {
                if (c->stop) return;
                c->asleep.store(true, std::memory_order_release);
                c->cv_lane.wait(lk);                 // nothing queued, nothing on the device
                c->asleep.store(false, std::memory_order_release);
                seen = c->submits.load(std::memory_order_acquire) - 1;
                continue;
            }
            if (take) { take->state = 2; take->t_seal = std::chrono::steady_clock::now(); seen = subs - 1; }   // (look again: another batch may be ready)
        }
        if (take) {
            while (take->copying.load(std::memory_order_acquire) > 0) __builtin_ia32_pause();   // submitters still copying their rows in: a microsecond
            take->t_run = std::chrono::steady_clock::now();
            int rc = KH_OK;
            int li = 0;
            while (c->lane_busy[li]) ++li;              // nfly < max_inflight <= MAX_LANES: one is free
            take->lane = &c->lanes[li];
            const bool on_device = co_start(e, *take, rc);
            if (on_device) c->lane_busy[li] = true;
            take->t_launched = std::chrono::steady_clock::now();
            if (on_device) fly[nfly++] = take;
            else complete(take, rc, false);
        }
        for (int i = 0; i < nfly;) {
            CoBatch* b = fly[i];
            bool done = __atomic_load_n(b->done, __ATOMIC_ACQUIRE) == b->serial;
            hipError_t q = hipSuccess;
            if (!done && (spin & 4095) == 4095) {         // safety net: a stream that failed never writes the word
                q = hipStreamQuery(b->lane->stream);
                done = q != hipErrorNotReady;
            }
            if (!done) { ++i; continue; }
            const int rc = q == hipSuccess ? co_finish_legal(*b) : fail(KH_ERR_HIP, "hipStreamQuery failed: %s", hipGetErrorString(q));
            b->W.reset();
            c->lane_busy[b->lane - c->lanes] = false;
            complete(b, rc, true);
            fly[i] = fly[--nfly];
            slot_back = true;
        }
        // Finished batches whose rows have not all been fetched: their waiters do that themselves, in parallel, the
        // moment they see state 4 — the dispatcher only sweeps up what is left after a few rounds' grace (callers that
        // are busy with another set, or never wait), so that a buffer always comes back.  A buffer is in the list at
        // most once (a second publish needs a release in between), so the list never outgrows the buffers.
        for (int i = 0; i < ndrain;) {
            CoBatch* b = drain[i];
            if (b->readers.load(std::memory_order_acquire) > 0 && (take || ++drain_age[i] <= 16)) { ++i; continue; }
            if (b->readers.load(std::memory_order_acquire) > 0) {
                CoTicket* mine[KH_MAX_OUTSTANDING];
                int nt = 0;
                { std::lock_guard<SpinLock> lk(c->mu); if (b->state == 2) for (CoTicket* t : b->tickets) mine[nt++] = t; }
                for (int k = 0; k < nt; ++k) if (mine[k]->from == b) (void)co_claim_fetch(c, mine[k]);
            }
            --ndrain;
            drain[i] = drain[ndrain]; drain_age[i] = drain_age[ndrain];
        }
        if (!take) {
            // a batch is filling or a launch is on the device: poll (a sleeping thread's wake-up, 5-15 us, would be
            // paid by every caller of the launch), but let a caller's thread have the core when it needs one
            for (int k = 0; k < 8; ++k) __builtin_ia32_pause();
            if ((spin & 15) == 15) sched_yield();
        }
    }
}

Coalescer* co_get(kh_engine* e)
{
    std::lock_guard<std::mutex> lk(e->co_mu);
    if (!e->co) {
        Coalescer* c = new Coalescer();
        c->e = e;
        c->trace = getenv("KAMI_CO_TRACE") != nullptr;
        if (getenv("KAMI_WAIT_SPIN_US")) c->spin_us = std::max(0, atoi(getenv("KAMI_WAIT_SPIN_US")));
        if (getenv("KAMI_CO_INFLIGHT")) c->max_inflight = std::min((int)Coalescer::MAX_LANES, std::max(1, atoi(getenv("KAMI_CO_INFLIGHT"))));
        // the lanes' streams now, one after the other (see Coalescer::lanes); a failure here shows up at the first launch
        if (set_device(e) == KH_OK)
            for (auto& l : c->lanes) (void)hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking);
        c->dispatcher = std::thread(co_dispatch, c);
        e->co = c;
        e->co_ready.store(c, std::memory_order_release);
    }
    return e->co;
}

void co_destroy(kh_engine* e)
{
    Coalescer* c = e->co;
    if (!c) return;
    { std::lock_guard<SpinLock> lk(c->mu); c->stop = true; c->submits.fetch_add(1); }
    c->cv_lane.notify_all();
    if (c->dispatcher.joinable()) c->dispatcher.join();
    (void)hipSetDevice(e->cfg.device);
    for (auto& l : c->lanes) {
        if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); l.stream = nullptr; }
    }
    if (c->trace && c->launches)
        fprintf(stderr, "[kami queue] %lld launches, %.1f rows each; per launch: filling %.1f us, waiting for copies %.1f us, "
                "engine call %.1f us (of which the launch call %.1f), hand-back %.1f us\n", (long long)c->launches, (double)c->rows_launched / c->launches,
                c->us_fill / c->launches, c->us_copywait / c->launches, c->us_run / c->launches, c->us_launch / c->launches, c->us_finish / c->launches);
    delete c;
    e->co = nullptr;
    e->co_ready.store(nullptr, std::memory_order_release);
}

int co_submit(kh_engine* e, int kind, const kh_board* boards, const float* planes, int batch, const int32_t* offsets,
              const int32_t* actions, float* priors, float* value, float* policy, int64_t* ticket)
{
    if (!e || !ticket || !value || batch < 1) return fail(KH_ERR_INVALID, "bad submit arguments");
    int nact = 0;
    if (kind == 0) {
        if (!boards || !offsets || !actions || !priors) return fail(KH_ERR_INVALID, "null buffer");
        if (e->cfg.features != KH_NFEATURES) return fail(KH_ERR_INVALID, "compact records need features == %d", KH_NFEATURES);
        if (offsets[0] != 0) return fail(KH_ERR_INVALID, "action_offsets[0] must be 0");
        for (int i = 0; i < batch; ++i)
            if (offsets[i + 1] < offsets[i]) return fail(KH_ERR_INVALID, "action_offsets must be non-decreasing");
        nact = offsets[batch];
        if (batch > CO_SMALL_LEGAL || nact > CO_ACTS) return fail(KH_ERR_INVALID, "submissions hold at most %d positions / %d actions (use the synchronous call for more)", CO_SMALL_LEGAL, CO_ACTS);
    } else {
        if (!planes || !policy) return fail(KH_ERR_INVALID, "null buffer");
        if (batch > CO_SMALL_PLANES) return fail(KH_ERR_INVALID, "plane submissions hold at most %d positions (use kh_infer for more)", CO_SMALL_PLANES);
    }
    if (!e->has_weights.load(std::memory_order_acquire)) return fail(KH_ERR_NO_WEIGHTS, "submit before kh_load_weights");   // (no shared_ptr copy under wmu per submission)
    Coalescer* c = e->co_ready.load(std::memory_order_acquire);
    if (!c) c = co_get(e);
    const size_t F = e->cfg.features;
    const int cap_rows = kind == 0 ? CO_ROWS : 2 * CO_SMALL_PLANES;
    std::unique_lock<SpinLock> lk(c->mu);
    static_assert(KH_MAX_OUTSTANDING == 64, "free_mask is one 64-bit word");
    const uint64_t fm = c->free_mask.load(std::memory_order_acquire);
    if (fm == 0) return fail(KH_ERR_INVALID, "%d submissions are outstanding on this engine: kh_wait for some before submitting more", KH_MAX_OUTSTANDING);
    const int tid = __builtin_ctzll(fm);               // taken NOW: the wait for a free buffer below drops the lock
    c->free_mask.fetch_and(~(1ull << tid), std::memory_order_acq_rel);
    CoTicket* t = &c->tickets[tid];
    t->state.store(1, std::memory_order_relaxed);
    CoBatch* b = nullptr;
    for (;;) {
        for (auto& x : c->batches)
            if (x.state == 1 && x.kind == kind && !x.full) {
                if (x.rows + batch <= cap_rows && x.nact + nact <= CO_ACTS) { b = &x; break; }
                x.full = true;                                   // does not fit: it goes as it is
                c->submits.fetch_add(1, std::memory_order_release);
            }
        if (b) break;
        for (auto& x : c->batches)
            if (x.state == 0) { b = &x; break; }
        if (b) {
            b->state = 1; b->kind = kind; b->rows = 0; b->nact = 0; b->full = false;
            b->first = b->last = std::chrono::steady_clock::now();
            if (kind == 0) {
                if (!b->boards) {
                    const size_t o_offs = (size_t)CO_ROWS * sizeof(kh_board), o_acts = o_offs + (((size_t)CO_ROWS + 1) * 4 + 15) / 16 * 16;
                    const size_t o_vals = (size_t)CO_ACTS * 4, o_flags = o_vals + (size_t)CO_ROWS * 4;
                    if (set_device(e) || b->pin_in.ensure(o_acts + (size_t)CO_ACTS * 4) || b->pin_out.ensure(o_flags + 128)) {
                        b->state = 0;
                        t->state.store(0, std::memory_order_release);
                        c->free_mask.fetch_or(1ull << tid, std::memory_order_release);
                        return KH_ERR_HIP;
                    }
                    b->boards = reinterpret_cast<kh_board*>(b->pin_in.at(0));
                    b->offsets = reinterpret_cast<int32_t*>(b->pin_in.at(o_offs));
                    b->actions = reinterpret_cast<int32_t*>(b->pin_in.at(o_acts));
                    b->priors = reinterpret_cast<float*>(b->pin_out.at(0));
                    b->values = reinterpret_cast<float*>(b->pin_out.at(o_vals));
                    b->flags_out = reinterpret_cast<int*>(b->pin_out.at(o_flags));
                    b->done = reinterpret_cast<unsigned*>(b->pin_out.at(o_flags + 64));       // a cache line of its own
                    *b->done = 0;
                }
                b->offsets[0] = 0;
            } else if (b->planes.size() < (size_t)cap_rows * 64 * F) b->planes.resize((size_t)cap_rows * 64 * F);
            break;
        }
        c->cv_space.wait(lk);                                    // every buffer is on the device: one of them comes back
    }
    // under the lock: only what the batch's bookkeeping needs; the ticket's own fields are written after it (the dispatcher
    // reads them when the launch completes, which is behind `copying` reaching zero)
    const int row0 = b->rows, act0 = b->nact;
    b->rows += batch; b->nact += nact;
    b->tickets.push_back(t);
    b->copying.fetch_add(1, std::memory_order_relaxed);
    b->last = std::chrono::steady_clock::now();           // (under the lock: the dispatcher reads it there)
    c->submits.fetch_add(1, std::memory_order_release);
    lk.unlock();
    t->status = KH_OK; t->err.clear(); ++t->serial;
    t->kind = kind; t->row0 = row0; t->rows = batch; t->act0 = act0; t->nact = nact;
    t->boards = boards; t->planes = planes; t->offsets = offsets; t->actions = actions;
    t->priors = priors; t->value = value; t->policy = policy;
    // this caller's rows into the merge buffers (every caller copies its own, in parallel)
    if (kind == 0) {
        memcpy(b->boards + t->row0, boards, (size_t)batch * sizeof(kh_board));
        if (nact) memcpy(b->actions + t->act0, actions, (size_t)nact * 4);
        for (int i = 1; i <= batch; ++i) b->offsets[t->row0 + i] = t->act0 + offsets[i];
    } else {
        memcpy(b->planes.data() + (size_t)t->row0 * 64 * F, planes, (size_t)batch * 64 * F * 4);
    }
    const uint32_t serial = t->serial;
    b->copying.fetch_sub(1, std::memory_order_release);   // (no lock: the dispatcher spins on it once it has sealed this batch; after
                                                          //  this the launch may complete and the ticket be waited for)
    if (c->asleep.load(std::memory_order_acquire)) {      // (a notify per submission was a std::mutex every caller met at once)
        std::lock_guard<SpinLock> lk2(c->mu);
        c->cv_lane.notify_all();
    }
    *ticket = (int64_t)tid | ((int64_t)serial << 32);
    return KH_OK;
}

int co_wait(kh_engine* e, int64_t ticket)
{
    if (!e || !e->co) return fail(KH_ERR_INVALID, "no such ticket");
    Coalescer* c = e->co;
    const int tid = (int)(ticket & 0xffffffff);
    const uint32_t serial = (uint32_t)(ticket >> 32);
    if (tid < 0 || tid >= KH_MAX_OUTSTANDING) return fail(KH_ERR_INVALID, "no such ticket");
    CoTicket& t = c->tickets[tid];
    if (t.state.load(std::memory_order_acquire) == 0 || t.serial != serial)
        return fail(KH_ERR_INVALID, "ticket already waited for (or never issued)");
    // a launch is ~100 us away at most: spin on the ticket first (no wake-up latency, no mutex convoy when a launch
    // releases many callers at once), sleep on the condition variable only when it takes longer.  State 4: the rows are
    // in the batch's block and this thread fetches them itself (unless the dispatcher got there first: state 5, then 2).
    auto settled = [&] {
        const int st = t.state.load(std::memory_order_acquire);
        if (st == 2) return true;
        if (st == 4) (void)co_claim_fetch(c, &t);
        return t.state.load(std::memory_order_acquire) == 2;
    };
    if (!settled() && c->spin_us > 0) {
        const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(c->spin_us);
        for (int k = 0; !settled(); ++k) {
            __builtin_ia32_pause();
            if ((k & 63) == 63) {
                sched_yield();                  // lets the dispatcher (or another caller) have the core if it needs one
                if (std::chrono::steady_clock::now() >= until) break;
            }
        }
    }
    if (!settled()) {
        std::unique_lock<SpinLock> lk(c->mu);
        ++c->sleepers;
        for (;;) {
            const int st = t.state.load(std::memory_order_acquire);
            if (st == 2) break;
            if (st == 4 || st == 5) { lk.unlock(); while (!settled()) __builtin_ia32_pause(); lk.lock(); break; }
            c->cv_done.wait(lk);
        }
        --c->sleepers;
    }
    const int rc = t.status;
    if (rc) g_err = t.err;
    t.state.store(0, std::memory_order_release);
    c->free_mask.fetch_or(1ull << tid, std::memory_order_release);
    return rc;
}

// a synchronous small call goes through the queue when other small calls are inside the engine right now
struct SmallCall {
    kh_engine* e;
    bool others;
    explicit SmallCall(kh_engine* e_) : e(e_) { others = e->small_calls.fetch_add(1) > 0; }
    ~SmallCall() { e->small_calls.fetch_sub(1); }
};

}  // namespace

// =============================================================================== C ABI
extern "C" {

const char* kh_last_error(void) { return g_err.c_str(); }
const char* kh_version(void) { return "kamihip 0.1 gfx950"; }

int kh_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

size_t kh_weight_count(int F, int C, int R)
{
    size_t n = 0;
    n += (size_t)C * F * 9 + C + 4 * (size_t)C;
    n += (size_t)R * 2 * ((size_t)C * C * 9 + C + 4 * (size_t)C);
    n += (size_t)KH_POLICY_MID * C + KH_POLICY_MID + 4 * KH_POLICY_MID;
    n += (size_t)KH_POLICY_PLANES * KH_POLICY_MID + KH_POLICY_PLANES;
    n += (size_t)C + 1 + 4;
    n += (size_t)KH_VALUE_WIDTH * 64 + KH_VALUE_WIDTH;
    return n;
}

int kh_create(const kh_config* cfg, kh_engine** out)
{
    if (!out) return fail(KH_ERR_INVALID, "null out");
    *out = nullptr;
    int rc = check_cfg(cfg);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(KH_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU path");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(KH_ERR_INVALID, "device %d out of range (%d visible)", cfg->device, ndev);
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, cfg->device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(KH_ERR_NO_DEVICE, "device %d is %s; this library is built for gfx950 only",
                    cfg->device, prop.gcnArchName);
    kh_engine* e = new kh_engine();
    e->cfg = *cfg;
    e->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    e->f32_simple = getenv("KAMI_F32_SIMPLE") && atoi(getenv("KAMI_F32_SIMPLE")) != 0;
    if (getenv("KAMI_SMALL_MAX")) e->small_max = std::max(0, atoi(getenv("KAMI_SMALL_MAX")));
    *out = e;
    return KH_OK;
}

void kh_destroy(kh_engine* e)
{
    if (!e) return;
    co_destroy(e);
    (void)hipSetDevice(e->cfg.device);
    auto kill = [](Slot* s) {
        if (s && s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
        if (s && s->stream2) { (void)hipStreamSynchronize(s->stream2); (void)hipStreamDestroy(s->stream2); }
        if (s && s->scratch_done) (void)hipEventDestroy(s->scratch_done);
    };
    for (auto& s : e->slots) kill(s.get());
    kill(e->devslot.get());
    for (auto& r : e->pinned) (void)hipHostUnregister(const_cast<char*>(r.first));
    delete e->train;
    delete e;
}

static int load_weights_impl(kh_engine* e, const float* blob, size_t nfloats, int generation, std::shared_ptr<Weights>* installed)
{
    if (!e || !blob) return fail(KH_ERR_INVALID, "null argument");
    const int F = e->cfg.features, C = e->cfg.filters, R = e->cfg.residuals;
    if (nfloats != kh_weight_count(F, C, R))
        return fail(KH_ERR_INVALID, "weight blob has %zu floats, expected %zu for F=%d C=%d R=%d",
                    nfloats, kh_weight_count(F, C, R), F, C, R);
    int rc = set_device(e);
    if (rc) return rc;
    auto W = std::make_shared<Weights>();
    W->generation = generation;
    W->blob.assign(blob, blob + nfloats);
    HostNet n = parse_blob(W->blob.data(), F, C, R);
    if (e->cfg.dtype == KH_F32) {
        if ((rc = build_simple(*W, n, F, C, R))) return rc;
        if (!e->f32_simple && (rc = build_layers(*W, n, KH_F32, F, C, R))) return rc;
    } else {
        if ((rc = build_tower(*W, n, e->cfg.dtype, F, C, R))) return rc;
        if (!W->tw_ok && (rc = build_layers(*W, n, e->cfg.dtype, F, C, R))) return rc;
    }
    std::lock_guard<std::mutex> lk(e->wmu);
    e->weights = W;                  // calls in flight keep their own reference
    e->has_weights.store(true, std::memory_order_release);
    if (installed) *installed = W;
    return KH_OK;
}

int kh_load_weights(kh_engine* e, const float* blob, size_t nfloats, int generation)
{
    return load_weights_impl(e, blob, nfloats, generation, nullptr);
}

int kh_train(kh_engine* e, const float* inputs, const float* obs_p, const float* obs_v, int trajectories,
             const kh_train_config* cfg, float* first_loss, float* last_loss)
{
    if (!e || !inputs || !obs_p || !obs_v || !cfg) return fail(KH_ERR_INVALID, "null argument");
    if (trajectories < 1 || cfg->batch < 2 || cfg->epochs < 1) return fail(KH_ERR_INVALID, "trajectories >= 1, batch >= 2, epochs >= 1 required");
    const auto t_call = std::chrono::steady_clock::now();
    auto W = current_weights(e);
    if (!W) return fail(KH_ERR_NO_WEIGHTS, "kh_train before kh_load_weights");
    int rc = set_device(e);
    if (rc) return rc;
    const int F = e->cfg.features, C = e->cfg.filters, R = e->cfg.residuals, B = cfg->batch;
    std::unique_ptr<kh::TrainNet, void (*)(kh::TrainNet*)> net(kh::train_layout_new(F, C, R), kh::train_layout_free);
    const size_t nfl = W->blob.size();
    std::lock_guard<std::mutex> train_lock(e->train_mu);      // one trainer per engine at a time (the reference: exclusive lock, nn.cpp:226)
    if (!e->train) e->train = new TrainCache();
    TrainCache& tc = *e->train;
    DevMem &params = tc.params, &grads = tc.grads, &work = tc.work, &dx = tc.dx, &dp = tc.dp, &dv = tc.dv, &dloss = tc.dloss;
    const bool valu_now = getenv("KAMI_TRAIN_VALU") && atoi(getenv("KAMI_TRAIN_VALU")) != 0;
    {
        const void* before[3] = { params.p, work.p, dx.p };
        if ((rc = params.ensure(nfl * 4)) || (rc = grads.ensure(nfl * 4)) || (rc = work.ensure(kh::train_workspace_floats(F, C, R, B) * 4)) ||
            (rc = dx.ensure((size_t)B * 64 * F * 4)) || (rc = dp.ensure((size_t)B * KH_PSIZE * 4)) || (rc = dv.ensure((size_t)B * 4)) ||
            (rc = dloss.ensure((size_t)B * 2 * 4 + 8)))
            return rc;
        // the recorded step holds buffer addresses, the batch size, the learning rate and the kernel choice
        if (before[0] != params.p || before[1] != work.p || before[2] != dx.p || tc.B != B || tc.lr != cfg->lr || tc.valu != valu_now) tc.drop_graph();
        if (before[0] != params.p) tc.on_device.reset();
        tc.B = B; tc.lr = cfg->lr; tc.valu = valu_now;
    }
    const auto t_bufs = std::chrono::steady_clock::now();
    if (!tc.st) HIPCHK(hipStreamCreateWithFlags(&tc.st, hipStreamNonBlocking));
    hipStream_t st = tc.st;
    if (tc.on_device.lock() != W) {
        // through a page-locked block of the trainer's own (the blob is a std::vector)
        if (tc.pin_params.ensure(nfl * 4)) return KH_ERR_HIP;
        memcpy(tc.pin_params.p, W->blob.data(), nfl * 4);
        HIPCHK(hipMemcpyAsync(params.p, tc.pin_params.p, nfl * 4, hipMemcpyHostToDevice, st));
    }                                            // else: `params` still holds exactly these weights — the previous call trained them
    tc.on_device.reset();                        // (until this call has installed its result, `params` belongs to nobody)
    const auto t_up = std::chrono::steady_clock::now();

    // nn.cpp:245-262: one engine for the whole call, one shuffle per epoch; staging rows persist
    std::vector<int> picker((size_t)trajectories);
    for (int i = 0; i < trajectories; ++i) picker[i] = i;
    auto rng = std::default_random_engine{};
    const size_t in_row = (size_t)64 * F;
    // batch staging in page-locked memory (rows persist from batch to batch like the reference's stack buffers)
    PinMem& pin = tc.pin;
    const size_t n_in = (size_t)B * in_row, n_p = (size_t)B * KH_PSIZE;
    if (pin.ensure((n_in + n_p + (size_t)B + (size_t)B * 2 + 2) * 4)) return KH_ERR_HIP;
    float* next_input = reinterpret_cast<float*>(pin.p);
    float* next_policy = next_input + n_in;
    float* next_value = next_policy + n_p;
    float* loss_rows = next_value + B;
    memset(pin.p, 0, (n_in + n_p + (size_t)B + (size_t)B * 2 + 2) * 4);
    float firstloss = 0.0f, lastloss = 0.0f;
    const kh::StepBuffers sb{ params.as<float>(), grads.as<float>(), work.as<float>() };
    // A step is ~120 small launches on fixed buffers: recorded once as a graph, replayed per batch (with the
    // tiled conv kernels the host's launch work, not the GPU, bounded a step).  Falls back to plain launches.
    TrainCache& graph = tc;
    bool& graph_tried = tc.graph_tried;
    HIPCHK(kh::conv_f32_raw_prepare());          // function attributes are not stream work: set them before any capture
    static const bool trace = getenv("KAMI_TRAIN_TRACE") != nullptr;
    const auto t_setup = std::chrono::steady_clock::now();
    for (int epoch = 0; epoch < cfg->epochs; ++epoch) {
        std::shuffle(picker.begin(), picker.end(), rng);
        float avgloss = 0.0f;
        int nbatches = 0;
        for (int base = 0; base <= trajectories - 1;) {
            int i = 0;
            for (; i < B && i + base <= trajectories - 1; ++i) {
                const size_t src = (size_t)picker[base + i];
                memcpy(&next_input[(size_t)i * in_row], inputs + src * in_row, in_row * 4);
                memcpy(&next_policy[(size_t)i * KH_PSIZE], obs_p + src * KH_PSIZE, (size_t)KH_PSIZE * 4);
                next_value[i] = obs_v[src];
            }
            base += i;
            if (cfg->detect_anomaly) {                        // nn.cpp:329-333: isnan on the batch as it goes to the device
                for (size_t k = 0; k < n_in; ++k)
                    if (next_input[k] != next_input[k]) return fail(KH_ERR_INVALID, "training input ind %d contains NaN", nbatches);
            }
            HIPCHK(hipMemcpyAsync(dx.p, next_input, n_in * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(dp.p, next_policy, n_p * 4, hipMemcpyHostToDevice, st));
            HIPCHK(hipMemcpyAsync(dv.p, next_value, (size_t)B * 4, hipMemcpyHostToDevice, st));
            static const bool no_graph = getenv("KAMI_TRAIN_NOGRAPH") != nullptr;
            if (!graph_tried && no_graph) graph_tried = true;
            if (!graph_tried) {
                graph_tried = true;
                if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
                    const hipError_t ce = kh::train_step(*net, sb, dx.as<float>(), dp.as<float>(), dv.as<float>(), B, cfg->lr, dloss.as<float>(), st);
                    const hipError_t ee = hipStreamEndCapture(st, &graph.g);
                    if (ce != hipSuccess || ee != hipSuccess || hipGraphInstantiate(&graph.x, graph.g, nullptr, nullptr, 0) != hipSuccess) {
                        if (graph.x) { (void)hipGraphExecDestroy(graph.x); graph.x = nullptr; }
                        (void)hipGetLastError();
                    }
                    if (trace) fprintf(stderr, "[kami train] step recorded as a graph: %s\n", graph.x ? "yes" : "NO (plain launches)");
                }
            }
            if (graph.x) HIPCHK(hipGraphLaunch(graph.x, st));
            else HIPCHK(kh::train_step(*net, sb, dx.as<float>(), dp.as<float>(), dv.as<float>(), B, cfg->lr, dloss.as<float>(), st));
            HIPCHK(hipMemcpyAsync(loss_rows, dloss.p, (size_t)B * 2 * 4 + 8, hipMemcpyDeviceToHost, st));
            HIPCHK(hipStreamSynchronize(st));
            if (cfg->detect_anomaly) {                        // nn.cpp:337-341: the value output first, then the policy
                const int* nf = reinterpret_cast<const int*>(loss_rows + 2 * B);
                if (nf[1]) return fail(KH_ERR_NAN_VALUE, "forward value output contains NaN");
                if (nf[0]) return fail(KH_ERR_NAN_POLICY, "forward policy output contains NaN");
            }
            float lp = 0.0f, lv = 0.0f;
            for (int b = 0; b < B; ++b) { lp += loss_rows[b]; lv += loss_rows[B + b]; }
            const float loss = lp + lv / (float)(B * KH_VALUE_WIDTH);
            if (loss != loss) return fail(KH_ERR_NAN_POLICY, "training loss is NaN (epoch %d, batch %d)", epoch, nbatches);
            avgloss += loss;
            ++nbatches;
        }
        avgloss /= (float)nbatches;
        if (!epoch) firstloss = avgloss;
        lastloss = avgloss;
    }
    const auto t_steps = std::chrono::steady_clock::now();
    std::vector<float> blob(nfl);
    HIPCHK(hipMemcpy(blob.data(), params.p, nfl * 4, hipMemcpyDeviceToHost));
    if (first_loss) *first_loss = firstloss;
    if (last_loss) *last_loss = lastloss;
    const auto t_read = std::chrono::steady_clock::now();
    std::shared_ptr<Weights> installed;
    const int lrc = load_weights_impl(e, blob.data(), nfl, W->generation + 1, &installed);         // nn.cpp:371 ++generation
    if (lrc == KH_OK) tc.on_device = installed;
    if (trace) {
        auto ms = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::milli>(d).count(); };
        fprintf(stderr, "[kami train] call: set-up %.2f ms (buffers %.2f, parameters up %.2f, staging %.2f), steps %.2f ms, parameters back %.2f ms, "
                "weights installed %.2f ms\n", ms(t_setup - t_call), ms(t_bufs - t_call), ms(t_up - t_bufs), ms(t_setup - t_up),
                ms(t_steps - t_setup), ms(t_read - t_steps), ms(std::chrono::steady_clock::now() - t_read));
    }
    return lrc;
}

int kh_train_order(int trajectories, int epochs, int32_t* order)
{
    if (trajectories < 1 || epochs < 1 || !order) return fail(KH_ERR_INVALID, "trajectories >= 1, epochs >= 1 and a buffer required");
    std::vector<int> picker((size_t)trajectories);
    for (int i = 0; i < trajectories; ++i) picker[i] = i;
    auto rng = std::default_random_engine{};                  // exactly kh_train's
    for (int epoch = 0; epoch < epochs; ++epoch) {
        std::shuffle(picker.begin(), picker.end(), rng);
        for (int i = 0; i < trajectories; ++i) order[(size_t)epoch * trajectories + i] = picker[i];
    }
    return KH_OK;
}

int kh_checkpoint_read(const char* path, int* features, int* filters, int* residuals, int* generation,
                       float* blob, size_t cap, size_t* nfloats)
{
    if (!path) return fail(KH_ERR_INVALID, "null path");
    FILE* f = fopen(path, "rb");
    if (!f) return fail(KH_ERR_INVALID, "cannot open %s", path);
    int32_t hdr[8] = { 0 };
    const size_t got = fread(hdr, 1, sizeof hdr, f);
    int F = 0, C = 0, R = 0, gen = 0;
    std::vector<float> data;
    if (got == sizeof hdr && hdr[0] == 0x574d414b /* "KAMW" */) {
        F = hdr[1]; C = hdr[2]; R = hdr[3]; gen = hdr[4];
        if (F < 1 || F > 4096 || C < 1 || C > 1024 || R < 0 || R > 256) { fclose(f); return fail(KH_ERR_INVALID, "%s: bad header", path); }
        // the header's numbers come from the file: size the buffer only once the file is known to hold that many floats
        const size_t need = kh_weight_count(F, C, R);
        fseek(f, 0, SEEK_END);
        const long fsz = ftell(f);
        fseek(f, (long)sizeof hdr, SEEK_SET);
        if (fsz < 0 || ((size_t)fsz - sizeof hdr) / 4 < need) { fclose(f); return fail(KH_ERR_INVALID, "%s: truncated", path); }
        data.resize(need);
        const size_t n = fread(data.data(), 4, data.size(), f);
        fclose(f);
        if (n != data.size()) return fail(KH_ERR_INVALID, "%s: truncated", path);
    } else {
        fclose(f);
        if (got < 4 || memcmp(hdr, "PK\003\004", 4) != 0)
            return fail(KH_ERR_INVALID, "%s is neither a libtorch archive nor an engine weight blob", path);
        try {
            kh_archive::Checkpoint ck = kh_archive::read_checkpoint(path);
            auto w = ck.tensors.find("conv1.weight");
            auto g = ck.ints.find("generation");
            if (w == ck.tensors.end() || w->second.shape.size() != 4 || g == ck.ints.end())
                return fail(KH_ERR_INVALID, "%s: not a kami checkpoint (no conv1.weight / generation)", path);
            // the same bounds as the KAMW path, on numbers that come from the file; every tensor below must have EXACTLY
            // the shape libtorch gives it (nn.cpp:20-23,45-56) — an element count alone would take a transposed tensor
            const int64_t c64 = w->second.shape[0], f64 = w->second.shape[1];
            if (f64 < 1 || f64 > 4096 || c64 < 1 || c64 > 1024) return fail(KH_ERR_INVALID, "%s: conv1.weight has an impossible shape", path);
            if (g->second < INT32_MIN || g->second > INT32_MAX) return fail(KH_ERR_INVALID, "%s: generation out of range", path);
            C = (int)c64; F = (int)f64; gen = (int)g->second;
            while (R <= 256 && ck.tensors.count("residual" + std::to_string(R) + ".conv1.weight")) ++R;
            if (R > 256) return fail(KH_ERR_INVALID, "%s: more than 256 residual blocks", path);
            // blob order of kh_weight_count (names per nn.cpp:20-23,45-56)
            std::vector<std::pair<std::string, std::vector<int64_t>>> order;
            auto convbn = [&](const std::string& conv, const std::string& bn, int64_t co, int64_t ci, int64_t k) {
                order.emplace_back(conv + ".weight", std::vector<int64_t>{ co, ci, k, k }); order.emplace_back(conv + ".bias", std::vector<int64_t>{ co });
                for (const char* s : { ".weight", ".bias", ".running_mean", ".running_var" }) order.emplace_back(bn + s, std::vector<int64_t>{ co });
            };
            convbn("conv1", "batchnorm1", C, F, 3);
            for (int i = 0; i < R; ++i) {
                const std::string r = "residual" + std::to_string(i);
                convbn(r + ".conv1", r + ".batchnorm1", C, C, 3);
                convbn(r + ".conv2", r + ".batchnorm2", C, C, 3);
            }
            convbn("policyconv", "pbatchnorm", KH_POLICY_MID, C, 1);
            order.emplace_back("policyconv2.weight", std::vector<int64_t>{ KH_POLICY_PLANES, KH_POLICY_MID, 1, 1 });
            order.emplace_back("policyconv2.bias", std::vector<int64_t>{ KH_POLICY_PLANES });
            convbn("valueconv", "vbatchnorm", 1, C, 1);
            order.emplace_back("valuefc.weight", std::vector<int64_t>{ KH_VALUE_WIDTH, 64 });
            order.emplace_back("valuefc.bias", std::vector<int64_t>{ KH_VALUE_WIDTH });
            if (order.size() != ck.tensors.size())
                return fail(KH_ERR_INVALID, "%s: %zu tensors, this network has %zu", path, ck.tensors.size(), order.size());
            for (auto& o : order) {
                auto t = ck.tensors.find(o.first);
                if (t == ck.tensors.end() || t->second.shape != o.second)
                    return fail(KH_ERR_INVALID, "%s: tensor %s missing or of the wrong shape", path, o.first.c_str());
                data.insert(data.end(), t->second.data.begin(), t->second.data.end());
            }
        } catch (const std::exception& ex) {
            return fail(KH_ERR_INVALID, "%s: %s", path, ex.what());
        }
    }
    if (features) *features = F;
    if (filters) *filters = C;
    if (residuals) *residuals = R;
    if (generation) *generation = gen;
    if (nfloats) *nfloats = data.size();
    if (blob) {
        if (cap < data.size()) return fail(KH_ERR_INVALID, "blob buffer holds %zu floats, the checkpoint has %zu", cap, data.size());
        memcpy(blob, data.data(), data.size() * sizeof(float));
    }
    return KH_OK;
}

int kh_load_checkpoint(kh_engine* e, const char* path)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    int F, C, R, gen;
    size_t n = 0;
    int rc = kh_checkpoint_read(path, &F, &C, &R, &gen, nullptr, 0, &n);
    if (rc) return rc;
    if (F != e->cfg.features || C != e->cfg.filters || R != e->cfg.residuals)
        return fail(KH_ERR_INVALID, "%s holds a %d-plane %dx%d network, this engine is %d-plane %dx%d", path, F, R, C,
                    e->cfg.features, e->cfg.residuals, e->cfg.filters);
    std::vector<float> blob(n);
    if ((rc = kh_checkpoint_read(path, nullptr, nullptr, nullptr, nullptr, blob.data(), blob.size(), nullptr))) return rc;
    return kh_load_weights(e, blob.data(), blob.size(), gen);
}

int kh_get_weights(kh_engine* e, float* blob, size_t nfloats)
{
    if (!e || !blob) return fail(KH_ERR_INVALID, "null argument");
    auto W = current_weights(e);
    if (!W) return fail(KH_ERR_NO_WEIGHTS, "no weights loaded");
    if (nfloats != W->blob.size()) return fail(KH_ERR_INVALID, "blob has %zu floats, caller asked for %zu", W->blob.size(), nfloats);
    memcpy(blob, W->blob.data(), nfloats * sizeof(float));
    return KH_OK;
}

int kh_generation(kh_engine* e)
{
    if (!e) return -1;
    auto W = current_weights(e);
    return W ? W->generation : 0;
}

int kh_clone(kh_engine* src, kh_engine** out)
{
    if (!src || !out) return fail(KH_ERR_INVALID, "null argument");
    int rc = kh_create(&src->cfg, out);
    if (rc) return rc;
    auto W = current_weights(src);
    if (W && (rc = kh_load_weights(*out, W->blob.data(), W->blob.size(), W->generation))) {
        kh_destroy(*out);
        *out = nullptr;
        return rc;
    }
    return KH_OK;
}

int kh_infer(kh_engine* e, const float* input, int batch, float* policy, float* value)
{
    if (!value) return fail(KH_ERR_INVALID, "null value buffer");
    // always the caller's private slot: a plane / full-policy call is 26-49 KB per position of PCIe traffic, which
    // concurrent callers' own streams overlap better than one merged launch does (measured: 4 threads x 16 positions
    // 0.44 M/s on private slots, 0.17 M/s merged) — kh_submit_infer stays for callers that want the queue anyway
    return infer_host(e, input, nullptr, batch, policy, value, nullptr, nullptr);
}

int kh_submit_infer(kh_engine* e, const float* input, int batch, float* policy, float* value, int64_t* ticket)
{
    return co_submit(e, 1, nullptr, input, batch, nullptr, nullptr, nullptr, value, policy, ticket);
}

int kh_submit_encode_infer_legal(kh_engine* e, const kh_board* boards, int batch, const int32_t* action_offsets,
                                 const int32_t* actions, float* priors, float* value, int64_t* ticket)
{
    return co_submit(e, 0, boards, nullptr, batch, action_offsets, actions, priors, value, nullptr, ticket);
}

int kh_wait(kh_engine* e, int64_t ticket) { return co_wait(e, ticket); }

int kh_try_wait(kh_engine* e, int64_t ticket, int* done)
{
    if (!done) return fail(KH_ERR_INVALID, "null done");
    *done = 0;
    if (!e || !e->co) return fail(KH_ERR_INVALID, "no such ticket");
    Coalescer* c = e->co;
    const int tid = (int)(ticket & 0xffffffff);
    if (tid < 0 || tid >= KH_MAX_OUTSTANDING) return fail(KH_ERR_INVALID, "no such ticket");
    CoTicket& t = c->tickets[tid];
    int st = t.state.load(std::memory_order_acquire);
    if (st == 0 || t.serial != (uint32_t)(ticket >> 32)) return fail(KH_ERR_INVALID, "ticket already waited for (or never issued)");
    if (st == 4) { (void)co_claim_fetch(c, &t); st = t.state.load(std::memory_order_acquire); }
    if (st != 2) return KH_OK;                  // queued, on the device, or the dispatcher is copying its rows right now
    *done = 1;
    return co_wait(e, ticket);                  // settled: returns at once with the ticket's status
}

int kh_set_coalesce(kh_engine* e, int target_batch, int max_wait_us)
{
    if (!e || target_batch < 0 || target_batch > CO_ROWS || max_wait_us < 0 || max_wait_us > 1000000)
        return fail(KH_ERR_INVALID, "target_batch in [0, %d], max_wait_us in [0, 1000000]", CO_ROWS);
    e->co_target = target_batch;
    e->co_wait_us = max_wait_us;
    if (e->co) { e->co->submits.fetch_add(1); e->co->cv_lane.notify_all(); }
    return KH_OK;
}

int kh_set_coalesce_callers(kh_engine* e, int callers)
{
    if (!e || callers < 0 || callers > KH_MAX_OUTSTANDING) return fail(KH_ERR_INVALID, "callers in [0, %d]", KH_MAX_OUTSTANDING);
    if (callers > 0 && getenv("KAMI_CO_CALLERS") && atoi(getenv("KAMI_CO_CALLERS")) > 0) callers = std::min(callers, atoi(getenv("KAMI_CO_CALLERS")));   // A/B knob: seal at fewer submissions
    e->co_callers = callers;
    if (e->co) { e->co->submits.fetch_add(1); e->co->cv_lane.notify_all(); }
    return KH_OK;
}

int kh_coalesce_stats(kh_engine* e, int64_t* launches, int64_t* rows)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    int64_t l = 0, r = 0;
    if (e->co) { std::lock_guard<SpinLock> lk(e->co->mu); l = e->co->launches; r = e->co->rows_launched; }
    if (launches) *launches = l;
    if (rows) *rows = r;
    return KH_OK;
}

int kh_infer_full(kh_engine* e, const float* input, int batch, float* policy, float* value_full,
                  float* logits)
{
    return infer_host(e, input, nullptr, batch, policy, nullptr, value_full, logits);
}

int kh_encode_infer(kh_engine* e, const kh_board* boards, int batch, float* policy, float* value)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    if (e->cfg.features != KH_NFEATURES)
        return fail(KH_ERR_INVALID, "kh_encode_infer needs features == %d (Env::observe planes)", KH_NFEATURES);
    if (!value || !boards) return fail(KH_ERR_INVALID, "null buffer");
    return infer_host(e, nullptr, boards, batch, policy, value, nullptr, nullptr);
}

int kh_infer_legal(kh_engine* e, const float* input, int batch, const int32_t* action_offsets,
                   const int32_t* actions, float* priors, float* value)
{
    if (!value || !input) return fail(KH_ERR_INVALID, "null buffer");
    LegalIO l{ action_offsets, actions, priors };
    return infer_host(e, input, nullptr, batch, nullptr, value, nullptr, nullptr, &l);
}

int kh_encode_infer_legal(kh_engine* e, const kh_board* boards, int batch, const int32_t* action_offsets,
                          const int32_t* actions, float* priors, float* value)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    if (e->cfg.features != KH_NFEATURES)
        return fail(KH_ERR_INVALID, "kh_encode_infer_legal needs features == %d (Env::observe planes)", KH_NFEATURES);
    if (!value || !boards) return fail(KH_ERR_INVALID, "null buffer");
    LegalIO l{ action_offsets, actions, priors };
    if (batch >= 1 && batch <= CO_SMALL_LEGAL / 4 && action_offsets && actions && priors) {
        SmallCall sc(e);                 // counts this call as inside the engine until it returns, whichever path it takes
        if (sc.others) {
            int64_t t;
            const int rc = co_submit(e, 0, boards, nullptr, batch, action_offsets, actions, priors, value, nullptr, &t);
            return rc ? rc : co_wait(e, t);
        }
        return infer_host(e, nullptr, boards, batch, nullptr, value, nullptr, nullptr, &l);
    }
    return infer_host(e, nullptr, boards, batch, nullptr, value, nullptr, nullptr, &l);
}

int kh_encode(kh_engine* e, const kh_board* boards, int batch, float* planes)
{
    if (!e || !boards || !planes) return fail(KH_ERR_INVALID, "null argument");
    if (batch < 0) return fail(KH_ERR_INVALID, "negative batch");
    if (batch == 0) return KH_OK;
    int rc = set_device(e);
    if (rc) return rc;
    SlotLease lease(e);
    Slot& s = *lease.s;
    if (!s.stream) HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    const size_t B = batch;
    if (s.boards.ensure(B * sizeof(kh_board)) || s.planes.ensure(B * 64 * KH_NFEATURES * 4)) return KH_ERR_HIP;
    HIPCHK(hipMemcpyAsync(s.boards.p, boards, B * sizeof(kh_board), hipMemcpyHostToDevice, s.stream));
    kh::launch_encode_f32(s.boards.as<kh_board>(), batch, s.planes.as<float>(), s.stream);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(planes, s.planes.p, B * 64 * KH_NFEATURES * 4, hipMemcpyDeviceToHost, s.stream));
    HIPCHK(hipStreamSynchronize(s.stream));
    return KH_OK;
}

static int dev_slot(kh_engine* e, int batch, void* stream, Slot** out)
{
    if (!e->devslot) e->devslot.reset(new Slot());
    Slot& s = *e->devslot;
    int rc = slot_ensure(e, s, batch, false);
    if (rc) return rc;
    *out = &s;
    // per-layer paths keep activations in the slot's scratch: order this call behind the previous one when it
    // runs on a different stream (same stream: stream order already does it)
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : s.stream;
    if (s.scratch_pending && s.scratch_stream != st) HIPCHK(hipStreamWaitEvent(st, s.scratch_done, 0));
    return KH_OK;
}

// after a device-API forward: remember where the scratch was last used (only the per-layer paths use it; the
// whole-network kernel keeps its activations in LDS)
static int dev_slot_done(kh_engine* e, const Weights& W, Slot& s, hipStream_t st)
{
    const bool uses_scratch = e->cfg.dtype == KH_F32 || !W.tw_ok;
    if (!uses_scratch) return KH_OK;
    if (!s.scratch_done) HIPCHK(hipEventCreateWithFlags(&s.scratch_done, hipEventDisableTiming));
    HIPCHK(hipEventRecord(s.scratch_done, st));
    s.scratch_stream = st;
    s.scratch_pending = true;
    return KH_OK;
}

int kh_infer_device(kh_engine* e, const void* d_input, int batch, float* d_policy,
                    float* d_value_full, void* stream)
{
    if (!e || !d_input || !d_policy || !d_value_full) return fail(KH_ERR_INVALID, "null argument");
    if (batch < 1) return fail(KH_ERR_INVALID, "batch must be >= 1");
    auto W = current_weights(e);
    if (!W) return fail(KH_ERR_NO_WEIGHTS, "kh_infer_device before kh_load_weights");
    int rc = set_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(e->dmu);
    Slot* s;
    if ((rc = dev_slot(e, batch, stream, &s))) return rc;
    // run on the caller's stream when given; scratch is then ordered by that stream
    hipStream_t own = s->stream;
    if (stream) s->stream = static_cast<hipStream_t>(stream);
    rc = forward_dispatch(e, *W, *s, static_cast<const float*>(d_input), batch, d_policy, d_value_full, nullptr);
    if (!rc) rc = dev_slot_done(e, *W, *s, s->stream);
    s->stream = own;
    return rc;
}

int kh_encode_device(kh_engine* e, const kh_board* d_boards, int batch, float* d_planes, void* stream)
{
    if (!e || !d_boards || !d_planes) return fail(KH_ERR_INVALID, "null argument");
    if (batch < 1) return fail(KH_ERR_INVALID, "batch must be >= 1");
    int rc = set_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(e->dmu);
    if (!e->devslot) e->devslot.reset(new Slot());
    Slot& s = *e->devslot;
    if (!s.stream) HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    kh::launch_encode_f32(d_boards, batch, d_planes, stream ? static_cast<hipStream_t>(stream) : s.stream);
    HIPCHK(hipGetLastError());
    return KH_OK;
}

int kh_encode_infer_device(kh_engine* e, const kh_board* d_boards, int batch, float* d_policy,
                           float* d_value_full, void* stream)
{
    if (!e || !d_boards || !d_policy || !d_value_full) return fail(KH_ERR_INVALID, "null argument");
    if (batch < 1) return fail(KH_ERR_INVALID, "batch must be >= 1");
    if (e->cfg.features != KH_NFEATURES)
        return fail(KH_ERR_INVALID, "kh_encode_infer_device needs features == %d (Env::observe planes)", KH_NFEATURES);
    auto W = current_weights(e);
    if (!W) return fail(KH_ERR_NO_WEIGHTS, "kh_encode_infer_device before kh_load_weights");
    int rc = set_device(e);
    if (rc) return rc;
    std::lock_guard<std::mutex> lk(e->dmu);
    Slot* s;
    if ((rc = dev_slot(e, batch, stream, &s))) return rc;
    hipStream_t own = s->stream;
    if (stream) s->stream = static_cast<hipStream_t>(stream);
    if (fused_ingest(e, *W)) {
        rc = forward_tower(e, *W, *s, nullptr, batch, d_policy, d_value_full, nullptr, d_boards);
    } else {
        if (s->planes.ensure((size_t)batch * 64 * KH_NFEATURES * 4)) { s->stream = own; return KH_ERR_HIP; }
        kh::launch_encode_f32(d_boards, batch, s->planes.as<float>(), s->stream);
        rc = forward_dispatch(e, *W, *s, s->planes.as<float>(), batch, d_policy, d_value_full, nullptr);
        if (!rc) rc = dev_slot_done(e, *W, *s, s->stream);
    }
    s->stream = own;
    return rc;
}

static int time_loop(kh_engine* e, int iters, float* ms, int (*body)(void*), void* ctx)
{
    if (iters < 1 || !ms) return fail(KH_ERR_INVALID, "bad timing arguments");
    int rc = set_device(e);
    if (rc) return rc;
    if (!e->devslot) e->devslot.reset(new Slot());
    Slot& s = *e->devslot;
    if (!s.stream) HIPCHK(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    hipEvent_t t0, t1;
    HIPCHK(hipEventCreate(&t0));
    HIPCHK(hipEventCreate(&t1));
    if ((rc = body(ctx))) return rc;                       // warm-up (sizes scratch)
    HIPCHK(hipStreamSynchronize(s.stream));
    HIPCHK(hipEventRecord(t0, s.stream));
    for (int i = 0; i < iters; ++i)
        if ((rc = body(ctx))) return rc;
    HIPCHK(hipEventRecord(t1, s.stream));
    HIPCHK(hipEventSynchronize(t1));
    float total = 0.f;
    HIPCHK(hipEventElapsedTime(&total, t0, t1));
    *ms = total / iters;
    (void)hipEventDestroy(t0);
    (void)hipEventDestroy(t1);
    return KH_OK;
}

struct InferCtx { kh_engine* e; const void* in; int B; float *p, *v; };
struct EncCtx { kh_engine* e; const kh_board* b; int B; float* planes; };

int kh_time_infer_device(kh_engine* e, const void* d_input, int batch, float* d_policy,
                         float* d_value_full, int iters, float* ms_per_launch)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    InferCtx c{ e, d_input, batch, d_policy, d_value_full };
    return time_loop(e, iters, ms_per_launch,
                     [](void* p) { auto* c = static_cast<InferCtx*>(p); return kh_infer_device(c->e, c->in, c->B, c->p, c->v, nullptr); }, &c);
}

int kh_time_encode_device(kh_engine* e, const kh_board* d_boards, int batch, float* d_planes,
                          int iters, float* ms_per_launch)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    EncCtx c{ e, d_boards, batch, d_planes };
    return time_loop(e, iters, ms_per_launch,
                     [](void* p) { auto* c = static_cast<EncCtx*>(p); return kh_encode_device(c->e, c->b, c->B, c->planes, nullptr); }, &c);
}

int kh_pin_buffer(kh_engine* e, void* ptr, size_t bytes)
{
    if (!e || !ptr || !bytes) return fail(KH_ERR_INVALID, "null argument");
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    std::lock_guard<std::mutex> lk(e->pin_mu);
    e->pinned.emplace_back(static_cast<const char*>(ptr), bytes);
    return KH_OK;
}

int kh_unpin_buffer(kh_engine* e, void* ptr)
{
    if (!e || !ptr) return fail(KH_ERR_INVALID, "null argument");
    {
        std::lock_guard<std::mutex> lk(e->pin_mu);
        auto it = std::find_if(e->pinned.begin(), e->pinned.end(), [&](const std::pair<const char*, size_t>& r) { return r.first == ptr; });
        if (it == e->pinned.end()) return fail(KH_ERR_INVALID, "buffer was not registered with kh_pin_buffer");
        e->pinned.erase(it);
    }
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipHostUnregister(ptr));
    return KH_OK;
}

int kh_dev_alloc(kh_engine* e, size_t bytes, void** d_ptr)
{
    if (!e || !d_ptr) return fail(KH_ERR_INVALID, "null argument");
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipMalloc(d_ptr, bytes));
    return KH_OK;
}
int kh_dev_free(kh_engine* e, void* d_ptr)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipFree(d_ptr));
    return KH_OK;
}
int kh_memcpy_h2d(kh_engine* e, void* d_dst, const void* h_src, size_t bytes)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipMemcpy(d_dst, h_src, bytes, hipMemcpyHostToDevice));
    return KH_OK;
}
int kh_memcpy_d2h(kh_engine* e, void* h_dst, const void* d_src, size_t bytes)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
    return KH_OK;
}
int kh_sync(kh_engine* e)
{
    if (!e) return fail(KH_ERR_INVALID, "null engine");
    int rc = set_device(e);
    if (rc) return rc;
    HIPCHK(hipDeviceSynchronize());
    return KH_OK;
}

}  // extern "C"
