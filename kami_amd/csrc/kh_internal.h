// kh_internal.h — declarations shared by the translation units of libkamihip.so.
// Not part of the public boundary (that is include/kami_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/kami_hip.h"

namespace kh {

// ---- encode.hip -------------------------------------------------------------------------
// Env::observe (kami/env.h:202-262) for n records -> fp32 planes [n][8][8][30].
void launch_encode_f32(const kh_board* d_boards, int n, float* d_planes, hipStream_t s);

// priors[k] = policy[i][actions[k]] / sum over position i's actions (mcts.h:273-276); one wave per position
void launch_gather_legal(const float* policy, const int32_t* offsets, const int32_t* actions,
                         float* priors, int B, hipStream_t s, const float* vfull = nullptr, int vstride = 0,
                         float* values = nullptr, const int* flags_in = nullptr, int* flags_out = nullptr);

// stores `serial` to a word of page-locked host memory once the stream's earlier work has finished
void launch_signal(unsigned* host_word, unsigned serial, hipStream_t s);

// ---- forward_simple.hip -----------------------------------------------------------------
// Plain fp32 VALU kernels, one launch per layer.  Exact-order fp32 (same tap-major,
// channel-inner accumulation order as the CPU oracle); the correctness anchor on device.
struct SimpleLayer {            // device pointers
    const float* wt;            // [taps][Ci][Co]
    const float* scale;         // [Co]  gamma / sqrt(var + eps)            (1 for plain conv)
    const float* shift;         // [Co]  (bias - mean) * scale + beta       (bias for plain conv)
    int Ci, Co, taps;           // taps = 9 (3x3 pad 1) or 1
    int relu;
};
// out = (skip ? skip : 0) + act(conv(in) * scale + shift); activations [B][64][C] fp32
void launch_simple_conv(const SimpleLayer& L, const float* in, const float* skip, float* out,
                        int B, hipStream_t s);
// logits [B][4672] -> policy = exp(log_softmax(logits)) (nn.cpp:80); sets flags[0] on NaN
void launch_softmax4672(const float* logits, float* policy, int B, int* flags, hipStream_t s);
// v64 [B][64] -> value_full [B][256] = tanh(v64 W^T + b) (nn.cpp:86-88); sets flags[1] on NaN
void launch_value_fc(const float* v64, const float* fcw, const float* fcb, float* value_full,
                     int B, int* flags, hipStream_t s);

// ---- tower_mfma.hip ---------------------------------------------------------------------
// Whole-network persistent MFMA kernel (bf16 / f16 operands, fp32 accumulate).
constexpr int TW_NB = 2;     // boards per workgroup pass
constexpr int TW_CP = 64;    // channel count the kernel is specialised for (smaller nets are zero-padded)

struct TowerArgs {
    const float* in;         // [B][8][8][F] fp32 planes (ignored when boards != nullptr)
    const kh_board* boards;  // nullable: compact ingest, Env::observe fused into the kernel (F must be 30)
    int B, F, R;
    const char* wstream;     // packed weight fragments, nchunks x 8 KB (see pack_tower in kh_api.hip)
    int nchunks;
    const float* params;     // folded shifts / value-conv weights, npar floats
    int npar;
    const float* fcw4;       // valuefc.weight re-laid [k/4][j][4]
    const float* fcb;        // [256]
    float* policy;           // [B][4672]
    float* vfull;            // [B][256]
    float* logits;           // nullable [B][4672]
    int* flags;              // [0] policy NaN, [1] value NaN
    // Legal-move mode (nullable as a whole: lg_offsets == nullptr): instead of the 4 672-entry policy row the kernel writes
    // the priors of each board's legal actions, renormalised over them (MCTS::expand, mcts.h:273-276,296 — what
    // gather_legal_kernel computes from the row, same arithmetic in the same order), one value per board (column 0 of the
    // value tensor) and the NaN flags, all of it straight into the caller's (page-locked) blocks: one launch per evaluation.
    const int32_t* lg_offsets = nullptr;   // [B + 1]
    const int32_t* lg_actions = nullptr;
    float* lg_priors = nullptr;
    float* lg_values = nullptr;            // [B]
    int* lg_flags = nullptr;               // [2], zeroed by the host before the launch; a flagging lane stores 1
};

// floats of LDS parameter area: shifts of the 1+2R 3x3 layers, policy shifts/bias, value conv, scratch
__host__ __device__ constexpr int tower_par_copy_floats(int R) { return (1 + 2 * R) * TW_CP + 128 + 128 + TW_CP + 4; }
__host__ __device__ constexpr int tower_par_floats(int R) { return tower_par_copy_floats(R) + TW_NB * 64 + 16; }
int tower_lds_bytes(int FP, int R);
hipError_t launch_tower(int dtype, int FP, const TowerArgs& a, int num_cus, hipStream_t s);
// tower8_mfma.hip: the same forward with specialised waves (4 compute + 4 helper waves per workgroup); launch_tower
// dispatches to it (KAMI_TOWER_V=4|8 overrides the build's default for A/B runs)
hipError_t launch_tower8(int dtype, int FP, const TowerArgs& a, int num_cus, hipStream_t s);
// 4 or 8: which of the two launch_tower runs.  They walk a 33..128-plane stem in different orders (two passes of 64
// planes / four of 32), so TowerArgs::wstream must be the stream packed for the running one.
int tower_variant();

// ---- layers_mfma.hip ----------------------------------------------------------------------
// bf16 / f16 path for wide nets (65..256 filters): one MFMA kernel launch per layer.
struct LayersArgs {
    const float* in;               // [B][8][8][F] fp32 planes
    int B, F, FP, CP, R;           // FP = F rounded up to 16, CP = filters rounded up to 64
    unsigned short* act_in;        // T [B][64][FP]
    unsigned short* act[3];        // T [B][64][CP]
    unsigned short* pmid;          // T [B][64][128]
    float* logits;                 // [B][4672]
    float* v64;                    // [B][64]
    const unsigned short* w;       // packed fragments of all layers (device)
    const size_t* w_off;           // HOST array: element offset of each layer in w
    const unsigned short* w4;      // 3x3 layers packed for conv4_mfma_kernel (device; nullptr when no layer is eligible)
    const size_t* w4_off;          // HOST array: element offset of each layer in w4, (size_t)-1 = not eligible
    const float* shift;            // folded shifts of all layers (device)
    const size_t* shift_off;       // HOST array: float offset of each layer in shift
    const float* vw;               // [CP] valueconv weight * bn scale (device)
    float vshift;
    const unsigned short* w2b;     // stem + tower packed for tower256_kernel (device; nullptr: not eligible)
    const unsigned short* wh;      // policy head packed for policy_head4_kernel (device; nullptr: not eligible)
    float* policy;                 // [B][4672]: launch_layers runs the softmax itself (nn.cpp:80)
    int* flags;                    // NaN flags ([0] policy, [1] value)
    bool want_logits;              // the caller asked for the pre-softmax logits in `logits`
    const float* fcw;              // valuefc.weight [256][64], .bias [256] (device); launch_layers runs the value FC too
    const float* fcb;
    const float* fc4 = nullptr;    // valuefc.weight as [k / 4][256][4] (nullable: the kernels then read `fcw`)
    float* vfull;                  // [B][256]
    // tower2s_kernel's exchange area (nullable): [x_pairs][2] flags 64 bytes apart, then [2][x_pairs][2][128][128] T
    unsigned* xflag = nullptr;
    unsigned short* xbuf = nullptr;
    int x_pairs = 0;
    int num_cus = 256;
};
size_t layers_lds_bytes(int Ci);
// bytes of the exchange area for up to `pairs` board pairs: flags first (the block the launcher zeroes), then the images
constexpr size_t layers_xflag_bytes(int pairs) { return (size_t)pairs * 2 * 64; }
constexpr size_t layers_xchg_bytes(int pairs) { return layers_xflag_bytes(pairs) + (size_t)2 * pairs * 2 * 128 * 128 * 2; }
// exact-fp32 MFMA convolution (conv_f32_kernel) for the trainer: out [B][64][Co] = conv(in [B][64][Ci]) + bias (nullable),
// or out += conv(in) when `accumulate`.  packed_w: fragments in conv_f32_kernel's order, Co rounded up to 64,
// [Co/64][Ci slices of <= 128][taps][slice/8][2][64 lanes][4] (train.hip packs them on the device).  Ci % 8 == 0.
hipError_t conv_f32_raw_prepare();
hipError_t launch_conv_f32_raw(const float* in, const float* packed_w, const float* bias, float* out, int B, int Ci, int Co, int taps,
                               bool accumulate, hipStream_t s);
hipError_t launch_layers(int dtype, const LayersArgs& L, hipStream_t s);

// ---- train.hip ------------------------------------------------------------------------------
// NN::train (nn.cpp:224-377) as fp32 device kernels: one SGD step per call on a device parameter blob.
struct TrainNet;
struct StepBuffers { float* params; float* grads; float* work; };
TrainNet* train_layout_new(int F, int C, int R);
void train_layout_free(TrainNet* n);
size_t train_workspace_floats(int F, int C, int R, int B);
hipError_t train_step(const TrainNet& n, const StepBuffers& sb, const float* x_in, const float* obsp, const float* obsv,
                      int B, float lr, float* loss_rows /* [2 B] floats + 2 ints (NaN flags of the forward's outputs) */, hipStream_t s);

}  // namespace kh
