/*
 * kami_hip.h — C ABI of libkamihip.so, the MI355X (gfx950) leaf-evaluation engine.
 *
 * This is the drop-in boundary for ONE path of codeandkey/kami: board->plane
 * encoding (kami/env.h:202-262, Env::observe) and the batched network forward
 * behind kami::NN::infer (kami/nn/nn.cpp:155-187 -> NNModule::forward nn.cpp:59-91
 * -> NNResidual::forward nn.cpp:26-34).  Everything is extern "C", plain pointers
 * and sizes; no torch / C++ types cross this line.  The C++ class kami::NN in
 * kami_amd/host/nn.h and the ctypes mirror in kami_amd/nn.py sit on top of it.
 *
 * Each entry point names the reference interface it replaces (file:line under
 * the reference tree).
 */
#ifndef KAMI_HIP_H
#define KAMI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Constants of the reference model (kami/env.h:19-23, kami/nn/nn.cpp:47-52). */
#define KH_WIDTH          8
#define KH_HEIGHT         8
#define KH_NFEATURES      30      /* env.h:19  NFEATURES = 8+6+4+12            */
#define KH_POLICY_PLANES  73      /* nn.cpp:51 policyconv2 out channels        */
#define KH_PSIZE          4672    /* env.h:20  PSIZE = 73*64                   */
#define KH_POLICY_MID     128     /* nn.cpp:47,50 policyconv out channels      */
#define KH_VALUE_WIDTH    256     /* nn.cpp:52 valuefc = Linear(w*h, 256)      */

/* Status codes.  kami::NN (host/nn.h) maps 4/5 to the reference's exception
 * strings "inference policy output contains NaN" / "inference value output
 * contains NaN" (nn.cpp:176-180). */
enum {
    KH_OK = 0,
    KH_ERR_INVALID = 1,      /* bad argument / unsupported configuration      */
    KH_ERR_HIP = 2,          /* HIP runtime error (message in kh_last_error)  */
    KH_ERR_NO_WEIGHTS = 3,   /* infer before kh_load_weights                  */
    KH_ERR_NAN_POLICY = 4,
    KH_ERR_NAN_VALUE = 5,
    KH_ERR_NO_DEVICE = 6     /* no gfx950 device visible: the engine has NO CPU fallback */
};

/* Arithmetic type of the conv tower (accumulation is always fp32). */
enum { KH_F32 = 0, KH_BF16 = 1, KH_F16 = 2 };

/* What kh_infer writes to value[0..batch):
 *  REFERENCE_FLAT: the first `batch` floats of the flattened [batch,256] value
 *      tensor — bit-for-bit what nn.cpp:186 memcpy's.
 *  PER_SAMPLE0:    value[i] = vh[i][0]  (opt-in, not reference behaviour). */
enum { KH_VALUE_REFERENCE_FLAT = 0, KH_VALUE_PER_SAMPLE0 = 1 };

/* Construction parameters: NN::NN(width,height,features,psize) nn.h:50 plus the
 * two option keys the module reads, "filters"/"residuals" (nn.cpp:42-43). */
typedef struct kh_config {
    int32_t width, height;   /* must be 8, 8                                   */
    int32_t features;        /* input planes F (30 = reference encoder; 119 = BASELINE metric) */
    int32_t psize;           /* must be 4672                                   */
    int32_t filters;         /* C                                              */
    int32_t residuals;       /* R                                              */
    int32_t dtype;           /* KH_F32 | KH_BF16 | KH_F16                      */
    int32_t value_mode;      /* KH_VALUE_*                                     */
    int32_t device;          /* HIP device ordinal                             */
    int32_t reserved[7];     /* zero                                           */
} kh_config;

/* Compact board record: exactly the state Env::observe reads (env.h:202-262):
 * piece placement (board.h:14 piece_occ/color_occ), side to move (position.h:40),
 * top-of-stack castle_rights and halfmove_clock (position.h:27,30) and
 * history.size() (env.h:211).  80 bytes, 16-byte aligned loads on device. */
typedef struct kh_board {
    uint64_t piece_occ[6];   /* P,N,B,R,Q,K (types.h:44-49), bit = square rank*8+file */
    uint64_t color_occ[2];   /* white, black                                   */
    int32_t  ply;            /* history.size()                                 */
    int32_t  halfmove_clock;
    uint8_t  ctm;            /* 0 white, 1 black                               */
    uint8_t  castle_rights;  /* WK=1 WQ=2 BK=4 BQ=8 (position.h:13-16)         */
    uint8_t  pad[6];
} kh_board;

typedef struct kh_engine kh_engine;

/* Number of fp32 values in a weight blob for (features, filters, residuals).
 * Blob order (all fp32, row-major, libtorch shapes; names per nn.cpp:20-23,45-56):
 *   conv1.weight[C,F,3,3] conv1.bias[C] batchnorm1.{weight,bias,running_mean,running_var}[C]
 *   for i in 0..R-1: residual{i}.conv1.weight[C,C,3,3] .conv1.bias[C]
 *                    residual{i}.batchnorm1.{weight,bias,running_mean,running_var}[C]
 *                    residual{i}.conv2.weight[C,C,3,3] .conv2.bias[C]
 *                    residual{i}.batchnorm2.{weight,bias,running_mean,running_var}[C]
 *   policyconv.weight[128,C,1,1] policyconv.bias[128] pbatchnorm.{w,b,rm,rv}[128]
 *   policyconv2.weight[73,128,1,1] policyconv2.bias[73]
 *   valueconv.weight[1,C,1,1] valueconv.bias[1] vbatchnorm.{w,b,rm,rv}[1]
 *   valuefc.weight[256,64] valuefc.bias[256]                                       */
size_t kh_weight_count(int features, int filters, int residuals);

/* Replaces NN::NN(...) nn.cpp:107-128.  Fails with KH_ERR_NO_DEVICE when no
 * gfx950 GPU is visible (there is deliberately no CPU path). */
int  kh_create(const kh_config* cfg, kh_engine** out);
void kh_destroy(kh_engine* e);

/* Replaces NN::read's effect on the model (nn.cpp:204-222): install a full
 * parameter set + generation.  In-flight kh_infer calls finish on the old set. */
int  kh_load_weights(kh_engine* e, const float* blob, size_t nfloats, int generation);

/* NN::train nn.cpp:224-377: `epochs` passes of plain SGD (batch `batch`, learning rate `lr`) over
 * `trajectories` samples, in the reference's order (one default_random_engine shuffled per epoch;
 * the last, short batch of an epoch is padded with the previous batch's rows like the reference's
 * stack buffers), network in training mode (BatchNorm on batch statistics, running statistics
 * updated), loss = -sum(obs_p * log(p + 0.001)) + mean((v - obs_v)^2) (nn.cpp:93-105).  The trained
 * parameters become the engine's weights with generation + 1.  fp32 arithmetic whatever cfg.dtype.
 *   inputs [n][8][8][F], obs_p [n][4672], obs_v [n]; first_loss / last_loss: average loss of the first
 * and last epoch (the reference prints them), nullable. */
typedef struct kh_train_config {
    float   lr;             /* "training_mlr" / 1000  (nn.cpp:236)  */
    int32_t epochs;         /* "training_epochs"       (nn.cpp:237)  */
    int32_t batch;          /* "training_batchsize"    (nn.cpp:238)  */
    int32_t detect_anomaly; /* NN::train's last argument (nn.cpp:224,231-232,329-344): nonzero = every batch is checked — its
                               input for NaN before the step, the forward's value and policy outputs behind it — and kh_train
                               fails with the reference's messages "training input ind <i> contains NaN" (KH_ERR_INVALID),
                               "forward value output contains NaN" (KH_ERR_NAN_VALUE), "forward policy output contains NaN"
                               (KH_ERR_NAN_POLICY); the parameters stay as they were before the call */
    int32_t reserved[4];
} kh_train_config;
int  kh_train(kh_engine* e, const float* inputs, const float* obs_p, const float* obs_v, int trajectories,
              const kh_train_config* cfg, float* first_loss, float* last_loss);
/* The order kh_train visits the samples in: order[epoch * trajectories + k] = index of the k-th sample of that epoch (one
 * std::default_random_engine{} for the call, one std::shuffle per epoch: nn.cpp:245-262); consecutive runs of `batch` of
 * them are the batches, a short last batch keeps the previous batch's rows behind its own (the reference's staging buffers
 * persist).  No engine needed: what a restatement of a multi-batch run has to follow. */
int  kh_train_order(int trajectories, int epochs, int32_t* order);

/* NN::read nn.cpp:204-222: parse a checkpoint file without an engine.  Two containers are understood:
 * the reference's own — the libtorch archive NN::write leaves (nn.cpp:189-202: module.save + the
 * "generation" IValue), read here with a zip walk and a pickle stack machine (csrc/torch_archive.h; no
 * libtorch, nothing from the file is executed) — and the engine's "KAMW" container (32-byte header + blob).
 * Fills the network shape, generation and *nfloats; when `blob` is non-NULL (capacity `cap` floats) also the
 * parameters in blob order (BatchNorm num_batches_tracked counters are not part of the blob).  No GPU needed. */
int  kh_checkpoint_read(const char* path, int* features, int* filters, int* residuals, int* generation,
                        float* blob, size_t cap, size_t* nfloats);
/* NN::read as a whole: kh_checkpoint_read + shape check against cfg + kh_load_weights. */
int  kh_load_checkpoint(kh_engine* e, const char* path);

/* The engine's current fp32 parameter set in blob order (what NN::write would serialise, nn.cpp:189-202). */
int  kh_get_weights(kh_engine* e, float* blob, size_t nfloats);

/* NN::get_generation nn.h:53-59. */
int  kh_generation(kh_engine* e);

/* NN(NN* other) nn.cpp:130-153: a new engine with the same config and weights. */
int  kh_clone(kh_engine* src, kh_engine** out);

/* NN::infer nn.cpp:155-187.  Host buffers, caller-owned:
 *   input  [batch][8][8][F] fp32 channels-last (nn.cpp:157)
 *   policy [batch][4672]   softmax over all 4672 entries (nn.cpp:78-80)
 *   value  [batch]         per cfg.value_mode
 * Returns KH_ERR_NAN_POLICY / KH_ERR_NAN_VALUE where the reference throws. */
int  kh_infer(kh_engine* e, const float* input, int batch, float* policy, float* value);

/* Optional, for kh_infer's host buffers: register a LONG-LIVED caller buffer (selfplay.cpp's `batch` / `inf_policy`
 * arrays live as long as their inference thread) with the device, so that kh_infer moves it by plain DMA instead of having
 * the runtime pin and unpin the caller's pages at every call, and overlaps the upload of one part of the batch with the
 * kernel and the policy download of the previous part.  Applies when BOTH `input` and `policy` of a call lie inside
 * registered buffers (bf16 / f16 engines of <= 64 filters); results are the same bits either way.  The caller must
 * kh_unpin_buffer before it frees the memory: a registration that outlives its allocation would let the device read
 * whatever is mapped there next — which is why this is opt-in and not something kh_infer does behind the caller's back. */
int  kh_pin_buffer(kh_engine* e, void* ptr, size_t bytes);
int  kh_unpin_buffer(kh_engine* e, void* ptr);

/* Diagnostic superset of kh_infer used by the parity tests: additionally returns
 * the whole value tensor [batch][256] (nn.cpp:86-88) and, if non-NULL, the
 * pre-softmax policy logits [batch][4672] (nn.cpp:75-79). */
int  kh_infer_full(kh_engine* e, const float* input, int batch,
                   float* policy, float* value_full, float* logits);

/* Env::observe env.h:202-262 for a batch of compact records -> fp32 planes
 * [batch][8][8][30], bit-exact.  Host buffers. */
int  kh_encode(kh_engine* e, const kh_board* boards, int batch, float* planes);

/* Compact ingest: Env::observe + NN::infer fused on device (selfplay.cpp:133,196
 * collapsed into one call).  Requires cfg.features == 30. */
int  kh_encode_infer(kh_engine* e, const kh_board* boards, int batch,
                     float* policy, float* value);

/* Legal-move policy gather: what MCTS::expand does with the network output (mcts.h:273-276,296)
 * moved to the device, so that only the priors of each position's legal actions cross PCIe
 * instead of 4672 floats.  Position i owns actions[action_offsets[i] .. action_offsets[i+1]) (action
 * codes as produced by Env::actions, env.h:398-423);
 *   priors[k] = policy_i[actions[k]] / sum_j policy_i[actions[j]]      (all zeros if the sum is 0)
 * `value` as in kh_infer.  The noise mixing of mcts.h:279-296 stays with the caller.
 * kh_infer_legal takes fp32 planes, kh_encode_infer_legal takes compact board records. */
int  kh_infer_legal(kh_engine* e, const float* input, int batch,
                    const int32_t* action_offsets, const int32_t* actions,
                    float* priors, float* value);
int  kh_encode_infer_legal(kh_engine* e, const kh_board* boards, int batch,
                           const int32_t* action_offsets, const int32_t* actions,
                           float* priors, float* value);

/* ---- submit / wait: the engine's coalescing queue (SURVEY 8b threading row) ----------------------------------
 * The reference's callers (selfplay.cpp:196 on `inference_threads` threads, evaluate.cpp:138,147) each bring a small
 * batch.  kh_submit_* queues one such batch and returns a ticket at once; kh_wait blocks until its results are in the
 * caller's buffers and returns the status kh_infer / kh_encode_infer_legal would have (the NaN codes included, attributed
 * to the submission whose rows hold the NaN).  Submissions of different callers that are queued while a launch is in
 * flight are evaluated as ONE launch (up to 1024 positions).  All buffers — inputs too — stay the caller's and must stay
 * valid and untouched until kh_wait returns.  At most KH_MAX_OUTSTANDING tickets per engine may be un-waited: one more
 * kh_submit_* returns KH_ERR_INVALID (it never blocks on the caller's own outstanding work).  A submission holds at most
 * 512 positions (kh_submit_encode_infer_legal) / 128 positions (kh_submit_infer); larger batches gain nothing from
 * merging and take the synchronous calls.  The synchronous kh_encode_infer_legal uses this queue itself whenever
 * several small calls are inside the engine at the same moment (its payload is ~200 B per position: the launch is
 * what costs), and the private-slot path otherwise; kh_infer always keeps its private slot (26-49 KB per position
 * over PCIe: concurrent callers' own streams overlap that better than a merged launch). */
#define KH_MAX_OUTSTANDING 64
int  kh_submit_infer(kh_engine* e, const float* input, int batch, float* policy, float* value, int64_t* ticket);
int  kh_submit_encode_infer_legal(kh_engine* e, const kh_board* boards, int batch,
                                  const int32_t* action_offsets, const int32_t* actions,
                                  float* priors, float* value, int64_t* ticket);
int  kh_wait(kh_engine* e, int64_t ticket);
/* kh_wait without the wait: *done = 1 and the ticket is consumed exactly as by kh_wait (return value = its status) when
 * the results are in the caller's buffers; *done = 0 and KH_OK when the launch is still on its way (the ticket stays
 * valid).  For a caller with several tickets in flight that wants whichever finishes first (the self-play pool's
 * workers: launches do not complete in submission order once several are on the device). */
int  kh_try_wait(kh_engine* e, int64_t ticket, int* done);
/* Launch policy of the queue.  target_batch 0 (default): whatever has accumulated goes as soon as one of the queue's
 * four launch streams is free.  target_batch > 0: a batch waits until it holds that many positions, but at most max_wait_us after its first
 * submission and no longer than max_wait_us / 8 after its latest one (the burst has ended) — for callers that know how
 * many positions they keep in flight (the self-play pool). */
int  kh_set_coalesce(kh_engine* e, int target_batch, int max_wait_us);
/* With a target set: a batch also goes as soon as it holds `callers` submissions (0, the default: rule off) — for a pool
 * whose every worker submits once per round: `callers` = the workers is a whole round whatever its size (terminal leaves
 * need no evaluation, so rounds rarely reach the target exactly); half of them (what ks_pool_run sets) lets the early
 * half go without waiting for the slowest worker. */
int  kh_set_coalesce_callers(kh_engine* e, int callers);
/* launches made by the queue so far and the positions they held (mean coalesced batch = rows / launches) */
int  kh_coalesce_stats(kh_engine* e, int64_t* launches, int64_t* rows);

/* Device-resident variants (pointers are HIP device pointers).  Asynchronous: no host sync, no NaN check.
 * d_value_full is [batch][256].
 * `stream` is a hipStream_t; NULL means the ENGINE'S OWN non-blocking stream — not the legacy default stream
 * (handle 0, which is what e.g. torch.cuda.current_stream() reports by default): with NULL the kernels are
 * unordered against work the caller queued elsewhere, so the inputs must already be complete and the caller
 * synchronises (kh_sync) before reading results.  Calls on different streams may overlap only as far as the
 * engine's scratch allows: networks that run layer by layer (fp32, > 64 filters) keep activations in one
 * per-engine scratch, so each call is ordered behind the previous one by an event; the whole-network kernel
 * (bf16 / f16, <= 64 filters) has no such scratch. */
int  kh_infer_device(kh_engine* e, const void* d_input, int batch,
                     float* d_policy, float* d_value_full, void* stream);
int  kh_encode_device(kh_engine* e, const kh_board* d_boards, int batch,
                      float* d_planes, void* stream);
/* Compact ingest, device-resident: records in, policy / value tensor out.  With a bf16 / f16 engine
 * of <= 64 filters the encoder runs INSIDE the forward kernel (no planes in HBM); otherwise the
 * encode kernel writes planes to engine scratch first.  Requires cfg.features == 30. */
int  kh_encode_infer_device(kh_engine* e, const kh_board* d_boards, int batch,
                            float* d_policy, float* d_value_full, void* stream);

/* Timing helper for bench/roofline: runs `iters` back-to-back kh_infer_device
 * launches on the engine stream bracketed by HIP events recorded on THAT stream
 * and returns the average milliseconds per launch in *ms_per_launch. */
int  kh_time_infer_device(kh_engine* e, const void* d_input, int batch,
                          float* d_policy, float* d_value_full,
                          int iters, float* ms_per_launch);
int  kh_time_encode_device(kh_engine* e, const kh_board* d_boards, int batch,
                           float* d_planes, int iters, float* ms_per_launch);

/* Plain device-memory plumbing so hosts without torch can drive the device API. */
int  kh_dev_alloc(kh_engine* e, size_t bytes, void** d_ptr);
int  kh_dev_free(kh_engine* e, void* d_ptr);
int  kh_memcpy_h2d(kh_engine* e, void* d_dst, const void* h_src, size_t bytes);
int  kh_memcpy_d2h(kh_engine* e, void* h_dst, const void* d_src, size_t bytes);
int  kh_sync(kh_engine* e);

/* Number of visible HIP devices (0 when none / no driver). */
int  kh_device_count(void);

/* Thread-local message for the last non-OK status returned on this thread. */
const char* kh_last_error(void);

/* Library version string, e.g. "kamihip 0.1 gfx950". */
const char* kh_version(void);

#ifdef __cplusplus
}
#endif
#endif /* KAMI_HIP_H */
