/*
 * kami_oracle.h — CPU restatement of kami's leaf-evaluation path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker; the product (libkamihip.so) never links or calls it.
 *
 * Pinning: the reference's own tests hold no golden numbers for this path
 * (SURVEY §8c); the oracle is pinned against outputs of the UNMODIFIED reference
 * compiled in the build container (oracle/_ref/kami_ref, recipe oracle/Makefile),
 * committed as fixtures under tests/golden/ by oracle/gen_golden.py.
 */
#ifndef KAMI_ORACLE_H
#define KAMI_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Same 80-byte record as kh_board in include/kami_hip.h (kept separate on purpose:
 * the oracle shares no header with the product). */
typedef struct ko_board {
    uint64_t piece_occ[6];
    uint64_t color_occ[2];
    int32_t  ply;
    int32_t  halfmove_clock;
    uint8_t  ctm;
    uint8_t  castle_rights;
    uint8_t  pad[6];
} ko_board;

size_t ko_weight_count(int features, int filters, int residuals);

/* Env::observe, kami/env.h:202-262.  dst = float[64*30]. */
void ko_observe(const ko_board* b, float* dst);
void ko_observe_batch(const ko_board* b, int n, float* dst);

/* Parse the FEN printed by Env::print (env.h:425-430 -> position.c:131-165) plus
 * history.size() into a record.  Returns 0 on success. */
int ko_board_from_fen(const char* fen, int ply, ko_board* out);

/* NNModule::forward, kami/nn/nn.cpp:59-91 (+ NNResidual::forward nn.cpp:26-34), eval-mode
 * BatchNorm, fp32.  in [B][8][8][F]; policy [B][4672]; value_full [B][256];
 * logits (nullable) [B][4672] pre-softmax.  nthreads<=0 -> all cores. */
int ko_forward(const float* blob, int F, int C, int R, const float* in, int B,
               float* policy, float* value_full, float* logits, int nthreads);

/* NN::infer, kami/nn/nn.cpp:155-187: forward + NaN guards (returns 4 for policy NaN,
 * 5 for value NaN, like KH_ERR_NAN_*) + the Q10 copy-out of the first B floats of the
 * flattened [B,256] value tensor (nn.cpp:186). */
int ko_infer(const float* blob, int F, int C, int R, const float* in, int B,
             float* policy, float* value, int nthreads);

int ko_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
