/*
 * kami_search.h — C ABI of libkamisearch.so: the host side that FEEDS the leaf-evaluation path
 * (SURVEY §8f row 2).  It mirrors, without libtorch / neocortex / thc:
 *   kami::Env      kami/env.h:41-485      rules, action code, terminal test   (kami_amd/host/env.h)
 *   kami::MCTS     kami/mcts.h:66-349     PUCT search                         (kami_amd/host/mcts.h)
 *   Selfplay::inference_main  kami/selfplay.cpp:58-213   the batch loop       (ks_pool_*)
 * and hands positions to the engine of kami_hip.h as compact records + legal-action lists
 * (kh_encode_infer_legal), so 80 bytes go in and one prior per legal move comes back per leaf.
 * The ks_env_* / ks_mcts_* entry points exist so that the mirrors can be checked against the
 * reference's fixtures from any language; a C++ caller includes the headers under kami_amd/host/.
 */
#ifndef KAMI_SEARCH_H
#define KAMI_SEARCH_H

#include "kami_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- rules (kami/env.h) --------------------------------------------------------------------- */
/* leaf count of the legal-move tree of `fen` to `depth` (standard perft; known answers pin the rules) */
int ks_perft(const char* fen, int depth, uint64_t* nodes);
/* legal action codes (ascending) of the position `fen` describes, side-to-move point of view; returns the count */
int ks_fen_actions(const char* fen, int32_t* out, int cap);

typedef struct ks_env ks_env;
ks_env* ks_env_new(void);                                  /* Env::Env            env.h:50-54   */
void    ks_env_free(ks_env* e);
int     ks_env_ply(ks_env* e);                             /* Env::ply            env.h:58      */
/* legal action codes, ascending (the reference's order is its move-ordering heuristic's; same set) */
int     ks_env_actions(ks_env* e, int32_t* out, int cap);  /* Env::actions        env.h:397-423 */
int     ks_env_push(ks_env* e, int action);                /* Env::push           env.h:264-271; 1 = not a legal action */
int     ks_env_pop(ks_env* e);                             /* Env::pop            env.h:273-279 */
int     ks_env_terminal(ks_env* e, float* value);          /* Env::terminal       env.h:386-390 */
float   ks_env_turn(ks_env* e);                            /* Env::turn           env.h:392-395 */
int     ks_env_fen(ks_env* e, char* buf, int cap);         /* Env::print          env.h:425-430 */
void    ks_env_record(ks_env* e, kh_board* out);           /* what Env::observe reads, env.h:202-262 */

/* ---- search (kami/mcts.h) ------------------------------------------------------------------- */
/* kami::MCTS driven by the synthetic evaluator of the reference harness (`kami_ref mcts`, test infrastructure), noise off,
 * `leaves` positions in flight per step (1 = the reference's schedule).  Writes the harness's text
 * (move / root / child / pick lines).  picks: optional moves to play instead of pick(0) (ties in the
 * visit count resolve by child order, which differs from the reference's). */
int ks_mcts_synthetic(int nodes, int nmoves, int leaves, const int32_t* picks, int npicks, char* out, int cap);

/* ---- self-play pool (kami/selfplay.cpp:58-213) ---------------------------------------------- */
typedef struct ks_pool_config {
    int32_t games;              /* trees in play (the reference: inference_threads x selfplay_batch)          */
    int32_t threads;            /* host worker threads; each owns games/threads trees and its own batches      */
    int32_t nodes;              /* visits per move ("selfplay_nodes")                                          */
    int32_t leaves_per_tree;    /* positions of one tree in flight per batch (1 = the reference's schedule)    */
    float   cpuct;              /* "cpuct"                                                                     */
    float   noise_weight;       /* "mcts_noise_weight"                                                         */
    float   alpha_initial, alpha_decay, alpha_final;   /* "selfplay_alpha_*" (selfplay.cpp:73-76)              */
    int32_t alpha_cutoff;
    int32_t draw_value_pct;     /* "draw_value_pct" (selfplay.cpp:71)                                          */
    uint32_t seed;
    int32_t pipeline;           /* 1 or 2: each worker keeps TWO sets of its trees in flight through the engine's queue
                                   (kh_submit_encode_infer_legal / kh_wait): one set on the device while the other is
                                   expanded and selected; 3, 4: that many sets (more, smaller launches in flight);
                                   0: one blocking call per round (the reference's schedule)                            */
    int32_t coalesce_target;    /* pipeline: kh_set_coalesce(engine, target, wait): positions per launch to wait for   */
    int32_t coalesce_wait_us;   /*           ... and for how long at most                                               */
    int32_t reserved[1];
} ks_pool_config;

typedef struct ks_pool_stats {
    int64_t evals;              /* leaf evaluations through the engine                                         */
    int64_t batches;            /* engine calls                                                                */
    int64_t moves;              /* plies played                                                                */
    int64_t games_finished, white_wins, black_wins, draws;
    int64_t records;            /* replay records produced                                                     */
    double  seconds, evals_per_s, mean_batch;
    double  engine_seconds;     /* time the workers spent inside kh_encode_infer_legal, summed over workers     */
} ks_pool_stats;

/* one finished-game position: replaybuffer.h:20-22 holds OBSIZE + PSIZE + 1 floats (26 372 B) for this */
#define KS_MAX_RECORD_ACTIONS 96
typedef struct ks_record {
    kh_board board;             /* the observation, compact                                                    */
    float    value;             /* training target (selfplay.cpp:176-184)                                      */
    int32_t  nact;
    int16_t  actions[KS_MAX_RECORD_ACTIONS];
    float    visits[KS_MAX_RECORD_ACTIONS];   /* MCTS::snapshot, sparse (mcts.h:341-348)                       */
} ks_record;

typedef struct ks_pool ks_pool;
int  ks_pool_create(kh_engine* engine, const ks_pool_config* cfg, ks_pool** out);
/* One process, one pool, SEVERAL engines — one evaluator per GPU of the node (kh_config.device), the reference's own shape
 * taken to N GPUs: its inference threads share one model and one replay ring in one process (selfplay.cpp:21-35,96-109,
 * replaybuffer.h:36-56).  Worker t feeds engines[t % n] (threads >= n), finished games of every worker land in the pool's
 * one ring (ks_pool_drain_records), ks_pool_publish_weights installs a new generation on every engine (selfplay.cpp:
 * 282-283).  No data-path collective: leaf evaluations are independent. */
int  ks_pool_create_multi(kh_engine* const* engines, int n_engines, const ks_pool_config* cfg, ks_pool** out);
int  ks_pool_publish_weights(ks_pool* p, const float* blob, size_t nfloats, int generation);
/* play until at least min_evals leaf evaluations were made or max_seconds passed; cumulative stats */
int  ks_pool_run(ks_pool* p, int64_t min_evals, double max_seconds, ks_pool_stats* stats);
int64_t ks_pool_drain_records(ks_pool* p, ks_record* out, int64_t cap);
void ks_pool_destroy(ks_pool* p);

const char* ks_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* KAMI_SEARCH_H */
