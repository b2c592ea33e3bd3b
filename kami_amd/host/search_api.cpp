// search_api.cpp — libkamisearch.so: C ABI over env.h / mcts.h and the self-play pool that feeds the
// engine (include/kami_search.h).  Host code only; the one device entry point it uses is
// kh_encode_infer_legal.
#include "kami_search.h"
#include "mcts.h"

#include <algorithm>
#include <atomic>
#include <sched.h>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <string>
#include <set>
#include <thread>

using namespace kami;

namespace {
thread_local std::string g_err;
int fail(const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

// the synthetic evaluator of the reference harness (`kami_ref mcts`, test infrastructure)
uint64_t fnv1a(const std::string& s)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
uint64_t splitmix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}
}  // namespace

extern "C" {

const char* ks_last_error(void) { return g_err.c_str(); }

int ks_perft(const char* fen, int depth, uint64_t* nodes)
{
    chess::Position p;
    if (!chess::Position::from_fen(fen, p)) return fail("bad FEN");
    *nodes = chess::perft(p, depth);
    return 0;
}

int ks_fen_actions(const char* fen, int32_t* out, int cap)
{
    chess::Position p;
    if (!chess::Position::from_fen(fen, p)) { fail("bad FEN"); return -1; }
    chess::Move mv[chess::MAX_MOVES];
    const int n = p.legal(mv);
    std::vector<int> a;
    for (int i = 0; i < n; ++i) a.push_back(chess::encode_action(p, mv[i]));
    std::sort(a.begin(), a.end());
    for (int i = 0; i < n && i < cap; ++i) out[i] = a[i];
    return n;
}

struct ks_env { Env env; };
ks_env* ks_env_new(void) { return new ks_env(); }
void ks_env_free(ks_env* e) { delete e; }
int ks_env_ply(ks_env* e) { return e->env.ply(); }
int ks_env_actions(ks_env* e, int32_t* out, int cap)
{
    const std::vector<int>& a = e->env.actions();
    for (size_t i = 0; i < a.size() && (int)i < cap; ++i) out[i] = a[i];
    return (int)a.size();
}
int ks_env_push(ks_env* e, int action)
{
    bool legal = false;
    for (int a : e->env.actions()) legal |= a == action;
    if (!legal) return fail("action %d is not legal here", action);
    e->env.push(action);
    return 0;
}
int ks_env_pop(ks_env* e)
{
    if (e->env.ply() == 0) return fail("pop at the initial position");
    e->env.pop();
    return 0;
}
int ks_env_terminal(ks_env* e, float* value) { return e->env.terminal(value) ? 1 : 0; }
float ks_env_turn(ks_env* e) { return e->env.turn(); }
int ks_env_fen(ks_env* e, char* buf, int cap)
{
    const std::string s = e->env.print();
    snprintf(buf, cap, "%s", s.c_str());
    return (int)s.size();
}
void ks_env_record(ks_env* e, kh_board* out) { e->env.record(out); }

int ks_mcts_synthetic(int nodes, int nmoves, int leaves, const int32_t* picks, int npicks, char* out, int cap)
{
    try {
        MCTSConfig cfg;
        cfg.mcts_noise_weight = 0.0f;
        MCTS tree(cfg);
        std::string text;
        char line[256];
        std::vector<float> policy(PSIZE);
        std::vector<MCTS::Leaf> batch((size_t)(leaves > 1 ? leaves : 1));
        for (int m = 0; m < nmoves; ++m) {
            while (tree.n() < nodes) {
                // up to `leaves` positions in flight, never more than the visits still missing
                int nb = 0;
                while (nb < (int)batch.size() && tree.n() + nb < nodes) {
                    bool blocked = false;
                    if (tree.select_leaf(&batch[nb], &blocked)) ++nb;
                    else if (blocked) break;
                    else if (tree.n() + nb >= nodes) break;         // a terminal visit was counted
                }
                for (int j = 0; j < nb; ++j) {
                    // the evaluator sees the leaf position: rebuild its FEN by replaying the path
                    std::vector<int> path;
                    for (size_t d = 1; d < batch[j].path.size(); ++d) path.push_back(batch[j].path[d]->action);
                    Env& e = tree.get_env();
                    for (int a : path) e.push(a);
                    const uint64_t h = fnv1a(e.print());
                    for (size_t k = 0; k < path.size(); ++k) e.pop();
                    double sum = 0.0;
                    for (int a = 0; a < PSIZE; ++a) { policy[a] = (float)(splitmix(h + (uint64_t)a) % 16777213ull + 1); sum += policy[a]; }
                    for (int a = 0; a < PSIZE; ++a) policy[a] = (float)(policy[a] / sum);
                    const float value = ((float)(splitmix(h ^ 0x7777) % 2001) - 1000.0f) / 1000.0f;
                    float ptotal = 0.0f;
                    for (int a : batch[j].actions) ptotal += policy[a];
                    std::vector<float> pr(batch[j].actions.size());
                    for (size_t i = 0; i < pr.size(); ++i) pr[i] = policy[batch[j].actions[i]] / ptotal;
                    tree.expand_leaf(batch[j], pr.data(), value);
                }
            }
            snprintf(line, sizeof(line), "move %d fen %s\n", m, tree.get_env().print().c_str()); text += line;
            snprintf(line, sizeof(line), "root n %d w %.9g\n", tree.root->n, tree.root->w); text += line;
            for (const Node& c : tree.root->children()) {
                snprintf(line, sizeof(line), "child %d n %d w %.9g p %.9g\n", c.action, c.n, c.w, c.p); text += line;
            }
            const int picked = m < npicks ? picks[m] : tree.pick(0.0f);
            snprintf(line, sizeof(line), "pick %d\n", picked); text += line;
            tree.push(picked);
            float v;
            if (tree.get_env().terminal(&v)) { snprintf(line, sizeof(line), "terminal %g\n", v); text += line; break; }
        }
        if ((int)text.size() + 1 > cap) return fail("output needs %zu bytes", text.size() + 1);
        memcpy(out, text.c_str(), text.size() + 1);
        return 0;
    } catch (std::exception& e) {
        return fail("%s", e.what());
    }
}

// ------------------------------------------------------------------------------------------------
// Self-play pool: Selfplay::inference_main (selfplay.cpp:58-213) with heap trees, several leaves per
// tree and batch, compact observations in, legal priors out.
struct ks_pool {
    std::vector<kh_engine*> engines;    // one evaluator per GPU (ks_pool_create_multi); worker t feeds engines[t % n]
    ks_pool_config cfg;
    struct Game {
        std::unique_ptr<MCTS> tree;
        struct Step { kh_board board; std::vector<int> actions; std::vector<float> visits; float pov; };
        std::vector<Step> trajectory;
        std::vector<MCTS::Leaf> leaves;     // leaves[0 .. nleaves) are in flight (the objects are kept: their vectors keep their memory)
        int nleaves = 0;
    };
    std::vector<Game> games;
    std::mutex rec_mutex;
    std::vector<ks_record> records;
    std::atomic<int64_t> engine_ns{ 0 };
    std::atomic<int64_t> evals{ 0 }, batches{ 0 }, moves{ 0 }, finished{ 0 }, wwins{ 0 }, bwins{ 0 }, draws{ 0 }, nrecords{ 0 };
    double seconds = 0.0;
    std::string error;
    std::mutex err_mutex;
};

namespace {

void finish_game(ks_pool* p, ks_pool::Game& g, float value)
{
    const float draw_value = (p->cfg.draw_value_pct / 100.0f) * 2.0f - 1.0f;      // selfplay.cpp:71
    std::vector<ks_record> out;
    out.reserve(g.trajectory.size());
    for (auto& t : g.trajectory) {
        ks_record r;
        memset(&r, 0, sizeof(r));
        r.board = t.board;
        r.value = value == 0.0f ? draw_value : t.pov * value;                       // selfplay.cpp:176-184
        // keep the most visited KS_MAX_RECORD_ACTIONS moves (positions with more legal moves are rare)
        std::vector<int> idx(t.actions.size());
        for (size_t i = 0; i < idx.size(); ++i) idx[i] = (int)i;
        if (idx.size() > KS_MAX_RECORD_ACTIONS) {
            std::partial_sort(idx.begin(), idx.begin() + KS_MAX_RECORD_ACTIONS, idx.end(), [&](int a, int b) { return t.visits[a] > t.visits[b]; });
            idx.resize(KS_MAX_RECORD_ACTIONS);
            std::sort(idx.begin(), idx.end());
        }
        r.nact = (int32_t)idx.size();
        for (size_t i = 0; i < idx.size(); ++i) { r.actions[i] = (int16_t)t.actions[idx[i]]; r.visits[i] = t.visits[idx[i]]; }
        out.push_back(r);
    }
    {
        std::lock_guard<std::mutex> lk(p->rec_mutex);
        p->records.insert(p->records.end(), out.begin(), out.end());
    }
    p->nrecords += (int64_t)out.size();
    p->finished += 1;
    if (value > 0) p->wwins += 1; else if (value < 0) p->bwins += 1; else p->draws += 1;
    g.trajectory.clear();
    g.tree->reset();
}

// the tree has its visits: record the position, play a move (selfplay.cpp:131-189)
void advance_game(ks_pool* p, ks_pool::Game& g)
{
    MCTS& tree = *g.tree;
    Env& env = tree.get_env();
    ks_pool::Game::Step st;
    env.record(&st.board);
    tree.snapshot_sparse(st.actions, st.visits);
    st.pov = -env.turn();                                                           // selfplay.cpp:146
    g.trajectory.push_back(std::move(st));
    float alpha = p->cfg.alpha_final;
    if (env.ply() < p->cfg.alpha_cutoff) alpha = std::pow(p->cfg.alpha_decay, (float)env.ply()) * p->cfg.alpha_initial;
    tree.push(tree.pick(alpha));
    p->moves += 1;
    float value;
    if (env.terminal(&value)) finish_game(p, g, value);
}

// One batch of a worker: the leaves of a contiguous range of its trees (up to L per tree), as the engine wants them.
struct LeafSet {
    int g0 = 0, g1 = 0;
    std::vector<kh_board> boards;
    std::vector<int32_t> offsets, actions;
    std::vector<float> priors, values;
    std::vector<std::pair<int, int>> owner;         // (game, leaf slot) of each batch row
    int64_t ticket = 0;
    bool in_flight = false;
    std::chrono::steady_clock::time_point t_submit;

    // select -> records + legal actions (selfplay.cpp:113-193); trees that reached their visit budget play a move first
    int build(ks_pool* p, int L)
    {
        boards.clear(); offsets.assign(1, 0); actions.clear(); owner.clear();
        for (int gi = g0; gi < g1; ++gi) {
            ks_pool::Game& g = p->games[gi];
            MCTS& tree = *g.tree;
            if (gi + 2 < g1) p->games[gi + 2].tree->prefetch();     // a worker's trees do not fit its caches
            if ((int)g.leaves.size() < L) g.leaves.resize((size_t)L);
            g.nleaves = 0;
            for (;;) {
                if (g.nleaves == 0 && tree.n() >= p->cfg.nodes) { advance_game(p, g); continue; }
                if (g.nleaves >= L || tree.n() + g.nleaves >= p->cfg.nodes) break;
                bool blocked = false;
                if (tree.select_leaf(&g.leaves[g.nleaves], &blocked)) { ++g.nleaves; continue; }
                if (blocked) break;
            }
            for (int j = 0; j < g.nleaves; ++j) {
                boards.push_back(g.leaves[j].record);
                actions.insert(actions.end(), g.leaves[j].actions.begin(), g.leaves[j].actions.end());
                offsets.push_back((int32_t)actions.size());
                owner.emplace_back(gi, j);
            }
        }
        priors.resize(actions.size() + 1);
        values.resize(boards.size() + 1);
        return (int)boards.size();
    }

    // expand with the evaluator's answer (selfplay.cpp:199-200)
    void expand(ks_pool* p)
    {
        const int nb = (int)boards.size();
        for (int j = 0; j < nb; ++j) {
            if (j + 2 < nb) { ks_pool::Game& a = p->games[owner[j + 2].first]; a.tree->prefetch_expand(a.leaves[owner[j + 2].second]); }
            ks_pool::Game& g = p->games[owner[j].first];
            g.tree->expand_leaf(g.leaves[owner[j].second], priors.data() + offsets[j], values[j]);
            if (owner[j].second + 1 == g.nleaves) g.nleaves = 0;       // nleaves > 0 <=> leaves marked in the tree
        }
        p->evals += nb;
        p->batches += 1;
    }

    void release(ks_pool* p)
    {
        for (int gi = g0; gi < g1; ++gi) {
            ks_pool::Game& g = p->games[gi];
            for (int j = 0; j < g.nleaves; ++j) g.tree->release_leaf(g.leaves[j]);
            g.nleaves = 0;
        }
    }
};

void worker(ks_pool* p, kh_engine* engine, int g0, int g1, int64_t target_evals, double deadline_s, std::chrono::steady_clock::time_point t0)
{
    const int L = p->cfg.leaves_per_tree > 0 ? p->cfg.leaves_per_tree : 1;
    // pipeline: the worker's trees in two (or more) sets, each set one submission to the engine's queue — while one set is
    // on the device the others are expanded and selected (kh_submit_encode_infer_legal / kh_wait); otherwise one blocking
    // call per round over all of its trees (the reference's schedule, selfplay.cpp:196)
    // (pipeline = 1 means two sets; 2..4 that many: more launches in flight, each smaller)
    constexpr int MAX_SETS = 4;
    const int want_sets = p->cfg.pipeline <= 0 ? 1 : std::min(MAX_SETS, std::max(2, p->cfg.pipeline));
    const int nsets = std::max(1, std::min(want_sets, g1 - g0));
    const bool pipeline = nsets >= 2;
    LeafSet sets[MAX_SETS];
    for (int k = 0; k < MAX_SETS; ++k) {
        sets[k].g0 = g0 + (int)((int64_t)(g1 - g0) * std::min(k, nsets) / nsets);
        sets[k].g1 = g0 + (int)((int64_t)(g1 - g0) * std::min(k + 1, nsets) / nsets);
    }
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto stop = [&] {
        return p->evals.load() >= target_evals || std::chrono::duration<double>(now() - t0).count() > deadline_s;
    };
    auto finish = [&](LeafSet& s) {                      // wait for a submitted set and expand it
        const int rc = kh_wait(engine, s.ticket);
        s.in_flight = false;
        p->engine_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(now() - s.t_submit).count();
        if (rc != KH_OK) throw std::runtime_error(std::string("kh_encode_infer_legal: ") + kh_last_error());
        s.expand(p);
    };
    try {
        if (!pipeline) {
            LeafSet& s = sets[0];
            while (!stop()) {
                const int nb = s.build(p, L);
                if (nb == 0) continue;
                const auto e0 = now();
                const int rc = kh_encode_infer_legal(engine, s.boards.data(), nb, s.offsets.data(), s.actions.data(), s.priors.data(), s.values.data());
                p->engine_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(now() - e0).count();
                if (rc != KH_OK) throw std::runtime_error(std::string("kh_encode_infer_legal: ") + kh_last_error());
                s.expand(p);
            }
        } else {
            for (int k = 0;;) {
                // the next set: one that is not in flight (start-up), else whichever of this worker's submissions has
                // come back — launches do not complete in submission order once several are on the device, and a worker
                // that waits for its oldest ticket leaves finished sets lying (three / four sets per worker lost half
                // their rate to that)
                int pick = -1;
                for (int j = 0; j < nsets && pick < 0; ++j)
                    if (!sets[(k + j) % nsets].in_flight) pick = (k + j) % nsets;
                for (unsigned spin = 0; pick < 0; ++spin) {
                    for (int j = 0; j < nsets && pick < 0; ++j) {
                        LeafSet& q = sets[(k + j) % nsets];
                        __builtin_ia32_pause();
                        int done = 0;
                        const int rc = kh_try_wait(engine, q.ticket, &done);
                        if (!done && rc != KH_OK) throw std::runtime_error(std::string("kh_try_wait: ") + kh_last_error());
                        if (!done) continue;
                        q.in_flight = false;
                        p->engine_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(now() - q.t_submit).count();
                        if (rc != KH_OK) throw std::runtime_error(std::string("kh_encode_infer_legal: ") + kh_last_error());
                        q.expand(p);
                        pick = (k + j) % nsets;
                    }
                    if (pick < 0) {
                        __builtin_ia32_pause();
                        if ((spin & 63) == 63) sched_yield();
                        if ((spin & 0xfffff) == 0xfffff && stop()) {     // an engine that takes seconds per launch: block on the oldest
                            finish(sets[k]);
                            pick = k;
                        }
                    }
                }
                k = pick;
                LeafSet& s = sets[k];
                if (stop()) break;
                const int nb = s.build(p, L);
                if (nb == 0) { k = (k + 1) % nsets; continue; }
                s.t_submit = now();
                const int rc = kh_submit_encode_infer_legal(engine, s.boards.data(), nb, s.offsets.data(), s.actions.data(), s.priors.data(),
                                                            s.values.data(), &s.ticket);
                if (rc != KH_OK) throw std::runtime_error(std::string("kh_encode_infer_legal: ") + kh_last_error());
                s.in_flight = true;
                k = (k + 1) % nsets;
            }
            for (auto& s : sets)
                if (s.in_flight) finish(s);
        }
    } catch (std::exception& e) {
        // a failed engine call must not leave virtual visits or tickets behind: the pool can be run again
        for (auto& s : sets) {
            if (s.in_flight) { (void)kh_wait(engine, s.ticket); s.in_flight = false; }
            s.release(p);
        }
        std::lock_guard<std::mutex> lk(p->err_mutex);
        p->error = e.what();
    }
}

}  // namespace

int ks_pool_create(kh_engine* engine, const ks_pool_config* cfg, ks_pool** out)
{
    return ks_pool_create_multi(&engine, 1, cfg, out);
}

int ks_pool_create_multi(kh_engine* const* engines, int n_engines, const ks_pool_config* cfg, ks_pool** out)
{
    if (!engines || n_engines < 1 || !cfg || !out) return fail("null argument");
    for (int i = 0; i < n_engines; ++i)
        if (!engines[i]) return fail("null engine");
    if (cfg->games < 1 || cfg->threads < 1 || cfg->nodes < 2) return fail("games >= 1, threads >= 1, nodes >= 2 required");
    if (n_engines > std::min(cfg->threads, cfg->games)) return fail("%d engines but only %d workers: every engine needs a worker of its own", n_engines, std::min(cfg->threads, cfg->games));
    ks_pool* p = new ks_pool();
    p->engines.assign(engines, engines + n_engines);
    p->cfg = *cfg;
    if (cfg->pipeline) {
        // The queue's limits (include/kami_hip.h: a submission holds at most 512 positions, an engine at most
        // KH_MAX_OUTSTANDING un-waited tickets), checked here instead of failing inside every worker's first round.
        const int T = std::min(cfg->threads, cfg->games), sets = std::min(4, std::max(2, cfg->pipeline));
        const int per_set = ((cfg->games + T - 1) / T + sets - 1) / sets * std::max(1, cfg->leaves_per_tree);
        if (per_set > 512) {
            delete p;
            return fail("pipelined pool: %d games / %d threads / %d sets x %d leaves per tree = %d positions per submission, the queue takes 512: "
                        "more threads or sets, fewer leaves per tree, or pipeline = 0", cfg->games, T, sets, cfg->leaves_per_tree, per_set);
        }
        if (T * sets > KH_MAX_OUTSTANDING) {
            delete p;
            return fail("pipelined pool: %d threads x %d sets in flight exceed the engine's %d outstanding tickets", T, sets, KH_MAX_OUTSTANDING);
        }
        if (cfg->coalesce_target < 0 || cfg->coalesce_target > 1024 || cfg->coalesce_wait_us < 0 || cfg->coalesce_wait_us > 1000000) {
            delete p;
            return fail("coalesce_target in [0, 1024], coalesce_wait_us in [0, 1000000]");
        }
    }
    p->games.resize((size_t)cfg->games);
    for (int i = 0; i < cfg->games; ++i) {
        MCTSConfig mc;
        mc.cpuct = cfg->cpuct > 0 ? cfg->cpuct : 1.0f;
        mc.mcts_noise_weight = cfg->noise_weight;
        mc.seed = cfg->seed * 2654435761u + (unsigned)i;
        p->games[i].tree.reset(new MCTS(mc));
    }
    *out = p;
    return 0;
}

// engines that a pipelined pool is running on right now: the queue's merge settings are per ENGINE, so two such pools
// on one engine would overwrite each other's
static std::mutex g_pipe_mu;
static std::set<const kh_engine*> g_pipe_engines;

int ks_pool_run(ks_pool* p, int64_t min_evals, double max_seconds, ks_pool_stats* stats)
{
    const auto t0 = std::chrono::steady_clock::now();
    const int T = std::min(p->cfg.threads, p->cfg.games);
    p->error.clear();
    // The engine's merge settings belong to this run only: set here, put back to "launch at once" before returning (a
    // later synchronous small call must not wait out this pool's quiet period, and ks_pool_destroy never touches the
    // engine: it may be gone by then).
    struct PipeGuard {
        std::vector<kh_engine*> es;
        ~PipeGuard()
        {
            for (kh_engine* e : es) {
                (void)kh_set_coalesce_callers(e, 0);
                (void)kh_set_coalesce(e, 0, 0);
            }
            std::lock_guard<std::mutex> lk(g_pipe_mu);
            for (kh_engine* e : es) g_pipe_engines.erase(e);
        }
    } guard;
    const int NE = (int)p->engines.size();
    if (p->cfg.pipeline) {
        {
            std::lock_guard<std::mutex> lk(g_pipe_mu);
            for (kh_engine* e : p->engines)
                if (g_pipe_engines.count(e)) return fail("another pipelined pool is running on one of these engines (the queue's merge settings are per engine)");
            for (kh_engine* e : p->engines) { g_pipe_engines.insert(e); guard.es.push_back(e); }
        }
        for (int i = 0; i < NE; ++i) {
            if (kh_set_coalesce(p->engines[i], p->cfg.coalesce_target, p->cfg.coalesce_wait_us) != KH_OK) return fail("%s", kh_last_error());
            // A launch goes as soon as it holds a submission of HALF the engine's workers (round 2: of every worker, "a
            // whole round" — but then every launch waits for the slowest worker, 30-40 us of a ~100 us cycle; with the
            // engine's one dispatcher keeping several launches on the device the early half need not wait for the late
            // one: configs[1] literally 4.6-5.0 -> 5.2-6.0 M/s, four leaves per tree 6.9 -> 9.6 M/s, same box)
            if (p->cfg.coalesce_target > 0) (void)kh_set_coalesce_callers(p->engines[i], ((T - i + NE - 1) / NE + 1) / 2);
        }
    }
    const int64_t target = p->evals.load() + min_evals;
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) {
        const int g0 = (int)((int64_t)p->cfg.games * t / T), g1 = (int)((int64_t)p->cfg.games * (t + 1) / T);
        th.emplace_back(worker, p, p->engines[(size_t)t % p->engines.size()], g0, g1, target, max_seconds, t0);
    }
    for (auto& x : th) x.join();
    p->seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    // leaves still marked in flight belong to nobody now: none — every batch is expanded before a worker leaves
    if (!p->error.empty()) return fail("%s", p->error.c_str());
    if (stats) {
        stats->evals = p->evals; stats->batches = p->batches; stats->moves = p->moves;
        stats->games_finished = p->finished; stats->white_wins = p->wwins; stats->black_wins = p->bwins; stats->draws = p->draws;
        stats->records = p->nrecords;
        stats->seconds = p->seconds;
        stats->evals_per_s = p->seconds > 0 ? (double)p->evals / p->seconds : 0.0;
        stats->mean_batch = p->batches ? (double)p->evals / (double)p->batches : 0.0;
        stats->engine_seconds = 1e-9 * (double)p->engine_ns;
    }
    return 0;
}

int64_t ks_pool_drain_records(ks_pool* p, ks_record* out, int64_t cap)
{
    std::lock_guard<std::mutex> lk(p->rec_mutex);
    const int64_t n = std::min<int64_t>(cap, (int64_t)p->records.size());
    if (n > 0 && out) memcpy(out, p->records.data(), (size_t)n * sizeof(ks_record));
    p->records.erase(p->records.begin(), p->records.begin() + n);
    return n;
}

int ks_pool_publish_weights(ks_pool* p, const float* blob, size_t nfloats, int generation)
{
    if (!p || !blob) return fail("null argument");
    // selfplay.cpp:282-283 (model->read(path) after an accepted candidate) for every evaluator of the pool: each engine
    // swaps atomically, evaluations in flight finish on the weights they started with
    for (kh_engine* e : p->engines)
        if (kh_load_weights(e, blob, nfloats, generation) != KH_OK) return fail("%s", kh_last_error());
    return 0;
}

void ks_pool_destroy(ks_pool* p)
{
    delete p;       // (never touches the engine: it may have been destroyed before its pools)
}

}  // extern "C"
