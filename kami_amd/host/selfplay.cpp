// selfplay.cpp — kami::Selfplay on this repository's stack.  Behaviour and option keys follow
// kami/selfplay.cpp (cited per block); what differs is what SURVEY §8f asks for: trees on the heap,
// the observation is an 80-byte record, the network returns the legal moves' priors, a finished
// game's planes are produced by the device encoder in one batch.
#include "selfplay.h"
#include "env.h"
#include "evaluate.h"
#include "mcts.h"
#include "options.h"

#include <chrono>
#include <cmath>
#include <ctime>
#include <iostream>
#include <list>
#include <memory>
#include <stdexcept>
#include <thread>
#include <vector>

namespace kami {

namespace {

// one self-play game of an inference thread
struct Game {
    struct Step { kh_board board; std::vector<float> visits; float pov; };
    std::unique_ptr<MCTS> tree;
    std::vector<Step> trajectory;
    std::vector<std::string> moves;      // coordinate notation, for the pgn command
    int generation = 0;                  // of the model the tree was searched with
    MCTS::Leaf leaf;

    int drop()                           // forget the game so far; returns how many steps were dropped
    {
        const int n = (int)trajectory.size();
        tree->reset();
        trajectory.clear();
        moves.clear();
        return n;
    }
};

struct PlayOptions {                     // selfplay.cpp:61-76
    bool flush_old_trees = options::getInt("flush_old_trees", 1) != 0;
    float draw_value = (options::getInt("draw_value_pct", 50) / 100.0f) * 2.0f - 1.0f;
    float alpha_initial = options::getFloat("selfplay_alpha_initial", 1.0f);
    float alpha_decay = options::getFloat("selfplay_alpha_decay", 1.0f);
    float alpha_final = options::getFloat("selfplay_alpha_final", 1.0f);
    int alpha_cutoff = (int)options::getFloat("selfplay_alpha_cutoff", 1.0f);

    float alpha(int ply) const { return ply < alpha_cutoff ? std::pow(alpha_decay, (float)ply) * alpha_initial : alpha_final; }
};

MCTSConfig search_options()
{
    MCTSConfig c;                        // mcts.h:83-95
    c.cpuct = options::getFloat("cpuct", 1.0f);
    c.force_expand_unvisited = options::getInt("force_expand_unvisited", 0);
    c.unvisited_node_value_pct = options::getInt("unvisited_node_value_pct", 100);
    c.scale_cpuct_by_actions = options::getInt("scale_cpuct_by_actions", 0);
    c.mcts_noise_weight = options::getFloat("mcts_noise_weight", 0.05f);
    return c;
}

std::string movetext(const Game& g, Env& env, float value)
{
    // coordinate notation (the reference prints SAN through the thc library, env.h:432-476)
    std::string out;
    for (size_t m = 0; m < g.moves.size(); ++m) {
        if (m % 2 == 0) out += (m ? " " : "") + std::to_string(m / 2 + 1) + ".";
        out += " " + g.moves[m];
    }
    std::string reason;
    float unused;
    env.terminal_str(&unused, reason);
    return out + " " + (value < 0 ? "0-1" : (value > 0 ? "1-0" : "1/2-1/2")) + " {" + reason + "}";
}

void check(int rc)
{
    if (rc == KH_ERR_NAN_POLICY) throw std::runtime_error("inference policy output contains NaN");   // nn.cpp:176-180
    if (rc == KH_ERR_NAN_VALUE) throw std::runtime_error("inference value output contains NaN");
    if (rc != KH_OK) throw std::runtime_error(kh_last_error());
}

}  // namespace

struct Selfplay::Impl {
    Selfplay* self;
    NN* model;
    // Option "engine_devices" = N > 1: one evaluator per GPU in this one process — the reference's shape (N inference
    // threads sharing one ring, selfplay.cpp:21-35,96-109) taken across the node.  engines[0] is the caller's model (the
    // trainer's and the gate's), engines[d] a replica on device d; inference thread i feeds engines[i % N]; an accepted
    // candidate is published to every replica (selfplay.cpp:282-283).
    std::vector<std::unique_ptr<NN>> replicas;
    std::vector<NN*> engines;
    ReplayBuffer replay;
    int ibatch, nodes;
    std::vector<std::thread> inference, training;
    std::atomic<bool> wants_pgn{ false };
    std::string ret_pgn;
    std::list<std::atomic<int>> partial_trajectories;

    Impl(Selfplay* s, NN* m)             // selfplay.cpp:13-19
        : self(s), model(m), replay(OBSIZE, PSIZE, options::getInt("replaybuffer_size", 512)),
          ibatch(options::getInt("selfplay_batch", 16)), nodes(options::getInt("selfplay_nodes", 512))
    {
        engines.push_back(model);
        const int ndev = std::max(1, options::getInt("engine_devices", 1));
        for (int d = 1; d < ndev; ++d) {
            replicas.emplace_back(new NN(model, model->get_device().index + d));
            engines.push_back(replicas.back().get());
        }
        if (ndev > 1) std::cout << "Selfplay: " << ndev << " evaluators (engine_devices), inference thread i on engine i % " << ndev << std::endl;
    }

    bool running() { return self->status.code() == RUNNING; }

    void finish_game(NN* mine, Game& g, float value, const PlayOptions& po, std::vector<float>& planes)
    {
        Env& env = g.tree->get_env();
        if (wants_pgn.exchange(false)) ret_pgn = movetext(g, env, value);
        // the game's observations: one device encode for the whole trajectory, then selfplay.cpp:176-184
        const int n = (int)g.trajectory.size();
        std::vector<kh_board> boards((size_t)n);
        for (int k = 0; k < n; ++k) boards[k] = g.trajectory[k].board;
        planes.resize((size_t)n * OBSIZE);
        check(kh_encode(mine->handle(), boards.data(), n, planes.data()));
        for (int k = 0; k < n; ++k) {
            const Game::Step& st = g.trajectory[k];
            replay.add(planes.data() + (size_t)k * OBSIZE, st.visits.data(), value == 0.0f ? po.draw_value : st.pov * value);
        }
    }

    void inference_main(int id)          // selfplay.cpp:58-213
    {
        std::cout << "Starting inference thread: " << id << std::endl;
        NN* const mine = engines[(size_t)id % engines.size()];
        const PlayOptions po;
        MCTSConfig cfg = search_options();
        std::vector<Game> games((size_t)ibatch);
        for (int i = 0; i < ibatch; ++i) {
            cfg.seed = (unsigned)time(nullptr) * 2654435761u + (unsigned)(id * 100003 + i);
            games[i].tree.reset(new MCTS(cfg));
            games[i].generation = mine->get_generation();
        }
        std::vector<kh_board> boards;
        std::vector<int32_t> offsets, actions;
        std::vector<float> priors, values, planes;
        std::vector<Game*> owner;
        auto partials = partial_trajectories.begin();
        std::advance(partials, id);
        int open_steps = 0;
        try {
            while (running()) {
                boards.clear(); actions.clear(); offsets.assign(1, 0); owner.clear();
                for (Game& g : games) {
                    MCTS& tree = *g.tree;
                    for (;;) {
                        if (po.flush_old_trees && g.generation < mine->get_generation()) {       // selfplay.cpp:115-127
                            open_steps -= g.drop();
                            g.generation = mine->get_generation();
                        }
                        bool leaf = false;
                        while (tree.n() < nodes && !(leaf = tree.select_leaf(&g.leaf))) {}        // selfplay.cpp:130
                        if (leaf) break;
                        // the tree has its visits: save the position, play a move (selfplay.cpp:135-160)
                        Env& env = tree.get_env();
                        Game::Step st;
                        env.record(&st.board);
                        st.visits.resize(PSIZE);
                        tree.snapshot(st.visits.data());
                        st.pov = -env.turn();
                        g.trajectory.push_back(std::move(st));
                        ++open_steps;
                        const int picked = tree.pick(po.alpha(env.ply()));
                        g.moves.push_back(env.debug_action(picked));
                        tree.push(picked);
                        float value;
                        if (env.terminal(&value)) {                                               // selfplay.cpp:163-189
                            finish_game(mine, g, value, po, planes);
                            open_steps -= g.drop();
                        }
                    }
                    boards.push_back(g.leaf.record);
                    actions.insert(actions.end(), g.leaf.actions.begin(), g.leaf.actions.end());
                    offsets.push_back((int32_t)actions.size());
                    owner.push_back(&g);
                }
                // the batch: one leaf of every tree (selfplay.cpp:196-200), as records + legal actions
                priors.resize(actions.size());
                values.resize(boards.size());
                check(kh_encode_infer_legal(mine->handle(), boards.data(), (int)boards.size(), offsets.data(), actions.data(),
                                            priors.data(), values.data()));
                for (size_t j = 0; j < owner.size(); ++j) owner[j]->tree->expand_leaf(owner[j]->leaf, priors.data() + offsets[j], values[j]);
                *partials = open_steps;                                                           // selfplay.cpp:203-205
            }
        } catch (std::exception& e) {
            std::cerr << "INFER " << id << ": " << e.what() << std::endl;
        }
        std::cout << "Terminating inference thread: " << id << std::endl;
    }

    void report_progress(long from, long target)
    {
        std::cout << "Gen " << model->get_generation() << " RPB " << 100 * (replay.count() - from) / (target - from) << "% ["
                  << replay.count() - from << " / " << target - from << "] | Partials: ";
        int inf = 0;
        for (auto& ct : partial_trajectories) std::cout << " Inf " << inf++ << ": " << ct;
        std::cout << std::endl;
    }

    void training_main(int id)           // selfplay.cpp:215-304
    {
        std::cout << "TRAIN " << id << ": starting thread " << id << std::endl;
        const std::string modelpath = options::getStr("model_path", "/tmp/model.pt");
        const long ring = replay.size();
        const long step = ring * options::getInt("rpb_train_pct", 40) / 100;              // experiences between attempts
        const int samples = (int)(ring * options::getInt("training_sample_pct", 60) / 100);
        const bool detect_anomaly = options::getInt("training_detect_anomaly", 0);
        std::vector<float> inputs((size_t)samples * OBSIZE), mcts((size_t)samples * PSIZE), results((size_t)samples);
        long target = ring, from = 0;
        while (running()) {
            if (replay.count() < target) {
                if (id == 0) report_progress(from, target);
                std::this_thread::sleep_for(std::chrono::milliseconds(1000));
                continue;
            }
            std::cout << "TRAIN " << id << ": training generation " << model->get_generation() << " with " << samples
                      << " trajectories sampled from last " << ring << std::endl;
            NN candidate(model);                                                           // selfplay.cpp:259
            replay.select_batch(inputs.data(), mcts.data(), results.data(), samples);
            candidate.train(samples, inputs.data(), mcts.data(), results.data(), detect_anomaly);
            bool accepted = false;
            try {
                accepted = eval(model, &candidate, id);
            } catch (std::exception& e) {
                std::cerr << "TRAIN " << id << ": evaluation failed: " << e.what() << std::endl;
            }
            from = replay.count();
            if (!accepted) {
                std::cout << "TRAIN " << id << ": candidate rejected: generation remains " << model->get_generation() << std::endl;
                target += step;
                continue;
            }
            candidate.write(modelpath);                                                    // selfplay.cpp:282-283
            model->read(modelpath);
            for (auto& r : replicas) r->sync_from(model);                                  // every evaluator serves the new generation
            std::cout << "TRAIN " << id << ": candidate accepted: using new generation " << model->get_generation() << std::endl;
            if (options::getInt("flush_old_rpb", 1)) replay.clear();
            from = replay.count();
            target = std::max(ring, replay.count() + step);
        }
        std::cout << "TRAIN " << id << ": stopping thread" << std::endl;
    }
};

Selfplay::Selfplay(NN* model) : impl(new Impl(this, model)) {}
Selfplay::~Selfplay() = default;

ReplayBuffer& Selfplay::get_rbuf() { return impl->replay; }

std::string Selfplay::get_next_pgn()     // selfplay.h:73-80
{
    impl->wants_pgn = true;
    while (impl->wants_pgn) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    return impl->ret_pgn;
}

void Selfplay::start()                   // selfplay.cpp:21-35
{
    status.code(RUNNING);
    const int n_inference = options::getInt("inference_threads", 1);
    for (int i = 0; i < n_inference; ++i) impl->partial_trajectories.emplace_back(0);
    for (int i = 0; i < n_inference; ++i) impl->inference.emplace_back(&Impl::inference_main, impl.get(), i);
    for (int i = 0; i < options::getInt("training_threads", 1); ++i) impl->training.emplace_back(&Impl::training_main, impl.get(), i);
}

void Selfplay::stop()                    // selfplay.cpp:37-56
{
    if (status.code() != RUNNING) throw std::runtime_error("stop() called when not running");
    status.code(WAITING);
    for (auto& t : impl->inference) t.join();
    for (auto& t : impl->training) t.join();
    impl->inference.clear();
    impl->training.clear();
    status.code(STOPPED);
}

}  // namespace kami
