// selfplay.cpp — kami::Selfplay on this repository's stack.  Structure and option keys follow
// kami/selfplay.cpp (cited per block); the differences are the ones SURVEY §8f asks for: trees on the
// heap, the observation is an 80-byte record, the network returns the legal moves' priors, finished
// games' planes are produced by the device encoder in one batch.
#include "selfplay.h"
#include "env.h"
#include "evaluate.h"
#include "mcts.h"
#include "options.h"

#include <cmath>
#include <iostream>
#include <memory>
#include <stdexcept>

using namespace kami;
using namespace std;

Selfplay::Selfplay(NN* model) :                                              // selfplay.cpp:13-19
    model(model),
    replay_buffer(OBSIZE, PSIZE, options::getInt("replaybuffer_size", 512)),
    ibatch(options::getInt("selfplay_batch", 16)),
    nodes(options::getInt("selfplay_nodes", 512)),
    wants_pgn(false) {}

void Selfplay::start()                                                       // selfplay.cpp:21-35
{
    status.code(RUNNING);
    const int n_inference = options::getInt("inference_threads", 1);
    for (int i = 0; i < n_inference; ++i) {
        partial_trajectories.emplace_back(0);
        inference.push_back(thread(&Selfplay::inference_main, this, i));
    }
    for (int i = 0; i < options::getInt("training_threads", 1); ++i)
        training.push_back(thread(&Selfplay::training_main, this, i));
}

void Selfplay::stop()                                                        // selfplay.cpp:37-56
{
    if (status.code() != RUNNING) throw runtime_error("stop() called when not running");
    status.code(WAITING);
    for (auto& t : inference) t.join();
    inference.clear();
    for (auto& t : training) t.join();
    training.clear();
    status.code(STOPPED);
}

void Selfplay::inference_main(int id)                                        // selfplay.cpp:58-213
{
    cout << "Starting inference thread: " << id << endl;
    const bool flush_old_trees = options::getInt("flush_old_trees", 1);
    const float draw_value = (options::getInt("draw_value_pct", 50) / 100.0f) * 2.0f - 1.0f;
    const float alpha_initial = options::getFloat("selfplay_alpha_initial", 1.0f);
    const float alpha_decay = options::getFloat("selfplay_alpha_decay", 1.0f);
    const float alpha_final = options::getFloat("selfplay_alpha_final", 1.0f);
    const int alpha_cutoff = (int)options::getFloat("selfplay_alpha_cutoff", 1.0f);

    struct Step { kh_board board; vector<float> mcts; float pov; };
    struct Game {
        unique_ptr<MCTS> tree;
        vector<Step> trajectory;
        vector<string> moves;           // for the pgn command
        int source_generation = 0;
        MCTS::Leaf leaf;
        bool waiting = false;
    };
    MCTSConfig cfg;
    cfg.cpuct = options::getFloat("cpuct", 1.0f);
    cfg.force_expand_unvisited = options::getInt("force_expand_unvisited", 0);
    cfg.unvisited_node_value_pct = options::getInt("unvisited_node_value_pct", 100);
    cfg.scale_cpuct_by_actions = options::getInt("scale_cpuct_by_actions", 0);
    cfg.mcts_noise_weight = options::getFloat("mcts_noise_weight", 0.05f);
    vector<Game> games((size_t)ibatch);
    for (int i = 0; i < ibatch; ++i) {
        cfg.seed = (unsigned)time(nullptr) * 2654435761u + (unsigned)(id * 100003 + i);
        games[i].tree.reset(new MCTS(cfg));
        games[i].source_generation = model->get_generation();
    }
    vector<kh_board> boards;
    vector<int32_t> offsets, actions;
    vector<float> priors, values, planes;
    vector<int> owner;
    int partials = 0;

    try {
    while (status.code() == RUNNING) {
        boards.clear(); actions.clear(); offsets.assign(1, 0); owner.clear();
        for (int i = 0; i < ibatch; ++i) {
            Game& g = games[i];
            MCTS& tree = *g.tree;
            if (flush_old_trees && g.source_generation < model->get_generation()) {     // selfplay.cpp:115-127
                tree.reset();
                partials -= (int)g.trajectory.size();
                g.trajectory.clear(); g.moves.clear();
                g.source_generation = model->get_generation();
            }
            g.waiting = false;
            while (tree.n() < nodes && !(g.waiting = tree.select_leaf(&g.leaf))) {}      // selfplay.cpp:130
            if (tree.n() < nodes) {                                                      // an observation is ready
                boards.push_back(g.leaf.record);
                actions.insert(actions.end(), g.leaf.actions.begin(), g.leaf.actions.end());
                offsets.push_back((int32_t)actions.size());
                owner.push_back(i);
                continue;
            }
            // the tree has its visits: save the position, play the move           selfplay.cpp:135-160
            Step st;
            tree.get_env().record(&st.board);
            st.mcts.resize(PSIZE);
            tree.snapshot(st.mcts.data());
            st.pov = -tree.get_env().turn();
            ++partials;
            g.trajectory.push_back(std::move(st));
            float alpha = alpha_final;
            if (tree.get_env().ply() < alpha_cutoff) alpha = pow(alpha_decay, (float)tree.get_env().ply()) * alpha_initial;
            const int picked = tree.pick(alpha);
            g.moves.push_back(tree.get_env().debug_action(picked));
            tree.push(picked);
            float value;
            if (tree.get_env().terminal(&value)) {                                       // selfplay.cpp:163-189
                if (wants_pgn.exchange(false)) {
                    // movetext in coordinate notation (the reference prints SAN through the thc library)
                    string out;
                    for (size_t m = 0; m < g.moves.size(); ++m) {
                        if (m % 2 == 0) out += (m ? " " : "") + to_string(m / 2 + 1) + ".";
                        out += " " + g.moves[m];
                    }
                    string reason;
                    float v2;
                    tree.get_env().terminal_str(&v2, reason);
                    ret_pgn = out + " " + (value < 0 ? "0-1" : (value > 0 ? "1-0" : "1/2-1/2")) + " {" + reason + "}";
                }
                // the finished game's observations: one device encode for the whole trajectory
                const int n = (int)g.trajectory.size();
                vector<kh_board> tb((size_t)n);
                for (int k = 0; k < n; ++k) tb[k] = g.trajectory[k].board;
                planes.resize((size_t)n * OBSIZE);
                if (kh_encode(model->handle(), tb.data(), n, planes.data()) != KH_OK) throw runtime_error(kh_last_error());
                for (int k = 0; k < n; ++k)
                    replay_buffer.add(planes.data() + (size_t)k * OBSIZE, g.trajectory[k].mcts.data(),
                                      value == 0.0f ? draw_value : g.trajectory[k].pov * value);
                tree.reset();
                partials -= n;
                g.trajectory.clear(); g.moves.clear();
            }
            --i;                                                                         // selfplay.cpp:192: same slot again
        }
        const int nb = (int)boards.size();
        if (nb) {                                                                        // selfplay.cpp:196-200
            priors.resize(actions.size());
            values.resize((size_t)nb);
            const int rc = kh_encode_infer_legal(model->handle(), boards.data(), nb, offsets.data(), actions.data(), priors.data(), values.data());
            if (rc == KH_ERR_NAN_POLICY) throw runtime_error("inference policy output contains NaN");
            if (rc == KH_ERR_NAN_VALUE) throw runtime_error("inference value output contains NaN");
            if (rc != KH_OK) throw runtime_error(kh_last_error());
            for (int j = 0; j < nb; ++j) {
                Game& g = games[owner[j]];
                g.tree->expand_leaf(g.leaf, priors.data() + offsets[j], values[j]);
            }
        }
        auto pt = partial_trajectories.begin();                                          // selfplay.cpp:203-205
        advance(pt, id);
        *pt = partials;
    }
    } catch (exception& e) {
        cerr << "INFER " << id << ": " << e.what() << endl;
    }
    cout << "Terminating inference thread: " << id << endl;
}

void Selfplay::training_main(int id)                                         // selfplay.cpp:215-304
{
    cout << "TRAIN " << id << ": starting thread " << id << endl;
    const string modelpath = options::getStr("model_path", "/tmp/model.pt");
    long target_count = replay_buffer.size(), target_from = 0;
    const int target_incr = replay_buffer.size() * options::getInt("rpb_train_pct", 40) / 100;
    const int trajectories = replay_buffer.size() * options::getInt("training_sample_pct", 60) / 100;
    const bool detect_anomaly = options::getInt("training_detect_anomaly", 0);
    vector<float> inputs((size_t)trajectories * OBSIZE), mcts((size_t)trajectories * PSIZE), results((size_t)trajectories);

    while (status.code() == RUNNING) {
        if (replay_buffer.count() < target_count) {
            if (!id) {
                cout << "Gen " << model->get_generation() << " RPB " << 100 * (replay_buffer.count() - target_from) / (target_count - target_from)
                     << "% [" << replay_buffer.count() - target_from << " / " << target_count - target_from << "] | Partials: ";
                int inf = 0;
                for (auto& ct : partial_trajectories) cout << " Inf " << inf++ << ": " << ct;
                cout << endl;
            }
            this_thread::sleep_for(chrono::milliseconds(1000));
            continue;
        }
        cout << "TRAIN " << id << ": training generation " << model->get_generation() << " with " << trajectories
             << " trajectories sampled from last " << replay_buffer.size() << endl;
        NN cmodel(model);                                                    // selfplay.cpp:259
        replay_buffer.select_batch(inputs.data(), mcts.data(), results.data(), trajectories);
        cmodel.train(trajectories, inputs.data(), mcts.data(), results.data(), detect_anomaly);
        bool eval_result;
        try {
            eval_result = eval(model, &cmodel, id);
        } catch (exception& e) {
            cerr << "TRAIN " << id << ": evaluation failed: " << e.what() << endl;
            eval_result = false;
        }
        if (eval_result) {                                                   // selfplay.cpp:279-293
            cmodel.write(modelpath);
            model->read(modelpath);
            cout << "TRAIN " << id << ": candidate accepted: using new generation " << model->get_generation() << endl;
            if (options::getInt("flush_old_rpb", 1)) replay_buffer.clear();
            target_count = max((long)replay_buffer.size(), replay_buffer.count() + (long)target_incr);
            target_from = replay_buffer.count();
            continue;
        }
        cout << "TRAIN " << id << ": candidate rejected: generation remains " << model->get_generation() << endl;
        target_from = replay_buffer.count();
        target_count += target_incr;
    }
    cout << "TRAIN " << id << ": stopping thread" << endl;
}
