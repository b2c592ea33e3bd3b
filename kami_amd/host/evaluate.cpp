// evaluate.cpp — kami::eval (kami/evaluate.cpp:10-160) on this repository's host search: "evaluate_games"
// trees are played to the end, at most "evaluate_batch" leaves per model and round, the model whose
// turn it is at a tree's ROOT evaluates that tree's leaves (one shared tree per game, expanded without
// bootstrap), moves by the visit maximum; the candidate passes at "evaluate_target_pct" of the points,
// with the reference's early pass / fail.  Leaves go to the engines as compact records + legal
// actions (kh_encode_infer_legal) instead of planes and full policy rows.
#include "evaluate.h"
#include "env.h"
#include "mcts.h"
#include "options.h"

#include <iostream>
#include <memory>
#include <stdexcept>
#include <vector>

using namespace kami;

namespace {
void infer_batch(NN* model, std::vector<MCTS::Leaf*>& leaves, std::vector<MCTS*>& owner)
{
    if (leaves.empty()) return;
    std::vector<kh_board> boards;
    std::vector<int32_t> offsets(1, 0), actions;
    for (MCTS::Leaf* l : leaves) {
        boards.push_back(l->record);
        actions.insert(actions.end(), l->actions.begin(), l->actions.end());
        offsets.push_back((int32_t)actions.size());
    }
    std::vector<float> priors(actions.size()), values(leaves.size());
    const int rc = kh_encode_infer_legal(model->handle(), boards.data(), (int)boards.size(), offsets.data(), actions.data(),
                                         priors.data(), values.data());
    if (rc == KH_ERR_NAN_POLICY) throw std::runtime_error("inference policy output contains NaN");
    if (rc == KH_ERR_NAN_VALUE) throw std::runtime_error("inference value output contains NaN");
    if (rc != KH_OK) throw std::runtime_error(kh_last_error());
    for (size_t i = 0; i < leaves.size(); ++i) owner[i]->expand_leaf(*leaves[i], priors.data() + offsets[i], values[i]);
}
}  // namespace

bool kami::eval(NN* current_model, NN* candidate_model, int trainer)
{
    const int ebatch = options::getInt("evaluate_batch");                   // evaluate.cpp:12-15
    const int egames = options::getInt("evaluate_games");
    const int enodes = options::getInt("evaluate_nodes");
    const int etarget = options::getInt("evaluate_target_pct");
    if (ebatch < 1 || egames < 1 || enodes < 2) throw std::runtime_error("evaluate_batch / evaluate_games / evaluate_nodes not set");

    MCTSConfig cfg;
    cfg.cpuct = options::getFloat("cpuct", 1.0f);
    cfg.mcts_noise_weight = options::getFloat("mcts_noise_weight", 0.05f);
    cfg.seed = (unsigned)rand();
    std::vector<std::unique_ptr<MCTS>> trees;
    std::vector<float> candidate_turns((size_t)egames);
    std::vector<MCTS::Leaf> pending((size_t)egames);
    for (int i = 0; i < egames; ++i) {
        cfg.seed += 1;
        trees.emplace_back(new MCTS(cfg));
        candidate_turns[i] = (float)((rand() % 2) * 2 - 1);                  // evaluate.cpp:21-22
    }
    float score = 0.0f;
    int games = 0;
    std::cout << "EVAL " << trainer << ": evaluating model generation " << candidate_model->get_generation() << " over " << egames
              << " games" << std::endl;

    while (games < egames) {
        if (current_model->get_generation() >= candidate_model->get_generation()) {   // evaluate.cpp:53-59
            std::cout << "EVAL " << trainer << ": model was updated during evaluation, skipping!" << std::endl;
            return false;
        }
        std::vector<MCTS::Leaf*> cur_leaves, cd_leaves;
        std::vector<MCTS*> cur_owner, cd_owner;
        for (int i = 0; i < egames; ++i) {
            if ((int)cur_leaves.size() >= ebatch && (int)cd_leaves.size() >= ebatch) break;
            MCTS& t = *trees[i];
            const bool candidate_to_move = t.get_env().turn() == candidate_turns[i];
            bool have = false;
            while (t.n() < enodes && !(have = t.select_leaf(&pending[i]))) {}
            if (t.n() < enodes) {
                // a leaf waits for the model whose turn it is at this tree's root (evaluate.cpp:79-92); a full
                // batch leaves it for a later round
                auto& leaves = candidate_to_move ? cd_leaves : cur_leaves;
                auto& owner = candidate_to_move ? cd_owner : cur_owner;
                if ((int)leaves.size() < ebatch) { leaves.push_back(&pending[i]); owner.push_back(&t); }
                else if (have) t.release_leaf(pending[i]);
                continue;
            }
            t.push(t.pick());                                                // evaluate.cpp:95
            float tvalue;
            if (t.get_env().terminal(&tvalue)) {
                score += tvalue * candidate_turns[i] / 2.0f + 0.5f;          // evaluate.cpp:100
                games++;
                std::cout << "EVAL " << trainer << ": game " << games << " of " << egames << " [" << tvalue * candidate_turns[i]
                          << "]: score " << (int)(score * 100 / games) << "%" << std::endl;
                t.reset();
                candidate_turns[i] = candidate_to_move ? 1.0f : -1.0f;       // evaluate.cpp:106
                const float target_score = (float)((egames * etarget) / 100);
                if ((score + (egames - games)) < target_score) {
                    std::cout << "EVAL " << trainer << ": aborting evaluation, score is too low" << std::endl;
                    return false;
                }
                if (score >= target_score && games < egames) {
                    std::cout << "EVAL " << trainer << ": finished evaluating early: score >=" << (int)(score * 100 / games)
                              << "%, target " << etarget << std::endl;
                    return true;
                }
            }
            --i;                                                             // evaluate.cpp:128: same tree again
        }
        infer_batch(current_model, cur_leaves, cur_owner);
        infer_batch(candidate_model, cd_leaves, cd_owner);
    }
    std::cout << "EVAL " << trainer << ": finished evaluating: score " << (int)(score * 100 / games) << "%, target " << etarget
              << std::endl;
    return score * 100 / games >= etarget;
}
