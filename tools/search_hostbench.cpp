// search_hostbench.cpp — host-only cost of the search loop (select_leaf / expand_leaf / push) with a
// stand-in evaluator (uniform priors, hashed value): what one worker thread can feed the engine.
//   g++ -O2 -std=c++17 -I include -I kami_amd/host tools/search_hostbench.cpp -o /tmp/hostbench
#include "mcts.h"
#include <chrono>
#include <cstdio>
using namespace kami;
int main(int argc, char** argv)
{
    const int games = argc > 1 ? atoi(argv[1]) : 256, nodes = argc > 2 ? atoi(argv[2]) : 64;
    const long target = argc > 3 ? atol(argv[3]) : 400000;
    std::vector<std::unique_ptr<MCTS>> trees;
    for (int i = 0; i < games; ++i) { MCTSConfig c; c.seed = 1234u + i; trees.emplace_back(new MCTS(c)); }
    std::vector<MCTS::Leaf> leaves((size_t)games);
    std::vector<char> have((size_t)games);
    float pri[chess::MAX_MOVES];
    long evals = 0, moves = 0, finished = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (evals < target) {
        for (int g = 0; g < games; ++g) {
            MCTS& t = *trees[g];
            if (g + 2 < games) trees[g + 2]->prefetch();
            have[g] = 0;
            for (;;) {
                if (t.n() >= nodes) {
                    t.push(t.pick(1.0f)); ++moves;
                    float v;
                    if (t.get_env().terminal(&v)) { t.reset(); ++finished; }
                    continue;
                }
                if (t.select_leaf(&leaves[g])) { have[g] = 1; break; }
            }
        }
        for (int g = 0; g < games; ++g) {
            if (!have[g]) continue;
            if (g + 2 < games && have[g + 2]) trees[g + 2]->prefetch_expand(leaves[g + 2]);
            const size_t na = leaves[g].actions.size();
            // peaked pseudo-priors (a network's are far from uniform; uniform ones make every PUCT choice a tie)
            uint64_t hsh = leaves[g].record.piece_occ[0] * 0x9e3779b97f4a7c15ull ^ leaves[g].record.color_occ[0];
            float tot = 0.0f;
            for (size_t i = 0; i < na; ++i) {
                hsh = hsh * 6364136223846793005ull + 1442695040888963407ull;
                const float u = (float)(hsh >> 40) * (1.0f / 16777216.0f);
                pri[i] = u * u * u * u + 0.01f;
                tot += pri[i];
            }
            for (size_t i = 0; i < na; ++i) pri[i] /= tot;
            const float v = (float)((leaves[g].record.piece_occ[0] * 0x9e3779b97f4a7c15ull >> 40) % 2001) / 1000.0f - 1.0f;
            trees[g]->expand_leaf(leaves[g], pri, v * 0.2f);
            ++evals;
        }
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("games %d nodes %d: %ld evals in %.3f s = %.0f leaf-evals/s (%.2f us/leaf), moves %ld, finished %ld\n", games, nodes, evals, s,
           evals / s, 1e6 * s / evals, moves, finished);
    return 0;
}
