// mcts.h — host-side mirror of kami::Node / kami::MCTS (kami/mcts.h:14-349): same public methods,
// same arithmetic (checked against the reference's own search in tests/golden/mcts_ref_*.txt), with
// the two changes SURVEY §8f row 2 asks for:
//   * trees live on the heap in per-tree node pools (the reference puts `MCTS trees[ibatch]` and
//     per-call VLAs on the thread stack, selfplay.cpp:96, mcts.h:160, and news / deletes every node);
//     the children of a node are one contiguous block of the pool, re-rooting keeps the chosen
//     subtree by moving it into a fresh pool;
//   * several leaves of one tree can be in flight at once: select_leaf() marks the path with a
//     virtual visit (n + 1, no reward) so that the next selection goes elsewhere, expand_leaf() takes
//     the mark back before the real backprop.  With one leaf in flight the arithmetic is exactly the
//     reference's.
// The observation handed to the evaluator is the compact kh_board record and the legal actions, not
// planes and not the full 4672-entry policy (see env.h); expand() with a full policy row is kept for
// parity tests.
#pragma once
#include "env.h"

#include <cmath>
#include <memory>
#include <random>
#include <stdexcept>
#include <vector>

namespace kami {

struct Node {                                   // mcts.h:14-64
    int n = 0;
    float w = 0.0f;
    float p = 0.0f;
    int action = -1;
    Node* kids = nullptr;                       // children: one contiguous block (the reference: vector<Node*>)
    int nkids = 0;
    Node* parent = nullptr;
    float turn = 0.0f;
    int inflight = 0;                           // virtual visits currently counted in n (this node and below)
    bool pending = false;                       // selected as a leaf, evaluation not back yet

    struct Range { Node *b, *e; Node* begin() const { return b; } Node* end() const { return e; } size_t size() const { return (size_t)(e - b); }
                   bool empty() const { return b == e; } };
    Range children() const { return Range{ kids, kids + nkids }; }

    float q(float def = 1.0f) const { return n > 0 ? w / n : def; }

    void backprop(float value)                  // mcts.h:34-41
    {
        n += 1;
        w += 0.5f + (value * turn) / 2.0f;
        if (parent) parent->backprop(value);
    }
};

// Node storage of one tree: fixed-size chunks, a node's children always inside one chunk.
class NodePool {
    static constexpr size_t CHUNK = 2048;
    std::vector<std::unique_ptr<Node[]>> chunks;
    size_t used = CHUNK;
public:
    Node* alloc(size_t count)
    {
        if (count > CHUNK) throw std::runtime_error("node block too large");
        if (used + count > CHUNK) { chunks.emplace_back(new Node[CHUNK]); used = 0; }
        Node* r = chunks.back().get() + used;
        used += count;
        return r;
    }
};

struct MCTSConfig {                             // option keys and defaults of mcts.h:83-95
    float cpuct = 1.0f;
    int force_expand_unvisited = 0;
    int unvisited_node_value_pct = 100;
    int scale_cpuct_by_actions = 0;
    float mcts_noise_alpha = 0.05f;             // read but unused by the reference as well
    float mcts_noise_weight = 0.05f;
    unsigned seed = 0;                          // the reference seeds with time(NULL)
};

class MCTS {
public:
    struct Leaf {                               // a position waiting for the network
        Node* node = nullptr;
        std::vector<int> actions;               // Env::actions() at the leaf
        kh_board record;                        // Env::record() at the leaf
    };

private:
    Env env;
    std::unique_ptr<NodePool> pool;
    Node* target = nullptr;                     // reference-shaped single-leaf API
    Leaf single;
    double cPUCT;
    bool force_expand_unvisited;
    float unvisited_node_value;
    float noise_weight;
    int scale_cpuct_by_actions;
    std::mt19937 rng;

    void copy_subtree(const Node* src, Node* d, Node* parent, NodePool& into)
    {
        d->n = src->n; d->w = src->w; d->p = src->p; d->action = src->action; d->turn = src->turn;
        d->parent = parent;
        d->nkids = src->nkids;
        d->kids = src->nkids ? into.alloc((size_t)src->nkids) : nullptr;
        for (int i = 0; i < src->nkids; ++i) copy_subtree(src->kids + i, d->kids + i, d, into);
    }

    void mark(Node* leaf, int delta)            // virtual visit on the path leaf -> root
    {
        for (Node* x = leaf; x; x = x->parent) { x->n += delta; x->inflight += delta; }
    }

public:
    Node* root = nullptr;

    explicit MCTS(const MCTSConfig& c = MCTSConfig())
        : pool(new NodePool()), cPUCT(c.cpuct), force_expand_unvisited(c.force_expand_unvisited != 0),
          unvisited_node_value((float)c.unvisited_node_value_pct / 100.0f), noise_weight(c.mcts_noise_weight),
          scale_cpuct_by_actions(c.scale_cpuct_by_actions), rng(c.seed)
    {
        root = pool->alloc(1);
        root->turn = -env.turn();               // mcts.h:82
    }

    int n() const { return root->n - root->inflight; }   // mcts.h:112 (virtual visits do not count)

    void push(int action)                       // mcts.h:114-137: re-root, keep the chosen subtree
    {
        if (root->inflight) throw std::runtime_error("push with leaves in flight");
        const Node* next = nullptr;
        for (const Node& c : root->children())
            if (c.action == action) next = &c;
        if (!next) throw std::runtime_error("no child for action");
        std::unique_ptr<NodePool> fresh(new NodePool());
        Node* r = fresh->alloc(1);
        copy_subtree(next, r, nullptr, *fresh);
        pool = std::move(fresh);
        root = r;
        target = nullptr;
        env.push(action);
    }

    int pick(float alpha = 0.0f)                // mcts.h:139-181
    {
        if (root->children().empty()) throw std::runtime_error("no children to pick from");
        if (alpha < 0.1f) {
            int best_n = 0, best_action = -1;
            for (const Node& c : root->children())
                if (c.n > best_n) { best_n = c.n; best_action = c.action; }
            return best_action;
        }
        double dist[chess::MAX_MOVES];
        double length = 0.0;
        const int nk = root->nkids;
        for (int i = 0; i < nk; ++i) { dist[i] = std::pow(root->kids[i].n, 1.0f / alpha); length += dist[i]; }
        double ind = std::generate_canonical<double, 53>(rng);      // (the reference draws rand() / RAND_MAX)
        for (int i = 0; i < nk; ++i) {
            ind -= dist[i] / length;
            if (ind <= 0.0) return root->kids[i].action;
        }
        return root->kids[nk - 1].action;
    }

    // Walk from the root to a leaf by the reference's rule (mcts.h:183-259).  Returns true and fills `out`
    // when a position needs the network; returns false when the walk ended in a terminal position
    // (backpropagated here, like the reference) or ran into a leaf that is already waiting
    // (`*blocked` set: nothing more to select in this tree until results come back).
    bool select_leaf(Leaf* out, bool* blocked = nullptr)
    {
        if (blocked) *blocked = false;
        Node* t = root;
        int depth = 0;
        auto unwind = [&]() { for (; depth > 0; --depth) env.pop(); };
        for (;;) {
            if (t->nkids == 0) {
                if (t->pending) { unwind(); if (blocked) *blocked = true; return false; }
                float value;
                if (env.terminal(&value)) {
                    // a terminal leaf is scored with the tree as the other in-flight walks left it
                    t->backprop(value);
                    unwind();
                    return false;
                }
                out->node = t;
                out->actions = env.actions();
                env.record(&out->record);
                t->pending = true;
                mark(t, +1);
                unwind();
                return true;
            }
            double best_uct = -1000.0;
            Node* best_child = nullptr;
            float cpuct = cPUCT;
            if (scale_cpuct_by_actions) cpuct /= (float)t->nkids;
            bool forced = false;
            const double sqrt_n = sqrt(t->n);          // the reference evaluates sqrt(target->n) per child: same value
            for (Node& cn : t->children()) {
                Node* c = &cn;
                if (force_expand_unvisited && !c->n) { best_child = c; forced = true; break; }
                double uct = c->q(unvisited_node_value * c->turn) + c->p * cpuct * sqrt_n / (double)(c->n + 1);
                if (uct > best_uct) { best_child = c; best_uct = uct; }
            }
            (void)forced;
            if (!best_child) { unwind(); throw std::runtime_error("no best child to select, but children present!"); }
            env.push(best_child->action);
            ++depth;
            t = best_child;
        }
    }

    // priors[i] belongs to leaf.actions[i] and is already policy[a_i] / sum_j policy[a_j] (mcts.h:273-276 on
    // the device: kh_encode_infer_legal); the noise mixing and the value convention are mcts.h:279-316.
    void expand_leaf(Leaf& leaf, const float* priors, float value)
    {
        Node* t = leaf.node;
        mark(t, -1);
        t->pending = false;
        const size_t na = leaf.actions.size();
        // mcts.h:279-288 draws gamma(1, 1) noise for every action of every expansion; gamma(1, 1) IS the
        // unit exponential, so -log(u) draws the same distribution at a fraction of the cost (and
        // nothing is drawn when the weight is zero)
        float noise[chess::MAX_MOVES];
        float total_noise = 1.0f;
        if (noise_weight != 0.0f) {
            total_noise = 0.0f;
            for (size_t i = 0; i < na; ++i) {
                const float u = (float)((rng() >> 8) + 1) * (1.0f / 16777217.0f);      // (0, 1)
                noise[i] = -std::log(u);
                total_noise += noise[i];
            }
        } else {
            for (size_t i = 0; i < na; ++i) noise[i] = 0.0f;
        }
        t->kids = na ? pool->alloc(na) : nullptr;
        t->nkids = (int)na;
        for (size_t i = 0; i < na; ++i) {
            Node* c = t->kids + i;
            *c = Node();
            c->action = leaf.actions[i];
            c->parent = t;
            c->turn = -t->turn;
            c->p = (1 - noise_weight) * priors[i] + noise_weight * (noise[i] / total_noise);
        }
        value *= t->turn;                       // mcts.h:304-310
        t->backprop(value);
    }

    // give a selected leaf back unevaluated (its batch was full): only the virtual visit is undone
    void release_leaf(Leaf& leaf)
    {
        mark(leaf.node, -1);
        leaf.node->pending = false;
    }

    // ---- the reference's one-leaf-at-a-time interface ------------------------------------------
    bool select(kh_board* obs)                  // mcts.h:183-259 (obs: compact record instead of planes)
    {
        if (target) throw std::runtime_error("select() before the previous leaf was expanded");
        if (!select_leaf(&single)) return false;
        target = single.node;
        *obs = single.record;
        return true;
    }

    void expand(const float* policy, float value)    // mcts.h:261-327, full policy row
    {
        if (!target) throw std::runtime_error("expand() without a selected leaf");
        float ptotal = 0.0f;
        for (int a : single.actions) ptotal += policy[a];
        std::vector<float> pr(single.actions.size());
        for (size_t i = 0; i < pr.size(); ++i) pr[i] = policy[single.actions[i]] / ptotal;
        expand_leaf(single, pr.data(), value);
        target = nullptr;
    }

    void expand_priors(const float* priors, float value)
    {
        if (!target) throw std::runtime_error("expand() without a selected leaf");
        expand_leaf(single, priors, value);
        target = nullptr;
    }

    Env& get_env() { return env; }              // mcts.h:329

    void reset()                                // mcts.h:331-339
    {
        env = Env();
        target = nullptr;
        pool.reset(new NodePool());
        root = pool->alloc(1);
        *root = Node();
        root->turn = -env.turn();
    }

    // visit distribution of the root as (action, n / (root n - 1)) pairs: mcts.h:341-348 without the 4672-wide row
    void snapshot_sparse(std::vector<int>& actions, std::vector<float>& visits) const
    {
        actions.clear(); visits.clear();
        for (const Node& c : root->children()) { actions.push_back(c.action); visits.push_back((float)c.n / (float)(root->n - 1)); }
    }
    void snapshot(float* pspace) const          // mcts.h:341-348
    {
        for (int i = 0; i < PSIZE; ++i) pspace[i] = 0.0f;
        for (const Node& c : root->children()) pspace[c.action] = (float)c.n / (float)(root->n - 1);
    }
};

}  // namespace kami
