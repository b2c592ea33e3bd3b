// mcts.h — host-side mirror of kami::Node / kami::MCTS (kami/mcts.h:14-349): same public methods,
// same arithmetic (checked against the reference's own search in tests/golden/mcts_ref_*.txt), with
// the two changes SURVEY §8f row 2 asks for:
//   * trees live on the heap in per-tree node pools (the reference puts `MCTS trees[ibatch]` and
//     per-call VLAs on the thread stack, selfplay.cpp:96, mcts.h:160, and news / deletes every node);
//     the children of a node are one contiguous block of the pool, re-rooting keeps the chosen
//     subtree by moving it into the tree's second pool.  A worker goes round hundreds of trees, so
//     every walk starts cache-cold: a node is 24 bytes (the reference's: 72 + a vector), with what
//     the reference keeps per node but can be had from the walk — the parent (the walk's path is
//     kept with the leaf), whose turn it is (alternates with depth) — left out;
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
#include <cstring>
#include <memory>
#include <random>
#include <stdexcept>
#include <vector>

namespace kami {

struct Node {                                   // mcts.h:14-64
    int n = 0;
    float w = 0.0f;
    float p = 0.0f;
    int16_t action = -1;
    int16_t nkids = 0;                          // -1: selected as a leaf, evaluation not back yet
    Node* kids = nullptr;                       // children: one contiguous block (the reference: vector<Node*>)

    struct Range { Node *b, *e; Node* begin() const { return b; } Node* end() const { return e; } size_t size() const { return (size_t)(e - b); }
                   bool empty() const { return b == e; } };
    Range children() const { return nkids > 0 ? Range{ kids, kids + nkids } : Range{ nullptr, nullptr }; }

    float q(float def = 1.0f) const { return n > 0 ? w / n : def; }
};
static_assert(sizeof(Node) == 24, "Node is meant to stay small");

// Node storage of one tree: fixed-size chunks of raw memory (whoever takes nodes initialises them), a
// node's children always inside one chunk; clear() keeps the chunks for the next tree.
class NodePool {
    static constexpr size_t CHUNK = 2048;
    struct Raw { void operator()(Node* p) const { ::operator delete(static_cast<void*>(p)); } };
    std::vector<std::unique_ptr<Node, Raw>> chunks;
    size_t cur = 0, used = CHUNK;              // chunks[cur - 1] is the one being filled
public:
    Node* alloc(size_t count)
    {
        if (count > CHUNK) throw std::runtime_error("node block too large");
        if (used + count > CHUNK) {
            if (cur == chunks.size()) chunks.emplace_back(static_cast<Node*>(::operator new(CHUNK * sizeof(Node))));
            ++cur;
            used = 0;
        }
        Node* r = chunks[cur - 1].get() + used;
        used += count;
        return r;
    }
    void clear() { cur = 0; used = CHUNK; }
    const Node* next() const { return cur ? chunks[cur - 1].get() + (used < CHUNK ? used : 0) : nullptr; }   // where alloc() will (mostly) hand out next
};

// -ln(u) for u in (0, 1): exponent + degree-5 fit of log2 on [1, 2) (|error| < 4e-5 in log2) — for the
// expansion noise, where only the distribution matters
inline float neg_log(float u)
{
    uint32_t bits;
    std::memcpy(&bits, &u, 4);
    const float e = (float)((int)(bits >> 23) - 127);
    bits = (bits & 0x007fffffu) | 0x3f800000u;
    float m;
    std::memcpy(&m, &bits, 4);
    const float l2 = ((((0.043428365f * m - 0.40486231f) * m + 1.5938846f) * m - 3.4924660f) * m + 5.0468531f) * m - 2.7868056f;
    return -0.69314718f * (e + l2);
}

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
        std::vector<Node*> path;                // root ... node (the reference follows parent pointers instead)
        float turn = 0.0f;                      // Node::turn of the leaf (mcts.h:21)
        std::vector<int> actions;               // Env::actions() at the leaf
        kh_board record;                        // Env::record() at the leaf
    };

private:
    Env env;
    std::unique_ptr<NodePool> pool, spare;      // push() copies the kept subtree from one into the other
    Node* target = nullptr;                     // reference-shaped single-leaf API
    Leaf single;
    double cPUCT;
    bool force_expand_unvisited;
    float unvisited_node_value;
    float noise_weight;
    int scale_cpuct_by_actions;
    std::mt19937 rng;                           // move choice (pick)
    uint64_t noise_state;                       // expansion noise: ~30 draws per leaf, splitmix64 is enough

    uint64_t next_noise()
    {
        uint64_t z = (noise_state += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }

    int inflight = 0;                           // virtual visits currently counted in the nodes' n
    std::vector<Node*> walk;                    // path of the walk in progress

    void copy_subtree(const Node* src, Node* d, NodePool& into)
    {
        d->n = src->n; d->w = src->w; d->p = src->p; d->action = src->action;
        d->nkids = src->nkids;
        d->kids = src->nkids > 0 ? into.alloc((size_t)src->nkids) : nullptr;
        for (int i = 0; i < src->nkids; ++i) copy_subtree(src->kids + i, d->kids + i, into);
    }

    // Node::turn (mcts.h:21,82,292): the root's is minus the side to move there, it flips with every level
    float turn_at(size_t depth) const { return (depth & 1) ? env_root_turn() : -env_root_turn(); }
    float env_root_turn() const { return root_turn_env; }

    // Node::backprop (mcts.h:34-41) along a stored path, leaf first
    void backprop(const std::vector<Node*>& path, float value)
    {
        for (size_t d = path.size(); d-- > 0;) {
            Node* x = path[d];
            x->n += 1;
            x->w += 0.5f + (value * turn_at(d)) / 2.0f;
        }
    }

    void mark(const std::vector<Node*>& path, int delta)       // virtual visit on the path leaf -> root
    {
        for (Node* x : path) x->n += delta;
        inflight += delta;
    }

    float root_turn_env = 1.0f;                 // env.turn() with the environment at the root

public:
    Node* root = nullptr;

    explicit MCTS(const MCTSConfig& c = MCTSConfig())
        : pool(new NodePool()), spare(new NodePool()), cPUCT(c.cpuct), force_expand_unvisited(c.force_expand_unvisited != 0),
          unvisited_node_value((float)c.unvisited_node_value_pct / 100.0f), noise_weight(c.mcts_noise_weight),
          scale_cpuct_by_actions(c.scale_cpuct_by_actions), rng(c.seed), noise_state(0x853c49e6748fea9bull ^ ((uint64_t)c.seed << 17))
    {
        root = pool->alloc(1);
        *root = Node();
        root_turn_env = env.turn();             // mcts.h:82
    }

    int n() const { return root->n - inflight; }         // mcts.h:112 (virtual visits do not count)

    void push(int action)                       // mcts.h:114-137: re-root, keep the chosen subtree
    {
        if (inflight) throw std::runtime_error("push with leaves in flight");
        const Node* next = nullptr;
        for (const Node& c : root->children())
            if (c.action == action) next = &c;
        if (!next) throw std::runtime_error("no child for action");
        spare->clear();
        Node* r = spare->alloc(1);
        copy_subtree(next, r, *spare);
        pool.swap(spare);
        root = r;
        target = nullptr;
        env.push(action);
        root_turn_env = env.turn();
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
        walk.clear();
        walk.push_back(t);
        auto unwind = [&]() { for (size_t d = walk.size(); d > 1; --d) env.pop(); };
        for (;;) {
            if (t->nkids <= 0) {
                if (t->nkids < 0) { unwind(); if (blocked) *blocked = true; return false; }
                float value;
                if (env.terminal(&value)) {
                    // a terminal leaf is scored with the tree as the other in-flight walks left it
                    backprop(walk, value);
                    unwind();
                    return false;
                }
                out->node = t;
                out->path = walk;
                out->turn = turn_at(walk.size() - 1);
                out->actions = env.actions();
                env.record(&out->record);
                t->nkids = -1;
                mark(walk, +1);
                unwind();
                return true;
            }
            double best_uct = -1000.0;
            Node* best_child = nullptr;
            float cpuct = cPUCT;
            if (scale_cpuct_by_actions) cpuct /= (float)t->nkids;
            const double sqrt_n = sqrt(t->n);          // the reference evaluates sqrt(target->n) per child: same value
            const float child_default = unvisited_node_value * turn_at(walk.size());
            for (Node& cn : t->children()) {
                Node* c = &cn;
                if (force_expand_unvisited && !c->n) { best_child = c; break; }
                double uct = c->q(child_default) + c->p * cpuct * sqrt_n / (double)(c->n + 1);
                if (uct > best_uct) { best_child = c; best_uct = uct; }
            }
            if (!best_child) { unwind(); throw std::runtime_error("no best child to select, but children present!"); }
            env.push(best_child->action);
            walk.push_back(best_child);
            t = best_child;
        }
    }

    // priors[i] belongs to leaf.actions[i] and is already policy[a_i] / sum_j policy[a_j] (mcts.h:273-276 on
    // the device: kh_encode_infer_legal); the noise mixing and the value convention are mcts.h:279-316.
    void expand_leaf(Leaf& leaf, const float* priors, float value)
    {
        Node* t = leaf.node;
        mark(leaf.path, -1);
        const size_t na = leaf.actions.size();
        // mcts.h:279-288 draws gamma(1, 1) noise for every action of every expansion; gamma(1, 1) IS the
        // unit exponential, so -log(u) draws the same distribution at a fraction of the cost (and
        // nothing is drawn when the weight is zero)
        float noise[chess::MAX_MOVES + 1];
        float total_noise = 1.0f;
        if (noise_weight != 0.0f) {
            for (size_t i = 0; i < na; i += 2) {                                                            // (0, 1)
                const uint64_t r = next_noise();
                noise[i] = (float)((uint32_t)(r >> 40) + 1) * (1.0f / 16777217.0f);
                noise[i + 1] = (float)((uint32_t)(r >> 8 & 0xffffff) + 1) * (1.0f / 16777217.0f);
            }
            total_noise = 0.0f;
            for (size_t i = 0; i < na; ++i) { noise[i] = neg_log(noise[i]); total_noise += noise[i]; }
        } else {
            for (size_t i = 0; i < na; ++i) noise[i] = 0.0f;
        }
        t->kids = na ? pool->alloc(na) : nullptr;
        t->nkids = (int16_t)na;
        for (size_t i = 0; i < na; ++i) {
            Node* c = t->kids + i;
            c->n = 0; c->w = 0.0f;
            c->p = (1 - noise_weight) * priors[i] + noise_weight * (noise[i] / total_noise);
            c->action = (int16_t)leaf.actions[i];
            c->nkids = 0;
            c->kids = nullptr;
        }
        value *= leaf.turn;                     // mcts.h:304-310
        backprop(leaf.path, value);
    }

    // give a selected leaf back unevaluated (its batch was full): only the virtual visit is undone
    void release_leaf(Leaf& leaf)
    {
        mark(leaf.path, -1);
        leaf.node->nkids = 0;
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

    // ... and this before expand_leaf(): the path that will be updated and the memory the children go to
    void prefetch_expand(const Leaf& leaf) const
    {
        for (const Node* x : leaf.path) __builtin_prefetch(x, 1);
        const char* k = reinterpret_cast<const char*>(pool->next());
        if (k) for (size_t off = 0, end = leaf.actions.size() * sizeof(Node); off < end; off += 64) __builtin_prefetch(k + off, 1);
    }

    // a worker that goes round many trees calls this for the tree it will visit next: the first level of
    // the walk (the root's children) and the position are cold by the time their turn comes
    void prefetch() const
    {
        const char* k = reinterpret_cast<const char*>(root->kids);
        for (int off = 0, end = (root->nkids > 0 ? root->nkids : 0) * (int)sizeof(Node); off < end; off += 64) __builtin_prefetch(k + off);
        env.prefetch();
    }

    void reset()                                // mcts.h:331-339
    {
        env = Env();
        target = nullptr;
        pool->clear();
        root = pool->alloc(1);
        *root = Node();
        root_turn_env = env.turn();
        inflight = 0;
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
