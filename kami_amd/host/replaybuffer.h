// replaybuffer.h — kami::ReplayBuffer with the reference's public methods (kami/replaybuffer.h:10-92):
// a fixed ring of (observation[obsize], mcts policy[psize], result) records under one mutex, uniform
// selection with replacement over the whole ring (written or not).  Records are stored interleaved,
// one contiguous row per record.
#ifndef KAMI_AMD_HOST_REPLAYBUFFER_H
#define KAMI_AMD_HOST_REPLAYBUFFER_H

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <vector>

namespace kami {

class ReplayBuffer {
    const int obs_len, pol_len, capacity;
    const size_t row;                       // floats per record: observation, policy, result
    std::vector<float> ring;
    std::mutex lock;
    int head = 0;                           // next slot to write
    long added = 0;                         // records ever added (the reference's `total`)

public:
    ReplayBuffer(int obsize, int psize, int bufsize)
        : obs_len(obsize), pol_len(psize), capacity(bufsize), row((size_t)obsize + psize + 1), ring(row * (size_t)bufsize, 0.0f) {}

    int size() { return capacity; }
    long count() { return added; }

    void clear()
    {
        head = 0;
        added = 0;
    }

    void add(const float* input, const float* mcts, float result)
    {
        std::lock_guard<std::mutex> hold(lock);
        float* slot = ring.data() + row * (size_t)head;
        std::copy(input, input + obs_len, slot);
        std::copy(mcts, mcts + pol_len, slot + obs_len);
        slot[obs_len + pol_len] = result;
        head = (head + 1) % capacity;
        ++added;
    }

    // n draws with replacement over ALL slots (replaybuffer.h:61-84: duplicates and never-written slots included)
    void select_batch(float* dst_input, float* dst_mcts, float* dst_result, int n)
    {
        std::lock_guard<std::mutex> hold(lock);
        for (int k = 0; k < n; ++k) {
            const float* slot = ring.data() + row * (size_t)(rand() % capacity);
            std::copy(slot, slot + obs_len, dst_input + (size_t)k * obs_len);
            std::copy(slot + obs_len, slot + obs_len + pol_len, dst_mcts + (size_t)k * pol_len);
            dst_result[k] = slot[obs_len + pol_len];
        }
    }
};

}  // namespace kami
#endif
