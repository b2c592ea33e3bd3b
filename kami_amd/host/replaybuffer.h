// replaybuffer.h — mirror of kami::ReplayBuffer (kami/replaybuffer.h:10-92): a fixed ring of
// (observation[obsize], mcts policy[psize], result) records under one mutex, uniform selection with
// replacement over the whole ring.  Same public methods; storage in vectors.
#pragma once

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace kami {

class ReplayBuffer {
    public:
        ReplayBuffer(int obsize, int psize, int bufsize)
            : obsize(obsize), psize(psize), bufsize(bufsize), input_buffer((size_t)obsize * bufsize),
              mcts_buffer((size_t)psize * bufsize), result_buffer((size_t)bufsize) {}

        void clear() { total = 0; write_index = 0; }                        // replaybuffer.h:31-34

        void add(const float* input, const float* mcts, float result)       // replaybuffer.h:36-56
        {
            std::lock_guard<std::mutex> lock(buffer_mut);
            std::memcpy(&input_buffer[(size_t)write_index * obsize], input, sizeof(float) * obsize);
            std::memcpy(&mcts_buffer[(size_t)write_index * psize], mcts, sizeof(float) * psize);
            result_buffer[write_index++] = result;
            write_index %= bufsize;
            ++total;
        }

        int size() { return bufsize; }
        long count() { return total; }

        void select_batch(float* dst_input, float* dst_mcts, float* dst_result, int n)   // replaybuffer.h:61-84
        {
            std::lock_guard<std::mutex> lock(buffer_mut);
            for (int i = 0; i < n; ++i) {
                const int source = rand() % bufsize;
                std::memcpy(dst_input + (size_t)i * obsize, &input_buffer[(size_t)source * obsize], sizeof(float) * obsize);
                std::memcpy(dst_mcts + (size_t)i * psize, &mcts_buffer[(size_t)source * psize], sizeof(float) * psize);
                dst_result[i] = result_buffer[source];
            }
        }

    private:
        int obsize, psize, bufsize;
        std::mutex buffer_mut;
        std::vector<float> input_buffer, mcts_buffer, result_buffer;
        int write_index = 0;
        long total = 0;
};

}  // namespace kami
