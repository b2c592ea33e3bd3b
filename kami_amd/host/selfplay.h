// selfplay.h — mirror of kami::Selfplay (kami/selfplay.h:23-101): same public interface, so kami.cpp
// (kami.cpp:53-54,124,147,189) compiles against it unchanged.  Inside, the inference threads run this
// repository's search (mcts.h: heap trees, compact observations, legal-move priors from
// kh_encode_infer_legal) instead of the reference's, and the trainer thread is selfplay.cpp:215-304 on
// kami::NN::train = kh_train.
#pragma once

#include "nn/nn.h"
#include "replaybuffer.h"

#include <atomic>
#include <chrono>
#include <cstdint>
#include <list>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace kami {

class Selfplay {
    public:
        Selfplay(NN* model);

        void start();               // selfplay.cpp:21-35
        void stop();                // selfplay.cpp:37-56

        enum StatusCode { STOPPED, RUNNING, WAITING };

        struct Status {             // selfplay.h:46-68
            StatusCode _code = STOPPED;
            std::mutex _lock;
            std::string _message;

            std::string message(std::string text = "")
            {
                std::lock_guard<std::mutex> lock(_lock);
                if (!text.size()) return text;
                return _message = text;
            }
            StatusCode code(int newcode = -1)
            {
                std::lock_guard<std::mutex> lock(_lock);
                if (newcode < 0) return _code;
                return _code = StatusCode(newcode);
            }
        };

        Status status;
        ReplayBuffer& get_rbuf() { return replay_buffer; }

        std::string get_next_pgn()  // selfplay.h:73-80
        {
            wants_pgn = true;
            while (wants_pgn) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            return ret_pgn;
        }

    private:
        std::vector<std::thread> inference;
        std::vector<std::thread> training;
        NN* model;
        ReplayBuffer replay_buffer;
        int ibatch;
        int nodes;
        std::atomic<bool> wants_pgn;
        std::string ret_pgn;
        std::list<std::atomic<int>> partial_trajectories;
        std::mutex partial_trajectories_lock;

        void inference_main(int id);
        void training_main(int id);
};

}  // namespace kami
