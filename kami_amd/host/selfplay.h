// selfplay.h — kami::Selfplay with the reference's public interface (kami/selfplay.h:23-101: constructor
// from the model, start / stop, the status object, get_rbuf, get_next_pgn), so that kami.cpp
// (kami.cpp:53-54,124,147,189) compiles against it unchanged.  Everything else lives behind one pointer
// (selfplay.cpp): the inference threads run this repository's search (mcts.h: heap trees, compact
// observations, legal-move priors from kh_encode_infer_legal), the trainer thread is
// selfplay.cpp:215-304 on kami::NN::train = kh_train.
#ifndef KAMI_AMD_HOST_SELFPLAY_H
#define KAMI_AMD_HOST_SELFPLAY_H

#include "nn/nn.h"
#include "replaybuffer.h"

#include <atomic>
#include <memory>
#include <mutex>
#include <string>

namespace kami {

class Selfplay {
public:
    enum StatusCode { STOPPED, RUNNING, WAITING };

    // Same two accessors as the reference's Status (selfplay.h:46-68): called without an argument they read,
    // with one they set.  The code is an atomic here, only the message needs the lock.
    class Status {
        std::atomic<int> state{ STOPPED };
        std::mutex guard;
        std::string text;
    public:
        StatusCode code(int newcode = -1)
        {
            if (newcode >= 0) state.store(newcode);
            return StatusCode(state.load());
        }
        std::string message(std::string update = "")
        {
            if (update.empty()) return update;          // (the reference returns the empty argument, not the stored text)
            std::lock_guard<std::mutex> hold(guard);
            text = update;
            return text;
        }
    };

    explicit Selfplay(NN* model);
    ~Selfplay();

    void start();
    void stop();
    ReplayBuffer& get_rbuf();
    std::string get_next_pgn();          // blocks until an inference thread finishes a game

    Status status;

private:
    struct Impl;
    std::unique_ptr<Impl> impl;
};

}  // namespace kami
#endif
