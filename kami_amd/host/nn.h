// kami_amd/host/nn.h — C++ host mirror of the reference's evaluator class over the C ABI.
//
// Drop this file (with nn.cpp) in place of kami/nn/nn.{h,cpp} and link libkamihip.so instead of
// libtorch: kami.cpp, selfplay.cpp, evaluate.cpp and the programs under test/ compile unchanged
// (recipe: kami_amd/host/Makefile, target dropin).  Same public method set, argument meaning
// and error behaviour as class kami::NN (reference kami/nn/nn.h:40-73):
//   - infer() throws std::runtime_error("inference policy output contains NaN") /
//     ("inference value output contains NaN") exactly where the reference does (nn.cpp:176-180);
//   - infer() may be called concurrently from many threads; read() swaps weights atomically
//     (the reference serialises with a shared_mutex, nn.cpp:166,206);
//   - "filters" / "residuals" come from kami::options like the reference's module (nn.cpp:42-43).
// Differences, on purpose: the engine only exists on the GPU (isCUDA() is always true, there is
// no force_cpu path: construction throws without an MI355X); train() runs the reference's SGD loop on
// the device in fp32 (kh_train); read() takes the reference's own libtorch checkpoints (parsed without
// libtorch, kh_checkpoint_read) and the engine's KAMW blobs, write() produces KAMW (kami_amd/weights.py);
// the GPU is option "engine_device" / LOCAL_RANK instead of the reference's fixed kCUDA:0 (nn.cpp:121).
#pragma once

#include <string>

struct kh_engine;

// kami.cpp:31-32 names two libtorch calls; they only tune libtorch's own CPU thread pools, which
// do not exist here.  This is an API-surface stub for those two names, not a runtime shim.
namespace torch {
inline void set_num_threads(int) {}
inline void set_num_interop_threads(int) {}
struct Device {
    bool cuda = true;
    int index = 0;
    bool is_cuda() const { return cuda; }
};
}  // namespace torch

namespace kami {

class NN {
    private:
        kh_engine* eng = nullptr;
        int width, height, features, psize;
        int filters, residuals;
        torch::Device device;
        int dtype = 0;          // KH_F32 / KH_BF16 / KH_F16 of this engine

        void create(int dtype);
        void load_blob(const float* blob, size_t n, int generation);

    public:
        NN(int width, int height, int features, int psize, bool force_cpu=false);
        NN(NN* other);
        ~NN();
        NN(const NN&) = delete;
        NN& operator=(const NN&) = delete;

        int get_generation();

        torch::Device get_device() { return device; }
        bool isCUDA() { return true; }
        int obsize() const { return width * height * features; }
        int polsize() const { return psize; }

        void infer(float* input, int batch, float* policy, float* value);
        void train(int trajectories, float* inputs, float* obs_p, float* obs_v, bool detect_anomaly=false);

        void read(std::string path);
        void write(std::string path);

        NN* clone();

        // engine extras (not in the reference): compact ingest, raw handle, one evaluator per GPU
        kh_engine* handle() { return eng; }
        NN(NN* other, int device_index);    // a replica of `other` on another GPU of the node (same weights, same generation)
        void sync_from(NN* other);          // install other's current weights and generation (the weight publish after training)
};

}  // namespace kami
