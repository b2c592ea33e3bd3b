// kami_amd/host/nn.cpp — kami::NN implemented on libkamihip.so (see nn.h).
#include "nn.h"
#include "../options.h"          // the reference's kami/options.h when dropped into kami/nn/

#include "kami_hip.h"

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <random>
#include <stdexcept>
#include <vector>

using namespace kami;

namespace {

[[noreturn]] void raise(int rc)
{
    // the reference's two NaN messages come through verbatim from the C ABI (nn.cpp:177,180)
    (void)rc;
    throw std::runtime_error(kh_last_error());
}

int dtype_from_options()
{
    const std::string d = options::getStr("engine_dtype", "bf16");
    if (d == "f32" || d == "fp32") return KH_F32;
    if (d == "f16" || d == "fp16") return KH_F16;
    return KH_BF16;
}

// The reference pins kCUDA:0 (nn.cpp:121).  Here one kami process drives one GPU of the node: option key
// "engine_device", else the launcher's LOCAL_RANK (torchrun / mpirun style), modulo the visible devices.
int device_from_options()
{
    int dev = options::getInt("engine_device", -1);
    if (dev < 0) {
        const char* lr = getenv("LOCAL_RANK");
        dev = lr ? atoi(lr) : 0;
    }
    const int n = kh_device_count();
    return n > 0 ? ((dev % n) + n) % n : 0;
}

// libtorch-default-like initialisation for a fresh model (the reference's fresh NN is random
// too, nn.cpp:107-116; bit-wise equality with libtorch's generator is not a goal):
// conv / linear weights and biases uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)), BatchNorm identity.
std::vector<float> fresh_blob(int F, int C, int R)
{
    std::vector<float> b;
    b.reserve(kh_weight_count(F, C, R));
    std::mt19937 rng(std::random_device{}());
    auto uni = [&](size_t n, int fan_in) {
        const float k = 1.0f / std::sqrt((float)fan_in);
        std::uniform_real_distribution<float> u(-k, k);
        for (size_t i = 0; i < n; ++i) b.push_back(u(rng));
    };
    auto bn = [&](int c) {
        b.insert(b.end(), c, 1.0f); b.insert(b.end(), c, 0.0f);      // weight, bias
        b.insert(b.end(), c, 0.0f); b.insert(b.end(), c, 1.0f);      // running_mean, running_var
    };
    auto convbn = [&](int co, int ci, int k) { uni((size_t)co * ci * k * k, ci * k * k); uni(co, ci * k * k); bn(co); };
    convbn(C, F, 3);
    for (int i = 0; i < 2 * R; ++i) convbn(C, C, 3);
    convbn(KH_POLICY_MID, C, 1);
    uni((size_t)KH_POLICY_PLANES * KH_POLICY_MID, KH_POLICY_MID); uni(KH_POLICY_PLANES, KH_POLICY_MID);
    convbn(1, C, 1);
    uni((size_t)KH_VALUE_WIDTH * 64, 64); uni(KH_VALUE_WIDTH, 64);
    return b;
}

constexpr int32_t MAGIC = 0x574D414B;   // "KAMW", same container as kami_amd/weights.py

}  // namespace

// host copies of the installed parameter sets (for write()), keyed by object: the class layout in
// nn.h stays free of std containers so that it matches what kami.cpp was compiled to expect
static std::mutex g_store_mu;
static std::map<const NN*, std::vector<float>> g_store;

void NN::create(int dtype)
{
    this->dtype = dtype;
    kh_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.width = width; cfg.height = height; cfg.features = features; cfg.psize = psize;
    cfg.filters = filters; cfg.residuals = residuals;
    cfg.dtype = dtype; cfg.value_mode = KH_VALUE_REFERENCE_FLAT;
    cfg.device = device.index >= 0 ? device.index : device_from_options();      // (a replica names its device itself)
    device.index = cfg.device;
    int rc = kh_create(&cfg, &eng);
    if (rc) raise(rc);
}

void NN::load_blob(const float* blob, size_t n, int generation)
{
    int rc = kh_load_weights(eng, blob, n, generation);
    if (rc) raise(rc);
    std::lock_guard<std::mutex> lk(g_store_mu);
    g_store[this].assign(blob, blob + n);
}

NN::NN(int width, int height, int features, int psize, bool force_cpu) :
    width(width), height(height), features(features), psize(psize)
{
    device.index = -1;
    if (force_cpu)
        throw std::runtime_error("kami::NN(force_cpu): the MI355X engine has no CPU path");
    filters = options::getInt("filters", 256);          // nn.cpp:42
    residuals = options::getInt("residuals", 2);        // nn.cpp:43
    int dtype = dtype_from_options();
    if (dtype != KH_F32 && (filters > 256 || features > 256)) {
        // bf16/f16: whole-network kernel up to 64 filters, per-layer MFMA kernels up to 256; beyond
        // that only the plain fp32 kernels apply
        std::cerr << "kami::NN: " << filters << " filters / " << features
                  << " features is outside the bf16/f16 kernels' range, using the fp32 HIP path\n";
        dtype = KH_F32;
    }
    create(dtype);
    // the reference computes in fp32; this engine's DEFAULT is bf16 operands with fp32 accumulation (option key
    // "engine_dtype": f32 | f16 | bf16) — said once per model so that nobody compares numerics unknowingly
    std::cerr << "kami::NN: MI355X engine, " << (dtype == KH_F32 ? "fp32 (exact, the reference's arithmetic)" : dtype == KH_F16 ? "f16 operands, fp32 accumulate" : "bf16 operands, fp32 accumulate")
              << " [engine_dtype], device " << device.index << "\n";
    std::vector<float> b = fresh_blob(features, filters, residuals);
    load_blob(b.data(), b.size(), 0);
}

NN::NN(NN* other) :
    width(other->width), height(other->height), features(other->features), psize(other->psize),
    filters(other->filters), residuals(other->residuals), device(other->device), dtype(other->dtype)
{
    int rc = kh_clone(other->eng, &eng);                // nn.cpp:130-153
    if (rc) raise(rc);
    std::lock_guard<std::mutex> lk(g_store_mu);
    g_store[this] = g_store[other];
}

// One evaluator per GPU: the same network on device `device_index` (modulo the visible devices, so that two
// replicas on one GPU can rehearse the multi-GPU control flow), with other's weights and generation.
NN::NN(NN* other, int device_index) :
    width(other->width), height(other->height), features(other->features), psize(other->psize),
    filters(other->filters), residuals(other->residuals)
{
    const int n = kh_device_count();
    device.index = n > 0 ? ((device_index % n) + n) % n : 0;
    create(other->dtype);
    sync_from(other);
}

void NN::sync_from(NN* other)
{
    std::vector<float> blob;
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        blob = g_store[other];
    }
    load_blob(blob.data(), blob.size(), other->get_generation());
}

NN::~NN()
{
    kh_destroy(eng);
    std::lock_guard<std::mutex> lk(g_store_mu);
    g_store.erase(this);
}

int NN::get_generation() { return kh_generation(eng); }

void NN::infer(float* input, int batch, float* policy, float* value)
{
    int rc = kh_infer(eng, input, batch, policy, value);
    if (rc) raise(rc);
}

void NN::train(int trajectories, float* inputs, float* obs_p, float* obs_v, bool detect_anomaly)
{
    // nn.cpp:236-238: the three option keys the reference reads
    kh_train_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.lr = (float)options::getInt("training_mlr", 5) / 1000.0f;
    cfg.epochs = options::getInt("training_epochs", 8);
    cfg.batch = options::getInt("training_batchsize", 8);
    cfg.detect_anomaly = detect_anomaly ? 1 : 0;                 // nn.cpp:231-232,329-344: the same three messages
    float first = 0.0f, last = 0.0f;
    int rc = kh_train(eng, inputs, obs_p, obs_v, trajectories, &cfg, &first, &last);
    if (rc) raise(rc);
    std::vector<float> blob(kh_weight_count(features, filters, residuals));
    if ((rc = kh_get_weights(eng, blob.data(), blob.size()))) raise(rc);
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        g_store[this] = blob;
    }
    std::cout << "Generated model " << get_generation() << ", average loss " << first << " to " << last << " over "
              << cfg.epochs << " epochs\n";                      // nn.cpp:372
}

void NN::write(std::string path)
{
    std::vector<float> blob;
    {
        std::lock_guard<std::mutex> lk(g_store_mu);
        blob = g_store[this];
    }
    std::ofstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error("cannot open " + path + " for writing");
    int32_t hdr[8] = { MAGIC, features, filters, residuals, get_generation(), 0, 0, 0 };
    f.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
    f.write(reinterpret_cast<const char*>(blob.data()), (std::streamsize)(blob.size() * sizeof(float)));
    std::cout << "Saved model to " << path << std::endl;          // nn.cpp:201
}

// nn.cpp:204-222.  Reads what the reference's NN::write leaves on disk — a libtorch archive (module.save +
// the "generation" IValue), parsed by the engine without libtorch (kh_checkpoint_read) — as well as the
// engine's own KAMW container that write() above produces.
void NN::read(std::string path)
{
    int F = 0, C = 0, R = 0, gen = 0;
    size_t n = 0;
    int rc = kh_checkpoint_read(path.c_str(), &F, &C, &R, &gen, nullptr, 0, &n);
    if (rc) raise(rc);
    if (F != features || C != filters || R != residuals)
        throw std::runtime_error(path + ": network shape does not match this NN");
    std::vector<float> blob(n);
    if ((rc = kh_checkpoint_read(path.c_str(), nullptr, nullptr, nullptr, nullptr, blob.data(), blob.size(), nullptr))) raise(rc);
    load_blob(blob.data(), blob.size(), gen);
}

NN* NN::clone() { return new NN(this); }
