// ref_harness.cpp — drives the UNMODIFIED reference (sources compiled where they lie under
// /root/reference; recipe: oracle/Makefile) to produce golden vectors and CPU timings.
// TEST INFRASTRUCTURE.  Output binary goes to oracle/_ref/kami_ref only.
//
// Nothing of the reference is copied here: this file only CALLS its public API
//   kami::NNModule / kami::NN   (kami/nn/nn.h:23-73)
//   kami::Env                   (kami/env.h:41-485)
//   kami::options::setInt       (kami/options.h:10)
//
// Sub-commands
//   infer   <weights.bin> <input.f32> <B> <policy.f32> <value_flat.f32> <value_full.f32>
//   observe <seed> <ngames> <maxply> <out.bin>
//   bench   <F> <C> <R> <B> <iters> <threads>
//   games   <seed> <ngames> <maxply> <out.bin>      random playouts: played action + terminal verdict per ply
//   train   <weights.bin> <n> <inputs.f32> <obs_p.f32> <obs_v.f32> <mlr> <epochs> <tbatch> <out_blob.f32>
//                                                   NN::train (nn.cpp:224-377) from the given weights; dumps the
//                                                   trained parameters + BatchNorm statistics in blob order
//   mcts    <nodes> <nmoves> <out.txt>              the reference's own MCTS (kami/mcts.h) under a
//                                                   deterministic synthetic evaluator, noise off
//   export  <weights.bin> <out.pt>                  blob -> a checkpoint written by the reference's own NN::write
//                                                   (nn.cpp:189-202): the torch archive an existing kami model is
//   convert <archive.pt> <F> <C> <R> <out.bin>      NN::read (nn.cpp:204-222) of such an archive -> KAMW blob
#include "kami/nn/nn.h"
#include "kami/env.h"
#include "kami/mcts.h"
#include "kami/options.h"

#include <torch/torch.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>

using namespace kami;

static std::vector<char> slurp(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) { fprintf(stderr, "cannot open %s\n", path); exit(2); }
    return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static void dump(const char* path, const void* p, size_t bytes)
{
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

// Canonical blob order documented in include/kami_hip.h (kh_weight_count).
static std::vector<std::string> canonical_names(int R)
{
    std::vector<std::string> n;
    auto convbn = [&](const std::string& conv, const std::string& bn) {
        n.push_back(conv + ".weight"); n.push_back(conv + ".bias");
        n.push_back(bn + ".weight"); n.push_back(bn + ".bias");
        n.push_back(bn + ".running_mean"); n.push_back(bn + ".running_var");
    };
    convbn("conv1", "batchnorm1");
    for (int i = 0; i < R; ++i) {
        std::string r = "residual" + std::to_string(i);
        convbn(r + ".conv1", r + ".batchnorm1");
        convbn(r + ".conv2", r + ".batchnorm2");
    }
    convbn("policyconv", "pbatchnorm");
    n.push_back("policyconv2.weight"); n.push_back("policyconv2.bias");
    convbn("valueconv", "vbatchnorm");
    n.push_back("valuefc.weight"); n.push_back("valuefc.bias");
    return n;
}

struct Blob { int F, C, R, gen; const float* data; size_t n; std::vector<char> raw; };

static Blob read_blob(const char* path)
{
    Blob b;
    b.raw = slurp(path);
    const int32_t* h = (const int32_t*)b.raw.data();
    if (b.raw.size() < 32 || h[0] != 0x574d414b /* "KAMW" */) { fprintf(stderr, "bad weight blob\n"); exit(2); }
    b.F = h[1]; b.C = h[2]; b.R = h[3]; b.gen = h[4];
    b.data = (const float*)(b.raw.data() + 32);
    b.n = (b.raw.size() - 32) / 4;
    return b;
}

static void fill_module(NNModule& mod, const Blob& b)
{
    torch::NoGradGuard g;
    auto params = mod.named_parameters(true);
    auto bufs = mod.named_buffers(true);
    const float* p = b.data;
    size_t used = 0;
    for (auto& name : canonical_names(b.R)) {
        torch::Tensor t;
        if (params.contains(name)) t = params[name];
        else if (bufs.contains(name)) t = bufs[name];
        else { fprintf(stderr, "reference module has no tensor %s\n", name.c_str()); exit(2); }
        size_t n = t.numel();
        if (used + n > b.n) { fprintf(stderr, "blob too short at %s\n", name.c_str()); exit(2); }
        t.copy_(torch::from_blob((void*)p, t.sizes(), torch::kFloat32));
        p += n; used += n;
    }
    if (used != b.n) { fprintf(stderr, "blob has %zu floats, module consumed %zu\n", b.n, used); exit(2); }
}

static int cmd_infer(int argc, char** argv)
{
    if (argc < 8) return 1;
    Blob b = read_blob(argv[2]);
    int B = atoi(argv[4]);
    options::setInt("filters", b.C);
    options::setInt("residuals", b.R);

    // A module of the reference's own class, filled with the blob, saved the way NN::write
    // does (nn.cpp:189-202), then ingested by the reference's NN::read (nn.cpp:204-222).
    auto mod = std::make_shared<NNModule>(8, 8, b.F, PSIZE);
    fill_module(*mod, b);
    mod->eval();
    std::string tmp = std::string(argv[5]) + ".model.pt";
    {
        torch::serialize::OutputArchive a;
        mod->save(a);
        a.write("generation", torch::IValue(b.gen));
        a.save_to(tmp);
    }
    NN net(8, 8, b.F, PSIZE, /*force_cpu=*/true);
    net.read(tmp);
    remove(tmp.c_str());
    if (net.get_generation() != b.gen) { fprintf(stderr, "generation mismatch\n"); return 2; }

    std::vector<char> in = slurp(argv[3]);
    if (in.size() != (size_t)B * 64 * b.F * 4) { fprintf(stderr, "input size mismatch\n"); return 2; }
    std::vector<float> policy((size_t)B * PSIZE), value(B);
    net.infer((float*)in.data(), B, policy.data(), value.data());          // nn.cpp:155-187
    dump(argv[5], policy.data(), policy.size() * 4);
    dump(argv[6], value.data(), value.size() * 4);

    // Whole [B,256] value tensor straight from NNModule::forward (nn.cpp:59-91).
    {
        torch::NoGradGuard g;
        auto x = torch::from_blob(in.data(), { B, 8, 8, b.F }, torch::kFloat32);
        auto out = mod->forward(x);
        auto vh = out[1].contiguous();
        auto ph = out[0].contiguous();
        if (vh.numel() != (int64_t)B * 256) { fprintf(stderr, "unexpected value shape\n"); return 2; }
        dump(argv[7], vh.data_ptr<float>(), (size_t)vh.numel() * 4);
        // NN::infer must agree bit-for-bit with the module it read from disk (test/nndisk.cpp:24-29)
        if (memcmp(ph.data_ptr<float>(), policy.data(), policy.size() * 4) != 0) {
            fprintf(stderr, "NN::infer and NNModule::forward disagree\n"); return 2;
        }
    }
    return 0;
}

// observe record, little-endian:
//   int32 ply; int32 nact; char fen[104]; int32 actions[128]; float obs[1920]
static int cmd_observe(int argc, char** argv)
{
    if (argc < 6) return 1;
    unsigned seed = (unsigned)atoi(argv[2]);
    int ngames = atoi(argv[3]), maxply = atoi(argv[4]);
    FILE* f = fopen(argv[5], "wb");
    if (!f) return 2;
    std::mt19937 rng(seed);
    long nrec = 0;
    for (int g = 0; g < ngames; ++g) {
        Env e;
        for (int ply = 0; ply <= maxply; ++ply) {
            std::vector<int> acts = e.actions();                 // env.h:398-423
            int32_t hdr[2] = { e.ply(), (int32_t)acts.size() };
            char fen[104] = { 0 };
            std::string s = e.print();                           // env.h:425-430
            strncpy(fen, s.c_str(), sizeof(fen) - 1);
            int32_t a[128] = { 0 };
            for (size_t i = 0; i < acts.size() && i < 128; ++i) a[i] = acts[i];
            std::vector<float> obs(OBSIZE);
            e.observe(obs.data());                               // env.h:202-262
            fwrite(hdr, 4, 2, f); fwrite(fen, 1, 104, f); fwrite(a, 4, 128, f);
            fwrite(obs.data(), 4, OBSIZE, f);
            ++nrec;
            // keep playing past draw conditions (50-move / repetition) while a legal move
            // exists, so that ply > 255 and large half-move clocks are covered
            if (acts.empty()) break;
            e.push(acts[rng() % acts.size()]);                   // env.h:264-271
        }
    }
    fclose(f);
    printf("%ld records\n", nrec);
    return 0;
}

static int cmd_bench(int argc, char** argv)
{
    if (argc < 8) return 1;
    int F = atoi(argv[2]), C = atoi(argv[3]), R = atoi(argv[4]), B = atoi(argv[5]);
    int iters = atoi(argv[6]), threads = atoi(argv[7]);
    options::setInt("filters", C);
    options::setInt("residuals", R);
    if (threads > 0) { torch::set_num_threads(threads); }
    NN net(8, 8, F, PSIZE, true);
    std::mt19937 rng(20240607);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    std::vector<float> in((size_t)B * 64 * F), policy((size_t)B * PSIZE), value(B);
    for (auto& v : in) v = u(rng);
    net.infer(in.data(), B, policy.data(), value.data());       // warm-up
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < iters; ++i) net.infer(in.data(), B, policy.data(), value.data());
    double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"evals_per_s\": %.3f, \"seconds\": %.6f, \"iters\": %d, \"batch\": %d, \"threads\": %d, "
           "\"features\": %d, \"filters\": %d, \"residuals\": %d}\n",
           (double)iters * B / s, s, iters, B, torch::get_num_threads(), F, C, R);
    return 0;
}

static int cmd_train(int argc, char** argv)
{
    if (argc < 11) return 1;
    Blob b = read_blob(argv[2]);
    const int n = atoi(argv[3]);
    options::setInt("filters", b.C);
    options::setInt("residuals", b.R);
    options::setInt("training_mlr", atoi(argv[7]));                // nn.cpp:236 (lr = mlr / 1000)
    options::setInt("training_epochs", atoi(argv[8]));             // nn.cpp:237
    options::setInt("training_batchsize", atoi(argv[9]));          // nn.cpp:238
    auto mod = std::make_shared<NNModule>(8, 8, b.F, PSIZE);
    fill_module(*mod, b);
    std::string tmp = std::string(argv[10]) + ".model.pt";
    {
        torch::serialize::OutputArchive a;
        mod->save(a);
        a.write("generation", torch::IValue(b.gen));
        a.save_to(tmp);
    }
    NN net(8, 8, b.F, PSIZE, /*force_cpu=*/true);
    net.read(tmp);
    std::vector<char> in = slurp(argv[4]), op = slurp(argv[5]), ov = slurp(argv[6]);
    if (in.size() != (size_t)n * 64 * b.F * 4 || op.size() != (size_t)n * PSIZE * 4 || ov.size() != (size_t)n * 4) {
        fprintf(stderr, "training data size mismatch\n"); return 2;
    }
    net.train(n, (float*)in.data(), (float*)op.data(), (float*)ov.data());   // nn.cpp:224-377
    net.write(tmp);                                                // nn.cpp:189-202
    auto out = std::make_shared<NNModule>(8, 8, b.F, PSIZE);
    {
        torch::serialize::InputArchive a;
        a.load_from(tmp);
        out->load(a);
    }
    remove(tmp.c_str());
    auto params = out->named_parameters(true);
    auto bufs = out->named_buffers(true);
    std::vector<float> blob;
    for (auto& name : canonical_names(b.R)) {
        torch::Tensor t = params.contains(name) ? params[name] : bufs[name];
        t = t.contiguous().to(torch::kFloat32);
        const float* p = t.data_ptr<float>();
        blob.insert(blob.end(), p, p + t.numel());
    }
    dump(argv[10], blob.data(), blob.size() * 4);
    printf("generation %d floats %zu\n", net.get_generation(), blob.size());
    return 0;
}

// games record, little-endian: int32 ply; int32 action (played from this position, -1 = none);
//   int32 terminal; float value; char fen[104]
static int cmd_games(int argc, char** argv)
{
    if (argc < 6) return 1;
    unsigned seed = (unsigned)atoi(argv[2]);
    int ngames = atoi(argv[3]), maxply = atoi(argv[4]);
    FILE* f = fopen(argv[5], "wb");
    if (!f) return 2;
    std::mt19937 rng(seed);
    long nrec = 0;
    for (int g = 0; g < ngames; ++g) {
        Env e;
        for (int ply = 0; ply <= maxply; ++ply) {
            float value = 0.0f;
            int32_t term = e.terminal(&value) ? 1 : 0;             // env.h:286-384
            std::vector<int> acts = e.actions();
            // like `observe`: keep playing through the draw verdicts while a legal move exists
            int32_t action = (acts.empty() || ply == maxply) ? -1 : acts[rng() % acts.size()];
            int32_t hdr[3] = { e.ply(), action, term };
            char fen[104] = { 0 };
            std::string s = e.print();
            strncpy(fen, s.c_str(), sizeof(fen) - 1);
            fwrite(hdr, 4, 3, f); fwrite(&value, 4, 1, f); fwrite(fen, 1, 104, f);
            ++nrec;
            if (action < 0) break;
            e.push(action);
        }
    }
    fclose(f);
    printf("%ld records\n", nrec);
    return 0;
}

// Synthetic evaluator shared with tests/test_search.py: FNV-1a of the FEN, splitmix64 per action.
static uint64_t fnv1a(const std::string& s)
{
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
static uint64_t splitmix(uint64_t x)
{
    x += 0x9e3779b97f4a7c15ull;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
    return x ^ (x >> 31);
}

static int cmd_mcts(int argc, char** argv)
{
    if (argc < 5) return 1;
    int nodes = atoi(argv[2]), nmoves = atoi(argv[3]);
    FILE* f = fopen(argv[4], "w");
    if (!f) return 2;
    options::setFloat("mcts_noise_weight", 0.0f);                // mcts.h:91: priors = policy / ptotal exactly
    MCTS tree;                                                     // mcts.h:66
    std::vector<float> obs(OBSIZE), policy(PSIZE);
    for (int m = 0; m < nmoves; ++m) {
        while (tree.n() < nodes) {
            if (!tree.select(obs.data())) continue;                // mcts.h:183-259 (terminal leaves backprop inside)
            const uint64_t h = fnv1a(tree.get_env().print());
            double sum = 0.0;
            for (int a = 0; a < PSIZE; ++a) { policy[a] = (float)(splitmix(h + (uint64_t)a) % 16777213ull + 1); sum += policy[a]; }
            for (int a = 0; a < PSIZE; ++a) policy[a] = (float)(policy[a] / sum);
            const float value = ((float)(splitmix(h ^ 0x7777) % 2001) - 1000.0f) / 1000.0f;
            tree.expand(policy.data(), value);                     // mcts.h:261-327
        }
        fprintf(f, "move %d fen %s\n", m, tree.get_env().print().c_str());
        fprintf(f, "root n %d w %.9g\n", tree.root->n, tree.root->w);
        for (Node* c : tree.root->children)
            fprintf(f, "child %d n %d w %.9g p %.9g\n", c->action, c->n, c->w, c->p);
        int picked = tree.pick(0.0f);                              // mcts.h:139-181
        fprintf(f, "pick %d\n", picked);
        tree.push(picked);                                         // mcts.h:114-137
        float v;
        if (tree.get_env().terminal(&v)) { fprintf(f, "terminal %g\n", v); break; }
    }
    fclose(f);
    return 0;
}

static void write_kamw(const char* path, int F, int C, int R, int gen, const std::vector<float>& blob)
{
    int32_t h[8] = { 0x574d414b, F, C, R, gen, 0, 0, 0 };
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(h, 4, 8, f) != 8 || fwrite(blob.data(), 4, blob.size(), f) != blob.size()) { fprintf(stderr, "cannot write %s\n", path); exit(2); }
    fclose(f);
}

// blob -> reference checkpoint: the module is filled from the blob, ingested by NN::read and written
// back by the reference's own NN::write, so the file is byte for byte what kami leaves on disk.
static int cmd_export(int argc, char** argv)
{
    if (argc < 4) return 1;
    Blob b = read_blob(argv[2]);
    options::setInt("filters", b.C);
    options::setInt("residuals", b.R);
    auto mod = std::make_shared<NNModule>(8, 8, b.F, PSIZE);
    fill_module(*mod, b);
    std::string tmp = std::string(argv[3]) + ".tmp";
    {
        torch::serialize::OutputArchive a;
        mod->save(a);
        a.write("generation", torch::IValue(b.gen));
        a.save_to(tmp);
    }
    NN net(8, 8, b.F, PSIZE, /*force_cpu=*/true);
    net.read(tmp);
    remove(tmp.c_str());
    net.write(argv[3]);                                            // nn.cpp:189-202
    return 0;
}

static int cmd_convert(int argc, char** argv)
{
    if (argc < 7) return 1;
    const int F = atoi(argv[3]), C = atoi(argv[4]), R = atoi(argv[5]);
    options::setInt("filters", C);
    options::setInt("residuals", R);
    NN net(8, 8, F, PSIZE, /*force_cpu=*/true);
    net.read(argv[2]);                                             // nn.cpp:204-222
    // NN keeps its module private: re-load the archive into a module of the same class to walk the tensors
    auto out = std::make_shared<NNModule>(8, 8, F, PSIZE);
    {
        torch::serialize::InputArchive a;
        a.load_from(argv[2]);
        out->load(a);
    }
    auto params = out->named_parameters(true);
    auto bufs = out->named_buffers(true);
    std::vector<float> blob;
    for (auto& name : canonical_names(R)) {
        torch::Tensor t = params.contains(name) ? params[name] : bufs[name];
        t = t.contiguous().to(torch::kFloat32);
        const float* p = t.data_ptr<float>();
        blob.insert(blob.end(), p, p + t.numel());
    }
    write_kamw(argv[6], F, C, R, net.get_generation(), blob);
    printf("generation %d floats %zu\n", net.get_generation(), blob.size());
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 2) { fprintf(stderr, "usage: kami_ref infer|observe|bench|games|mcts|train|export|convert ...\n"); return 1; }
    std::string c = argv[1];
    try {
        if (c == "infer") return cmd_infer(argc, argv);
        if (c == "observe") return cmd_observe(argc, argv);
        if (c == "bench") return cmd_bench(argc, argv);
        if (c == "games") return cmd_games(argc, argv);
        if (c == "mcts") return cmd_mcts(argc, argv);
        if (c == "train") return cmd_train(argc, argv);
        if (c == "export") return cmd_export(argc, argv);
        if (c == "convert") return cmd_convert(argc, argv);
    } catch (std::exception& e) {
        fprintf(stderr, "kami_ref: %s\n", e.what());
        return 3;
    }
    return 1;
}
