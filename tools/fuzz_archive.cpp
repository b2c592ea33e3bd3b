// CPU-only fuzz harness for csrc/torch_archive.h (tools/sanitize.sh builds it with -fsanitize=address,undefined):
//     fuzz_archive <reference-written archive> <seed> <iterations>   -- byte mutations biased to the pickle and the zip directory
#include "torch_archive.h"
#include <random>
#include <fstream>
#include <iostream>
int main(int argc, char** argv) {
    std::ifstream f(argv[1], std::ios::binary);
    std::vector<char> src((std::istreambuf_iterator<char>(f)), {});
    std::mt19937 rng(atoi(argv[2]));
    int ok = 0, bad = 0;
    const int iters = atoi(argv[3]);
    // where the pickle lives (found by its magic) and the directory at the end
    size_t pk = 0;
    for (size_t i = 0; i + 8 < src.size(); ++i) if (!memcmp(&src[i], "data.pkl", 8)) { pk = i; break; }
    for (int it = 0; it < iters; ++it) {
        std::vector<char> b = src;
        const int region = rng() % 3, n = 1 << (rng() % 5);
        for (int k = 0; k < n; ++k) {
            size_t i = region == 0 ? pk + rng() % 4000 : region == 1 ? b.size() - 1 - rng() % 6000 : rng() % b.size();
            if (i < b.size()) b[i] = (char)(rng() & 255);
        }
        if (rng() % 10 == 0) b.resize(rng() % b.size());
        std::ofstream o("/tmp/kami_fuzz_m.pt", std::ios::binary); o.write(b.data(), b.size()); o.close();
        try { auto ck = kh_archive::read_checkpoint("/tmp/kami_fuzz_m.pt"); ++ok; } catch (const std::exception&) { ++bad; }
    }
    std::cout << "accepted " << ok << " refused " << bad << "\n";
}
