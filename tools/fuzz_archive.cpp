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
    // structure-aware cases the byte mutations cannot reach: random opcode streams of container-building opcodes (deep
    // nesting, memo references into containers that are still open, BUILD / SETITEMS / APPEND on adopted values)
    auto zip_of = [&](const std::string& pkl) {      // a stored-only zip: m/data.pkl, m/data/0, m/version
        std::string z, cd;
        auto u16 = [](std::string& o, unsigned v) { o.push_back((char)(v & 255)); o.push_back((char)(v >> 8 & 255)); };
        auto u32 = [&](std::string& o, unsigned v) { u16(o, v & 0xffff); u16(o, v >> 16); };
        int count = 0;
        auto add = [&](const std::string& name, const std::string& data) {
            const unsigned off = (unsigned)z.size();
            u32(z, 0x04034b50); u16(z, 20); u16(z, 0); u16(z, 0); u16(z, 0); u16(z, 0); u32(z, 0); u32(z, (unsigned)data.size()); u32(z, (unsigned)data.size());
            u16(z, (unsigned)name.size()); u16(z, 0); z += name; z += data;
            u32(cd, 0x02014b50); u16(cd, 20); u16(cd, 20); u16(cd, 0); u16(cd, 0); u16(cd, 0); u16(cd, 0); u32(cd, 0); u32(cd, (unsigned)data.size());
            u32(cd, (unsigned)data.size()); u16(cd, (unsigned)name.size()); u16(cd, 0); u16(cd, 0); u16(cd, 0); u16(cd, 0); u32(cd, 0); u32(cd, off); cd += name;
            ++count;
        };
        add("m/data.pkl", pkl); add("m/data/0", std::string(16, '\0')); add("m/version", "3\n");
        const unsigned cdoff = (unsigned)z.size();
        z += cd;
        u32(z, 0x06054b50); u16(z, 0); u16(z, 0); u16(z, count); u16(z, count); u32(z, (unsigned)cd.size()); u32(z, cdoff); u16(z, 0);
        return z;
    };
    const char ops[] = { ')', '}', ']', '(', 't', '\x85', '\x86', 'a', 'e', 's', 'u', 'b', 'N', 'K', 'q', 'h', '\x81', 'c' };
    int sok = 0, sbad = 0;
    for (int it = 0; it < iters; ++it) {
        std::string pkl = "\x80\x02";
        const int len = 1 << (4 + rng() % 14);
        const int mode = rng() % 4;
        for (int k = 0; k < len; ++k) {
            char op = mode == 0 ? '\x85' : mode == 1 ? (k < len / 2 ? ']' : 'a') : ops[rng() % sizeof ops];
            pkl.push_back(op);
            if (op == 'K' || op == 'q' || op == 'h') pkl.push_back((char)(rng() % 4));
            if (op == 'c') pkl += "__torch__.M\nM\n";
            if (k == 0 && mode == 0) pkl.insert(2, ")");
        }
        pkl.push_back('.');
        const std::string z = zip_of(pkl);
        std::ofstream o("/tmp/kami_fuzz_m.pt", std::ios::binary); o.write(z.data(), z.size()); o.close();
        try { auto ck = kh_archive::read_checkpoint("/tmp/kami_fuzz_m.pt"); ++sok; } catch (const std::exception&) { ++sbad; }
    }
    std::cout << "accepted " << ok << " refused " << bad << "; structured: accepted " << sok << " refused " << sbad << "\n";
}
