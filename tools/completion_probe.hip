// How soon after a kernel's last instruction does the host learn that it has finished?  A 64-thread kernel sleeps for
// ~25 us, then writes a word to page-locked host memory (system-scope store).  The host polls that word and, in a
// second run, hipStreamQuery.  Prints: launch-call time, launch -> word seen, launch -> hipStreamQuery success, and the
// cost of one hipStreamQuery call on a busy stream.      hipcc -O2 --offload-arch=gfx950 completion_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <vector>
__global__ void nap(unsigned* done, unsigned serial, long long cycles, long long* dur)
{
    const long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        *dur = __builtin_readcyclecounter() - t0;
        __threadfence_system();
        __hip_atomic_store(done, serial, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int main()
{
    unsigned* done; long long* dur;
    (void)hipHostMalloc(&done, 64, hipHostMallocDefault);
    (void)hipHostMalloc(&dur, 64, hipHostMallocDefault);
    *done = 0;
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
    const long long cyc = 2500;     // shader clocks: about a microsecond
    for (int mode = 0; mode < 3; ++mode) {
        std::vector<double> tl, tf, tq, qc;
        for (int it = 0; it < 300; ++it) {
            const unsigned serial = mode * 1000 + it + 1;
            const auto t0 = now();
            hipLaunchKernelGGL(nap, dim3(1), dim3(64), 0, st, done, serial, cyc, dur);
            const auto t1 = now();
            if (mode == 0) {
                while (__atomic_load_n(done, __ATOMIC_ACQUIRE) != serial) __builtin_ia32_pause();
                tf.push_back(us(now() - t0));
                (void)hipStreamSynchronize(st);
            } else if (mode == 1) {
                int n = 0; double q = 0;
                for (;;) { const auto a = now(); const hipError_t r = hipStreamQuery(st); q += us(now() - a); ++n; if (r == hipSuccess) break; }
                tq.push_back(us(now() - t0)); qc.push_back(q / n);
            } else {
                // both: which comes first, and by how much
                bool seen = false; double f = 0;
                for (;;) {
                    if (!seen && __atomic_load_n(done, __ATOMIC_ACQUIRE) == serial) { seen = true; f = us(now() - t0); }
                    if (hipStreamQuery(st) == hipSuccess) break;
                }
                tq.push_back(us(now() - t0)); tf.push_back(seen ? f : us(now() - t0));
            }
            tl.push_back(us(t1 - t0));
        }
        auto med = [](std::vector<double>& v) { if (v.empty()) return 0.0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        printf("mode %d: launch call %.1f us | launch -> word in host memory %.1f us | launch -> hipStreamQuery success %.1f us | one hipStreamQuery %.2f us | kernel %.0f shader clocks\n",
               mode, med(tl), med(tf), med(tq), med(qc), (double)*dur);
    }
    return 0;
}
