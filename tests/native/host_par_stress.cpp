// tests/native/host_par_stress.cpp — fork/join helper threads (sp1_amd/csrc/host_par.hpp): thousands of short jobs of random
// length and grain in many scopes must give the serial sums; a second scope opened while the first is alive runs inline.
// Built and run by tests/test_host_par.py (g++, no GPU).
#include "host_par.hpp"
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
using namespace sp1hip;
int main() {
    std::vector<uint64_t> v(100000);
    for (size_t i = 0; i < v.size(); i++) v[i] = i * 2654435761u;
    int bad = 0;
    for (int scope = 0; scope < 60; scope++) {
        HostPar::Scope par;
        if (scope == 0) printf("threads %d\n", par.threads());
        for (int job = 0; job < 2000; job++) {
            const size_t n = (size_t)((scope * 7919 + job * 104729) % 3000);
            uint64_t parts[16] = {0};
            par.run(n, 16 + job % 50, [&](int part, size_t b, size_t e) { uint64_t a = 0; for (size_t i = b; i < e; i++) a += v[i]; parts[part] += a; });
            uint64_t got = 0, want = 0;
            for (int q = 0; q < 16; q++) got += parts[q];
            for (size_t i = 0; i < n; i++) want += v[i];
            bad += got != want;
        }
    }
    // two scopes at once: the second one runs inline
    { HostPar::Scope a; HostPar::Scope b; printf("nested: %d %d\n", a.threads(), b.threads()); }
    auto t0 = std::chrono::steady_clock::now();
    { HostPar::Scope par; for (int job = 0; job < 20000; job++) par.run(4096, 1, [&](int, size_t, size_t) {}); }
    auto t1 = std::chrono::steady_clock::now();
    printf("bad %d; empty job round trip %.2f us\n", bad, std::chrono::duration<double, std::micro>(t1 - t0).count() / 2e4);
    return bad != 0;
}
