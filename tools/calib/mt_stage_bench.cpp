// Where does the sequential stage of pfg_legacy_streams spend its time?  Stage timings on this host for one
// T = N = 1000 window's worth of work (7300 MT19937 blocks): build with
//   g++ -O3 -std=c++17 -pthread -I include -ffp-contract=off tools/calib/mt_stage_bench.cpp -o tools/calib/mt_stage_bench
#include <chrono>
#include <cstdio>
#include "../../stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc/pfg_legacy_rng.hip"

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
    static uint32_t key[624], out[624];
    for (int i = 0; i < 624; ++i) key[i] = 1812433253u * (i + 17) + 12345u;
    const int B = 7300;
    const bool avx2 = __builtin_cpu_supports("avx2");
    static double x1[156], x2[156], r2[156], d[312];
    std::vector<double> q1(600016), q2(600016), q3(600016);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now();
        for (int b = 0; b < B; ++b) { if (avx2) mt_block_avx2(key); else mt_block_generic(key); }
        double t1 = now();
        for (int b = 0; b < B; ++b) { if (avx2) mt_temper_avx2(key, out, 0); else mt_temper_generic(key, out, 0); key[b % 624] ^= out[5]; }
        double t2 = now();
        for (int b = 0; b < B; ++b) { if (avx2) doubles_avx2(out, 312, d); else doubles_generic(out, 312, d); out[b % 624] += (uint32_t)(d[7] * 8); }
        double t3 = now();
        for (int b = 0; b < B; ++b) { if (avx2) candidates_avx2(out, 156, x1, x2, r2); else candidates_generic(out, 156, x1, x2, r2); out[b % 624] += (uint32_t)(r2[7] * 8); }
        double t4 = now();
        size_t np = 0;
        for (int b = 0; b < B / 2; ++b) {
            candidates_avx2(out, 156, x1, x2, r2);
            size_t got = 0;
            if (avx2) accept_avx2(x1, x2, r2, 156, 1000, q1.data() + np, q2.data() + np, q3.data() + np, got);
            else accept_generic(x1, x2, r2, 156, 1000, q1.data() + np, q2.data() + np, q3.data() + np, got);
            np = (np + got) % 500000;
            out[b % 624] += (uint32_t)np;
        }
        double t5 = now();
        double acc = 0;
        for (size_t q = 0; q < 500000; ++q) { double a, bb; transform(0.3, 0.4, 0.25 + 1e-7 * (q % 1000), a, bb); acc += a; }
        double t6 = now();
        std::printf("avx2 %d | recurrence %.2f ms | temper %.2f | doubles(all blocks) %.2f | candidates(all blocks) %.2f | candidates+accept(half) %.2f | 5e5 transforms %.2f  (%g)\n",
                    (int)avx2, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3, (t6 - t5) * 1e3, acc);
    }
    return 0;
}
