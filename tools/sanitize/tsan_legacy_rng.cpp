// ThreadSanitizer driver of the threaded legacy-stream generator (csrc/pfg_legacy_rng.hip is host-only C++): runs
// pfg_legacy_streams with 1, 2, 3 and 4 worker threads on several window shapes and checks that every thread count
// returns bitwise the same streams and final state.  Built and run by tools/sanitize/run_tsan.sh (CPU build; GPU
// sanitizers are not available on the pool).
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "pfgrad.h"

static void seed_state(uint32_t *key, uint32_t s) {      // MT19937 init_genrand
    key[0] = s;
    for (int i = 1; i < 624; ++i) key[i] = 1812433253u * (key[i - 1] ^ (key[i - 1] >> 30)) + (uint32_t)i;
}

int main() {
    const int shapes[][2] = {{1000, 1000}, {1000, 24}, {257, 33}, {4000, 40}, {64, 3}};
    int bad = 0;
    for (const auto &sh : shapes) {
        const int N = sh[0], T = sh[1];
        std::vector<double> ref_z0, ref_u, ref_z;
        uint32_t ref_key[624];
        int32_t ref_pos = 0, ref_hg = 0;
        double ref_g = 0.0;
        for (int threads = 1; threads <= 4; ++threads) {
            uint32_t key[624];
            seed_state(key, 12345u + (uint32_t)N);
            int32_t pos = 624, hg = 0;
            double g = 0.0;
            std::vector<double> z0(N), u((size_t)T * N), z((size_t)T * N);
            const int rc = pfg_legacy_streams(key, &pos, &hg, &g, N, T, z0.data(), u.data(), z.data(), threads);
            if (rc != 0) { std::printf("rc=%d\n", rc); return 2; }
            if (threads == 1) {
                ref_z0 = z0; ref_u = u; ref_z = z; std::memcpy(ref_key, key, sizeof key); ref_pos = pos; ref_hg = hg; ref_g = g;
            } else if (z0 != ref_z0 || u != ref_u || z != ref_z || std::memcmp(ref_key, key, sizeof key) || pos != ref_pos ||
                       hg != ref_hg || std::memcmp(&g, &ref_g, 8)) {
                std::printf("N=%d T=%d threads=%d: differs from the single-thread stream\n", N, T, threads);
                bad = 1;
            }
        }
        std::printf("N=%d T=%d: 1..4 threads bitwise identical\n", N, T);
    }
    return bad;
}
