// Host time of the gene layers' LDS-staged schedule (build_blocked_layout) at pose0-syn scale (no GPU), stage by stage on stderr:
//   g++ -O3 -std=c++17 -pthread -DGN_LAYOUT_TIMES -I gripnet_amd/csrc tools/probes/graph_host_time.cpp
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "host_layout.hpp"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int64_t N = 19081, e_dir = argc > 1 ? atoll(argv[1]) : 715612;
    const int R = 16;
    std::mt19937_64 rng(7);
    std::vector<std::vector<int32_t>> rows((size_t)N);
    for (int64_t k = 0; k < e_dir; ++k) {
        const int32_t a = (int32_t)(rng() % N), b = (int32_t)(rng() % N);
        if (a == b) continue;
        rows[a].push_back(b); rows[b].push_back(a);
    }
    std::vector<int32_t> rp(N + 1, 0), col;
    for (int64_t i = 0; i < N; ++i) { rows[i].push_back((int32_t)i); for (int32_t c : rows[i]) col.push_back(c); rp[i + 1] = (int32_t)col.size(); }
    std::vector<float> dis((size_t)N + 64, 1.f);
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        gn_layout::BlockedLayout L = gn_layout::build_blocked_layout(N, R, rp, col, dis);
        const double t1 = now();
        uint64_t h = 1469598103934665603ull;
        for (uint16_t v : L.ids) h = (h ^ v) * 1099511628211ull;
        for (int32_t v : L.tile_off) h = (h ^ (uint32_t)v) * 1099511628211ull;
        std::printf("nnz=%lld: blocked layout %.1f ms (ok=%d, iters=%lld, hash %016llx)\n", (long long)col.size(), 1e3 * (t1 - t0), (int)L.ok,
                    (long long)L.iters_total, (unsigned long long)h);
    }
}
