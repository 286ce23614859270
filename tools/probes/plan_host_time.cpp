// Host time of the decoder plan's builders at pose0-syn scale (no GPU), stage by stage on stderr:
//   g++ -O3 -std=c++17 -pthread -DGN_LAYOUT_TIMES -I gripnet_amd/csrc tools/probes/plan_host_time.cpp
// (the ids as the plan sees them: narrowed to 16 bits on the device)
#include <chrono>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <random>
#include "host_layout.hpp"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int64_t n = 645; const int R = 964; const int64_t e_dir = argc > 1 ? atoll(argv[1]) : 1000000;
    std::mt19937_64 rng(7);
    std::vector<uint16_t> hu, hv, hr;
    double wsum = 0; for (int r = 1; r <= R; ++r) wsum += std::pow((double)r, -0.8);
    for (int r = 0; r < R; ++r) {
        const int64_t cnt = std::max<int64_t>(1, (int64_t)(e_dir * std::pow((double)(r + 1), -0.8) / wsum));
        std::vector<int64_t> u(cnt), v(cnt);
        for (int64_t k = 0; k < cnt; ++k) { u[k] = rng() % n; v[k] = rng() % n; }
        for (int64_t k = 0; k < cnt; ++k) { hu.push_back(u[k]); hv.push_back(v[k]); hr.push_back(r); }
        for (int64_t k = 0; k < cnt; ++k) { hu.push_back(v[k]); hv.push_back(u[k]); hr.push_back(r); }
    }
    const int64_t E = hu.size();
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        gn::RawVec<int64_t> mirror_of; gn::RawVec<char> covered;
        gn_layout::pair_mirrors(hu, hv, hr, 13, mirror_of, covered);
        double t1 = now();
        const gn::RawVec<int64_t> scored = gn_layout::scored_edges(covered);
        double t2 = now();
        const int64_t NBs = (scored.size() + 63) / 64;
        std::vector<int> slots((size_t)NBs * 64);
        gn::parallel_for(NBs, 64, [&](int64_t b0, int64_t b1) {
            int64_t cu[64], cv[64];
            for (int64_t bi = b0; bi < b1; ++bi) {
                const int count = (int)std::min<int64_t>(64, (int64_t)scored.size() - bi * 64);
                for (int k = 0; k < count; ++k) { cu[k] = hu[scored[bi * 64 + k]]; cv[k] = hv[scored[bi * 64 + k]]; }
                gn_layout::deal_batch(cu, cv, count, slots.data() + bi * 64);
            }
        });
        double t3 = now();
        gn_layout::ClassLayout L = gn_layout::build_class_layout(hu, hv, hr, scored, mirror_of, n, 80, 256);
        double t4 = now();
        uint64_t h = 1469598103934665603ull;
        for (uint32_t v : L.packed) h = (h ^ v) * 1099511628211ull;
        for (uint32_t v : L.own) h = (h ^ v) * 1099511628211ull;
        std::printf("hash %016llx ", (unsigned long long)h);
        std::printf("E=%lld: pair_mirrors %.1f ms, scored %.1f ms, column-phase deal %.1f ms, class layout %.1f ms (ok=%d)\n", (long long)E,
                    1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), (int)L.ok);
    }
}
