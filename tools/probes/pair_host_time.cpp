// Host time of the relational layer's destination-major layout (build_pair_layout) at pose0-syn scale (no GPU), stage by stage on stderr:
//   g++ -O3 -std=c++17 -pthread -DGN_LAYOUT_TIMES -I gripnet_amd/csrc tools/probes/pair_host_time.cpp
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include "host_layout.hpp"
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const int64_t N = 645, R = 964, e_dir = argc > 1 ? atoll(argv[1]) : 1000000;
    const int cus = 256;
    const int chunks = (int)gn::ceil_div(N, 32), kpad = chunks * 32;
    const int D = (int)gn::ceil_div(N, cus), G = (int)std::min<int64_t>(N, cus);
    std::mt19937_64 rng(7);
    std::vector<int64_t> src, dst;
    std::vector<uint32_t> rel;
    double wsum = 0; for (int r = 1; r <= R; ++r) wsum += std::pow((double)r, -0.8);
    for (int r = 0; r < R; ++r) {
        const int64_t cnt = std::max<int64_t>(1, (int64_t)(e_dir * std::pow((double)(r + 1), -0.8) / wsum));
        for (int64_t k = 0; k < cnt; ++k) {
            const int64_t a = rng() % N, b = rng() % N;
            src.push_back(a); dst.push_back(b); rel.push_back(r);
            src.push_back(b); dst.push_back(a); rel.push_back(r);
        }
    }
    const int64_t E = (int64_t)src.size();
    std::vector<int32_t> outdeg((size_t)N, 0);
    for (int64_t e = 0; e < E; ++e) outdeg[src[e]]++;
    std::vector<int32_t> order((size_t)N), perm((size_t)kpad, (int32_t)N), kpos((size_t)N);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return outdeg[x] > outdeg[y]; });
    for (int64_t k = 0; k < N; ++k) { perm[k] = order[k]; kpos[order[k]] = (int32_t)k; }
    std::vector<int64_t> idx((size_t)E);
    std::iota(idx.begin(), idx.end(), 0);
    auto key = [&](int64_t e) { return dst[e] * kpad + kpos[src[e]]; };
    std::stable_sort(idx.begin(), idx.end(), [&](int64_t x, int64_t y) { return key(x) < key(y); });
    std::vector<int32_t> rp((size_t)N * kpad + 1, 0);
    std::vector<uint32_t> rels((size_t)E);
    for (int64_t i = 0; i < E; ++i) { rels[i] = rel[idx[i]]; rp[key(idx[i]) + 1]++; }
    for (size_t i = 1; i < rp.size(); ++i) rp[i] += rp[i - 1];
    for (int rep = 0; rep < 3; ++rep) {
        const double t0 = now();
        gn_layout::PairLayout L = gn_layout::build_pair_layout(N, R, chunks, kpad, G, D, rp, rels, perm);
        const double t1 = now();
        uint64_t h = 1469598103934665603ull;
        for (uint32_t v : L.stream) h = (h ^ v) * 1099511628211ull;
        for (uint32_t v : L.desc) h = (h ^ v) * 1099511628211ull;
        for (uint32_t v : L.wave_first) h = (h ^ v) * 1099511628211ull;
        std::printf("E=%lld: pair layout %.1f ms (ok=%d, blocks=%lld, hash %016llx)\n", (long long)E, 1e3 * (t1 - t0), (int)L.ok, (long long)L.blocks,
                    (unsigned long long)h);
    }
}
