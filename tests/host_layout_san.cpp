// Runs the pure host-side layout code of the plan builders (gripnet_amd/csrc/host_layout.hpp) on random inputs, once on one
// builder thread and once on sixteen, under the sanitizer this file was compiled with (make -C gripnet_amd/csrc SAN=asan
// or SAN=tsan; tests/test_host_layout.py builds and runs both).  Checks that every edge lands in exactly one slot and that
// a plan does not depend on the thread count.  No HIP, no GPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <random>
#include <thread>
#include <utility>

#include <sys/wait.h>
#include <unistd.h>

#include "host_layout.hpp"

#define CHECK(cond)                                                                  \
    do {                                                                             \
        if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (%s:%d)\n", #cond, __FILE__, __LINE__); std::exit(1); } \
    } while (0)

namespace {

void set_threads(int n) {
    char buf[16];
    std::snprintf(buf, sizeof buf, "%d", n);
    setenv("GN_PLAN_THREADS", buf, 1);
}

// ---- decoder: pairing + row classes ------------------------------------------------------------------------------------
gn_layout::ClassLayout decoder_case(int64_t n, int R, int64_t e_dir, int64_t features, unsigned seed, bool check,
                                   int64_t window_bytes = gn_layout::kClsWindowBytes, bool expect_ok = true) {
    std::mt19937_64 rng(seed);
    std::vector<int64_t> hu, hv, hr;
    for (int r = 0; r < R; ++r) {
        const int64_t cnt = std::max<int64_t>(1, e_dir / (r + 1));
        std::vector<int64_t> u(cnt), v(cnt);
        for (int64_t k = 0; k < cnt; ++k) { u[k] = (int64_t)(rng() % (uint64_t)n); v[k] = (int64_t)(rng() % (uint64_t)n); }
        for (int64_t k = 0; k < cnt; ++k) { hu.push_back(u[k]); hv.push_back(v[k]); hr.push_back(r); }
        for (int64_t k = 0; k < cnt; ++k) { hu.push_back(v[k]); hv.push_back(u[k]); hr.push_back(r); }   // to_bidirection
    }
    const int64_t E = (int64_t)hu.size();
    gn::RawVec<int64_t> mirror_of;
    gn::RawVec<char> covered;
    gn_layout::pair_mirrors(hu, hv, hr, 13, mirror_of, covered);
    const gn::RawVec<int64_t> scored = gn_layout::scored_edges(covered);
    gn_layout::ClassLayout L = gn_layout::build_class_layout(hu, hv, hr, scored, mirror_of, n, features, 256, window_bytes);
    if (!check) return L;
    if (!expect_ok) { CHECK(!L.ok); return L; }             // (a list the row-class kernel does not take: the column-phase kernel serves it)
    CHECK(L.ok);
    // every edge position is written exactly once: as a scored pair's own position or as its mirror
    std::vector<int> seen((size_t)E, 0);
    for (size_t s = 0; s < (size_t)L.batches * 64; ++s) {
        if (L.own[s] != gn_layout::kNoMirror) { CHECK(L.own[s] < (uint32_t)E); seen[L.own[s]]++; }
        else CHECK(L.mirror[s] == gn_layout::kNoMirror);
        if (L.mirror[s] != gn_layout::kNoMirror) { CHECK(L.mirror[s] < (uint32_t)E); seen[L.mirror[s]]++; }
    }
    for (int64_t e = 0; e < E; ++e) CHECK(seen[e] == 1);
    // workgroups: contiguous batch ranges that together tile [0, batches) (with eight position parts workgroup 8 l + x is
    // the l-th of part x, so they are not in index order); local rows inside the class's rows; relations inside the D cache
    std::vector<std::pair<int64_t, int64_t>> ranges;
    CHECK(L.walks >= 1 && L.walks <= gn_layout::kClsMaxWalks);
    for (int g = 0; g < L.groups; ++g) {
        const int32_t* h = L.wg.data() + (size_t)g * (4 + 4 * L.walks);
        const int rows = h[1] + h[3];
        for (int wk = 0; wk < L.walks; ++wk) {                  // a workgroup's batch range of every walk of its position range
            const int32_t* d = h + 4 * wk;
            CHECK(d[5] >= d[4] && d[4] >= 0 && d[5] <= L.batches);
            if (d[5] > d[4]) ranges.push_back({d[4], d[5]});
            CHECK(d[7] >= 1 && d[7] <= gn_layout::kClsDCache);
            for (int64_t b = d[4]; b < d[5]; ++b) {
                for (int s = 0; s < 64; ++s) {
                    const uint32_t w = L.packed[(size_t)b * 64 + s];
                    CHECK((int)(w & 0xffffu) < rows && (int)(w >> 16) < rows);
                }
                for (int k = 0; k < 4; ++k) {
                    const uint32_t word = L.rel32[(size_t)b * 2 + (k >> 1)];
                    const int rel = (int)((k & 1) ? word >> 16 : word & 0xffffu);
                    CHECK(rel >= d[6] && rel < d[6] + d[7]);
                }
            }
        }
    }
    std::sort(ranges.begin(), ranges.end());
    int64_t at = 0;
    for (auto& r : ranges) { CHECK(r.first == at); at = r.second; }
    CHECK(at == L.batches);
    // eight position ranges, three classes: the parts are cut by batches and the free pairs dealt so that no workgroup is far
    // above the mean (a wave takes whole batches: one batch too many in a workgroup is one more trip for the whole launch)
    if (L.groups == 256 && n > 512 && L.batches >= 256 * 32) {
        int64_t most = 0;
        for (int g = 0; g < L.groups; ++g) {
            const int32_t* h = L.wg.data() + (size_t)g * (4 + 4 * L.walks);
            int64_t nb = 0;
            for (int wk = 0; wk < L.walks; ++wk) nb += h[5 + 4 * wk] - h[4 + 4 * wk];
            most = std::max(most, nb);
        }
        CHECK(most * L.groups <= L.batches + 3 * L.groups + L.batches / 50);
    }
    return L;
}

// ---- relational layer: destination-major streams --------------------------------------------------------------------------
gn_layout::PairLayout pair_case(int64_t N, int64_t R, int64_t E, unsigned seed, bool check) {
    std::mt19937_64 rng(seed);
    const int cus = 256;
    const int chunks = (int)gn::ceil_div(N, 32), kpad = chunks * 32;
    const int D = (int)gn::ceil_div(N, cus), G = (int)std::min<int64_t>(N, cus);
    std::vector<int64_t> src((size_t)E), dst((size_t)E);
    std::vector<uint32_t> rel((size_t)E);
    std::vector<int32_t> outdeg((size_t)N, 0);
    for (int64_t e = 0; e < E; ++e) {
        src[e] = (int64_t)(rng() % (uint64_t)N);
        dst[e] = (int64_t)((rng() % (uint64_t)N) * (rng() % 3 ? 1 : 0));   // a third of the edges into node 0: a hub, runs longer than one unit
        rel[e] = (uint32_t)(rng() % (uint64_t)R);
        outdeg[src[e]]++;
    }
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
    gn_layout::PairLayout L = gn_layout::build_pair_layout(N, R, chunks, kpad, G, D, rp, rels, perm);
    if (!check) return L;
    CHECK(L.ok);
    // every edge is one stream word that names a real att row; everything else names one of the two zero rows
    int64_t real = 0;
    std::vector<int64_t> per_rel((size_t)R, 0), want((size_t)R, 0);
    for (size_t i = 0; i < (size_t)L.blocks * 16; ++i) {
        const uint32_t row = L.stream[i] / gn_layout::kPairRowBytes;
        CHECK(L.stream[i] % gn_layout::kPairRowBytes == 0 && row <= (uint32_t)R + 1);
        if (row < (uint32_t)R) { ++real; per_rel[row]++; }
    }
    for (int64_t e = 0; e < E; ++e) want[rel[e]]++;
    CHECK(real == E);
    for (int64_t r = 0; r < R; ++r) CHECK(per_rel[r] == want[r]);
    // every destination row belongs to exactly one workgroup
    std::vector<int> owner((size_t)N, 0);
    for (int g = 0; g < G; ++g)
        for (int d = 0; d < 3; ++d)
            if (L.wg_dst[(size_t)g * 4 + d] >= 0) owner[L.wg_dst[(size_t)g * 4 + d]]++;
    for (int64_t i = 0; i < N; ++i) CHECK(owner[i] == 1);
    return L;
}

// ---- gene layers: LDS-staged gather ----------------------------------------------------------------------------------------
gn_layout::BlockedLayout blocked_case(int64_t N, int deg, int R, unsigned seed, bool check) {
    std::mt19937_64 rng(seed);
    std::vector<int32_t> rp((size_t)N + 1, 0), col;
    for (int64_t i = 0; i < N; ++i) {
        const int d = 1 + (int)(rng() % (uint64_t)(2 * deg));
        for (int k = 0; k < d; ++k) col.push_back((int32_t)(rng() % (uint64_t)N));
        rp[i + 1] = (int32_t)col.size();
    }
    std::vector<float> dis((size_t)N + 64, 0.f);
    for (int64_t i = 0; i < N; ++i) dis[i] = 1.0f / std::sqrt((float)(rp[i + 1] - rp[i]));
    gn_layout::BlockedLayout L = gn_layout::build_blocked_layout(N, R, rp, col, dis);
    if (!check) return L;
    CHECK(L.ok && !L.failed);
    int64_t real = 0;
    for (uint16_t id : L.ids) real += id != (uint16_t)N;
    CHECK(real == (int64_t)col.size());
    std::vector<int> seen((size_t)N, 0);
    for (int32_t r : L.tile_rows)
        if (r >= 0) seen[r]++;
    for (int64_t i = 0; i < N; ++i) CHECK(seen[i] == 1);
    return L;
}

// ---- relational weight gradient: units of four chunks of (relation, source) rows -----------------------------------------
gn_layout::RelGradLayout rel_grad_case(int64_t n, int64_t R, int64_t E, int groups, unsigned seed, bool check) {
    std::mt19937_64 rng(seed);
    // (relation, source)-major CSR with a hub relation (a quarter of the edges), empty relations and empty rows
    std::vector<std::vector<int32_t>> rows((size_t)(R * n));
    for (int64_t e = 0; e < E; ++e) {
        int64_t r = (e % 4 == 0) ? 0 : (int64_t)(rng() % (uint64_t)R);
        if (R > 3 && r == 2) r = 3;                                       // relation 2 stays empty
        rows[(size_t)(r * n + (int64_t)(rng() % (uint64_t)n))].push_back((int32_t)(rng() % (uint64_t)n));
    }
    std::vector<int32_t> rp((size_t)(R * n) + 1, 0), col;
    for (size_t k = 0; k < rows.size(); ++k) { col.insert(col.end(), rows[k].begin(), rows[k].end()); rp[k + 1] = (int32_t)col.size(); }
    gn_layout::RelGradLayout L = gn_layout::build_rel_grad_layout(rp.data(), col.data(), n, R, groups);
    if (!check) return L;
    CHECK(L.ok && L.groups == groups);
    const size_t entries = L.entry.size() / 4;
    CHECK(L.wg_off.size() == (size_t)groups + 1 && (size_t)L.wg_off.back() == entries && L.wave_cnt.size() == entries * gn_layout::kRelWaves);
    // every wave's units are one contiguous run, in workgroup / wave / entry order, a multiple of the ring depth per entry
    int64_t u = 0;
    std::vector<int64_t> got((size_t)(R * n), 0);                         // edges found per (relation, source) row
    std::vector<int> parts_seen((size_t)R, 0);
    for (int g = 0; g < groups; ++g)
        for (int w = 0; w < gn_layout::kRelWaves; ++w) {
            CHECK(L.wave_u0[(size_t)g * gn_layout::kRelWaves + w] == u);
            for (int e = L.wg_off[g]; e < L.wg_off[g + 1]; ++e) {
                const int rel = L.entry[4 * (size_t)e], cnt = L.wave_cnt[(size_t)e * gn_layout::kRelWaves + w];
                CHECK(rel >= 0 && rel < R && cnt % gn_layout::kRelRing == 0);
                if (w == 0) parts_seen[(size_t)rel]++;
                for (int k = 0; k < cnt; ++k, ++u)
                    for (int lg = 0; lg < 4; ++lg) {
                        const uint16_t s = L.src[(size_t)u * 4 + lg];
                        for (int t = 0; t < gn_layout::kRelChunk; ++t) {
                            const uint16_t d = L.ids[((size_t)u * 4 + lg) * gn_layout::kRelChunk + t];
                            if (s == gn_layout::kRelNoSource) { CHECK(d == (uint16_t)n); continue; }
                            CHECK(s < n && d <= n);
                            if (d < n) got[(size_t)(rel * n + s)]++;
                        }
                    }
            }
        }
    CHECK(u == L.units && L.src.size() == (size_t)(u + gn_layout::kRelSlackUnits) * 4);
    for (size_t k = 0; k < got.size(); ++k) CHECK(got[k] == rp[k + 1] - rp[k]);
    // every relation is listed, all its parts exactly once, and the parts' scratch slots do not overlap
    int slots = 0;
    for (int64_t r = 0; r < R; ++r) CHECK(parts_seen[(size_t)r] >= 1);
    for (size_t e = 0; e < entries; ++e) {
        const int rel = L.entry[4 * e], parts = L.entry[4 * e + 1], index = L.entry[4 * e + 2], slot0 = L.entry[4 * e + 3];
        CHECK(parts == parts_seen[(size_t)rel] && index >= 0 && index < parts);
        if (parts > 1) { CHECK(slot0 >= 0 && slot0 + parts <= L.scratch_slots); if (index == 0) slots += parts; }
    }
    CHECK(slots == L.scratch_slots);
    return L;
}

template <typename V>
bool same(const V& a, const V& b) { return a == b; }

}  // namespace

// balance_batch64: a permutation of the batch, never more access cycles than the input order, close to the 32 of a perfect split
void balance_case(unsigned seed) {
    std::mt19937 rng(seed);
    for (int round = 0; round < 200; ++round) {
        uint8_t cls[64];
        const int skew = round % 5;                      // 0: uniform classes; else: one class rare / absent
        for (int i = 0; i < 64; ++i) {
            int c = (int)(rng() % 16);
            if (skew == 1 && (c & 3) == 3) c &= 12;
            if (skew == 2) c &= 5;
            if (skew == 3) c = 6;
            cls[i] = (uint8_t)c;
        }
        int order[64], ident[64];
        for (int i = 0; i < 64; ++i) { order[i] = -1; ident[i] = i; }
        gn_layout::balance_batch64(cls, order);
        bool seen[64] = {false};
        for (int i = 0; i < 64; ++i) {
            if (order[i] < 0 || order[i] >= 64 || seen[order[i]]) { std::fprintf(stderr, "balance_batch64: not a permutation\n"); std::exit(1); }
            seen[order[i]] = true;
        }
        const int before = gn_layout::batch64_access_cycles(cls, ident), after = gn_layout::batch64_access_cycles(cls, order);
        if (round < 5 && seed == 5) std::printf("balance_batch64 (class mix %d): %d -> %d LDS cycles per batch (32 = no conflict)\n", skew, before, after);
        if (after > before || (skew == 0 && after > 56)) {
            std::fprintf(stderr, "balance_batch64: %d -> %d cycles (round %d)\n", before, after, round);
            std::exit(1);
        }
    }
}

// ---- relational layer of any size: degree order, weight-gradient items ----------------------------------------------------------
void general_case(int64_t N, int64_t R, int64_t E, unsigned seed) {
    std::mt19937_64 rng(seed);
    std::vector<int32_t> rp((size_t)N + 1, 0);
    for (int64_t e = 0; e < E; ++e) rp[(size_t)(1 + (e < E / 3 ? e % std::max<int64_t>(1, N / 50) : (int64_t)(rng() % (uint64_t)N)))]++;   // a few hub rows
    for (int64_t i = 0; i < N; ++i) rp[(size_t)i + 1] += rp[(size_t)i];
    std::vector<int32_t> order;
    int64_t heavy = -1;
    gn_layout::degree_order(rp, order, heavy);
    CHECK((int64_t)order.size() == N);
    std::vector<char> seen((size_t)N, 0);
    int64_t heavy_ref = 0;
    for (int64_t i = 0; i < N; ++i) {
        CHECK(order[(size_t)i] >= 0 && order[(size_t)i] < N && !seen[(size_t)order[(size_t)i]]);
        seen[(size_t)order[(size_t)i]] = 1;
        const int64_t d = rp[(size_t)order[(size_t)i] + 1] - rp[(size_t)order[(size_t)i]];
        if (i > 0) {
            const int64_t dp = rp[(size_t)order[(size_t)i - 1] + 1] - rp[(size_t)order[(size_t)i - 1]];
            CHECK(dp > d || (dp == d && order[(size_t)i - 1] < order[(size_t)i]));       // largest first, ties by row id
        }
        heavy_ref += d > gn_layout::kBasisHeavyEdges ? 1 : 0;
        if (i < heavy) CHECK(d > gn_layout::kBasisHeavyEdges);                             // the heavy rows lead the order
    }
    CHECK(heavy == heavy_ref);
    // relation ranges (a hub relation, empty relations), the whole list and two shards' edge ranges
    std::vector<int64_t> ranges((size_t)(2 * R));
    int64_t at = 0;
    for (int64_t r = 0; r < R; ++r) {
        const int64_t len = r == 0 ? E / 2 : (r % 5 == 1 ? 0 : (int64_t)(rng() % (uint64_t)std::max<int64_t>(1, 2 * (E - E / 2) / R)));
        ranges[(size_t)(2 * r)] = at;
        at = std::min(E, at + len);
        ranges[(size_t)(2 * r + 1)] = r == R - 1 ? E : at;
        if (r == R - 1) at = E;
    }
    const int64_t cuts[3][2] = {{0, E}, {0, E / 3}, {E / 3, E}};
    for (auto& c : cuts) {
        const gn_layout::RelDwItems L = gn_layout::build_rel_dw_items(ranges, c[0], c[1]);
        CHECK(L.ok && L.items.size() % 4 == 0 && L.multi.size() % 4 == 0);
        // the items tile [lo, hi) exactly, in order, inside their relation's range, at most kRelDwItemEdges edges each
        int64_t pos = c[0], slots = 0;
        for (size_t i = 0; i < L.items.size(); i += 4) {
            const int64_t r = L.items[i], a = L.items[i + 1], b = L.items[i + 2], slot = L.items[i + 3];
            CHECK(r >= 0 && r < R && a < b && b - a <= gn_layout::kRelDwItemEdges);
            CHECK(a >= ranges[(size_t)(2 * r)] && b <= ranges[(size_t)(2 * r + 1)]);
            while (pos < a) { bool inside_empty = true; (void)inside_empty; CHECK(false); }   // (no gap: a shard's relations are contiguous)
            CHECK(a == pos);
            pos = b;
            if (slot >= 0) { CHECK(slot == slots); ++slots; }
        }
        CHECK(pos == c[1] && slots == L.parts);
        for (size_t m = 0; m < L.multi.size(); m += 4) CHECK(L.multi[m + 2] > 1 && L.multi[m + 1] + L.multi[m + 2] <= L.parts);
    }
}

// After a fork the child has none of the parent's parked builder threads (host_layout.hpp, WorkerPool): its first parallel pass
// must make its own instead of waiting for threads that do not exist there.
int fork_case() {
    set_threads(16);
    gn_layout::ClassLayout before = decoder_case(645, 40, 20000, 80, 7, true);      // the parent's pool exists from here on
    const pid_t child = fork();
    CHECK(child >= 0);
    if (child == 0) {
        gn_layout::ClassLayout in_child = decoder_case(645, 40, 20000, 80, 7, true);
        _exit(same(before.packed, in_child.packed) && same(before.own, in_child.own) ? 0 : 3);
    }
    int status = 0;
    CHECK(waitpid(child, &status, 0) == child);
    CHECK(WIFEXITED(status) && WEXITSTATUS(status) == 0);
    gn_layout::ClassLayout after = decoder_case(645, 40, 20000, 80, 7, true);       // ... and still works in the parent
    CHECK(same(before.packed, after.packed));
    std::printf("host layout: builder threads ok across fork\n");
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1 && std::strcmp(argv[1], "fork") == 0) return fork_case();
    balance_case(5);
    general_case(20000, 600, 300000, 21);
    general_case(50, 3, 40, 22);
    general_case(1, 1, 1500, 23);
    // (two giant relations; one relation; a long tail of small relations - more of them per workgroup than its D cache holds: not
    // taken, as before)
    struct { int64_t n; int R; int64_t e; int64_t f; bool ok; } dec[] = {{645, 40, 20000, 80, true}, {200, 7, 3000, 80, true}, {645, 3, 50000, 48, true},
                                                                     {30, 2, 5, 16, true}, {645, 200, 150000, 80, true}, {645, 2, 300000, 80, true},
                                                                     {645, 1, 200000, 80, true}, {645, 964, 60000, 80, false}};
    for (auto& c : dec) {
        set_threads(1);
        gn_layout::ClassLayout a = decoder_case(c.n, c.R, c.e, c.f, 7, true, gn_layout::kClsWindowBytes, c.ok);
        set_threads(16);
        gn_layout::ClassLayout b = decoder_case(c.n, c.R, c.e, c.f, 7, true, gn_layout::kClsWindowBytes, c.ok);
        CHECK(same(a.packed, b.packed) && same(a.own, b.own) && same(a.mirror, b.mirror) && same(a.rel32, b.rel32) && same(a.wg, b.wg));
    }
    // the builders' host arena (host_layout.hpp): the same plan with its large arrays out of the kept block - the first hold finds no
    // block (everything by malloc, the demand noted), the later ones are served from it; a second builder meanwhile gets malloc
    {
        set_threads(16);
        std::vector<uint32_t> want_packed;
        {
            gn_layout::ClassLayout plain = decoder_case(645, 40, 20000, 80, 7, false);
            want_packed.assign(plain.packed.begin(), plain.packed.end());
            CHECK(!gn::arena_owns(plain.packed.data()));
        }
        for (int pass = 0; pass < 3; ++pass) {
            gn::ArenaHold hold;
            CHECK(hold.held);
            gn_layout::ClassLayout l = decoder_case(645, 40, 20000, 80, 7, true);
            CHECK(l.packed.size() == want_packed.size() && std::equal(want_packed.begin(), want_packed.end(), l.packed.begin()));
            CHECK(l.packed.size() * sizeof(uint32_t) >= gn::HostArena::kMinBytes);
            CHECK(gn::arena_owns(l.packed.data()) == (pass > 0));
            std::thread other([] {
                gn::ArenaHold second;
                CHECK(!second.held);
                gn::RawVec<int> v((size_t)1 << 20);
                CHECK(!gn::arena_owns(v.data()));
            });
            other.join();
        }
    }
    for (int64_t window : {(int64_t)8 << 10, (int64_t)40 << 10}) {     // an XCD's position range walked in several sub-ranges
        set_threads(16);
        gn_layout::ClassLayout w = decoder_case(645, 40, 20000, 80, 9, true, window);
        CHECK(w.walks == (window == ((int64_t)8 << 10) ? 8 : 3));
    }
    struct { int64_t N, R, E; } pr[] = {{645, 30, 60000}, {200, 5, 9000}, {1, 1, 300}, {300, 964, 20000}};
    for (auto& c : pr) {
        set_threads(1);
        gn_layout::PairLayout a = pair_case(c.N, c.R, c.E, 11, true);
        set_threads(16);
        gn_layout::PairLayout b = pair_case(c.N, c.R, c.E, 11, true);
        CHECK(same(a.stream, b.stream) && same(a.desc, b.desc) && same(a.wave_first, b.wave_first) && same(a.wave_units, b.wave_units) &&
              same(a.wave_desc, b.wave_desc) && same(a.wg_dst, b.wg_dst));
    }
    struct { int64_t N; int deg, R; } bk[] = {{5000, 16, 32}, {700, 3, 11}, {19081, 38, 32}};
    for (auto& c : bk) {
        set_threads(1);
        gn_layout::BlockedLayout a = blocked_case(c.N, c.deg, c.R, 13, true);
        set_threads(16);
        gn_layout::BlockedLayout b = blocked_case(c.N, c.deg, c.R, 13, true);
        CHECK(same(a.ids, b.ids) && same(a.tile_off, b.tile_off) && same(a.tile_rows, b.tile_rows) && same(a.cell, b.cell));
    }
    struct { int64_t n, R, E; int groups; } rg[] = {{645, 40, 200000, 256}, {100, 7, 5000, 256}, {37, 3, 10, 8}, {300, 964, 60000, 64}};
    for (auto& c : rg) {
        set_threads(1);
        gn_layout::RelGradLayout a = rel_grad_case(c.n, c.R, c.E, c.groups, 17, true);
        set_threads(16);
        gn_layout::RelGradLayout b = rel_grad_case(c.n, c.R, c.E, c.groups, 17, true);
        CHECK(same(a.src, b.src) && same(a.ids, b.ids) && same(a.entry, b.entry) && same(a.wave_cnt, b.wave_cnt) && same(a.wave_u0, b.wave_u0));
    }
    std::printf("host layout: all builders ok on 1 and 16 threads\n");
    return 0;
}
