// Pure host-side layout code of the plan builders: which edge goes into which slot of which stream.  No HIP in here -
// the .hip files copy inputs to the host, call these, and upload what comes back - so that this file also builds with
// plain g++, where tests/host_layout_san.cpp runs every builder under AddressSanitizer / UBSan and under ThreadSanitizer
// with GN_PLAN_THREADS=16 (sanitizers; `make -C gripnet_amd/csrc SAN=asan|tsan`).
#pragma once

#include <algorithm>
#include <atomic>
#include <condition_variable>
#ifdef GN_LAYOUT_TIMES
#include <chrono>
#include <cstdio>
#endif
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <type_traits>
#include <numeric>
#include <thread>
#include <utility>
#include <vector>

#include <unistd.h>
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif

namespace gn {

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// The builder threads: parked between builds (a plan of pose0-syn runs thirty parallel passes of a fraction of a
// millisecond each, and starting fifteen threads for every pass was 0.3-0.5 ms of it - a quarter of the decoder plan's build
// time).  One pass at a time uses the pool (a second builder, or a pass started from inside a pass, starts its own threads
// as before); the pool belongs to the process that made it - after a fork the child makes its own at its first pass (the
// parent's threads do not exist there) - and is never torn down.
struct WorkerPool {
    std::mutex run_lock;               // held by the pass that is using the pool
    std::mutex m;                      // guards everything below
    std::condition_variable wake, done;
    std::vector<std::thread> threads;
    std::function<void(int64_t)> job;  // job(chunk)
    int64_t chunks = 0, next = 0, pending = 0;
    uint64_t generation = 0;
    long owner = 0;                    // the process the threads live in
};
inline void pool_worker(WorkerPool* p, uint64_t seen) {
    std::unique_lock<std::mutex> lk(p->m);
    for (;;) {
        p->wake.wait(lk, [&] { return p->generation != seen; });
        seen = p->generation;
        while (p->next < p->chunks) {
            const int64_t c = p->next++;
            lk.unlock();
            p->job(c);
            lk.lock();
            if (--p->pending == 0) p->done.notify_one();
        }
    }
}
inline WorkerPool* worker_pool() {
    static std::atomic<WorkerPool*> pool{nullptr};
    WorkerPool* p = pool.load(std::memory_order_acquire);
    const long me = (long)getpid();
    if (p != nullptr && p->owner == me) return p;
    WorkerPool* fresh = new WorkerPool();                     // (a pool inherited through fork is left alone: its threads are gone)
    fresh->owner = me;
    if (pool.compare_exchange_strong(p, fresh, std::memory_order_acq_rel)) return fresh;
    delete fresh;
    p = pool.load(std::memory_order_acquire);
    return (p != nullptr && p->owner == me) ? p : nullptr;
}

// Host side of the plan builders: fn(begin, end) over contiguous chunks of [0, n) on up to GN_PLAN_THREADS (default 16:
// the CPU share of one GPU on the boxes this runs on) threads.  The chunks are fixed by n and the thread count only and
// every chunk writes its own outputs, so a plan does not depend on scheduling.
template <typename F>
inline void parallel_for(int64_t n, int64_t grain, F fn) {
    int want = 16;
    if (const char* e = getenv("GN_PLAN_THREADS")) want = std::max(1, atoi(e));
    const unsigned hw = std::thread::hardware_concurrency();
    if (hw > 0) want = std::min<int>(want, (int)hw);
    const int64_t chunks = std::max<int64_t>(1, std::min<int64_t>(want, (n + grain - 1) / std::max<int64_t>(grain, 1)));
    if (chunks <= 1 || n <= 0) { if (n > 0) fn((int64_t)0, n); return; }
    WorkerPool* p = worker_pool();
    if (p != nullptr && p->run_lock.try_lock()) {
        std::unique_lock<std::mutex> lk(p->m);
        while ((int64_t)p->threads.size() < chunks - 1) p->threads.emplace_back(pool_worker, p, p->generation);
        p->job = [&](int64_t c) { fn(n * c / chunks, n * (c + 1) / chunks); };
        p->chunks = chunks; p->next = 1; p->pending = chunks - 1;
        ++p->generation;
        lk.unlock();
        p->wake.notify_all();
        fn((int64_t)0, n / chunks);
        lk.lock();
        while (p->next < p->chunks) {                          // (chunks no parked thread has picked up yet)
            const int64_t c = p->next++;
            lk.unlock();
            fn(n * c / chunks, n * (c + 1) / chunks);
            lk.lock();
            --p->pending;
        }
        p->done.wait(lk, [&] { return p->pending == 0; });
        p->job = nullptr;
        p->chunks = 0; p->next = 0;
        lk.unlock();
        p->run_lock.unlock();
        return;
    }
    std::vector<std::thread> pool;
    pool.reserve((size_t)chunks - 1);
    for (int64_t c = 1; c < chunks; ++c) pool.emplace_back([=]() { fn(n * c / chunks, n * (c + 1) / chunks); });
    fn((int64_t)0, n / chunks);
    for (std::thread& t : pool) t.join();
}

// The builders' large host arrays come out of ONE block of the process that is kept between builds (grow-only up to
// kMaxBytes, never given back).  A plan of pose0-syn asks for ~90 MB in arrays of 2-16 MB; malloc serves each with a fresh
// mapping and free unmaps it, and in a long-lived process (bench.py after its training epochs) that traffic with the kernel
// - page faults on every first touch, the unmapping at the end - cost as much as the builders' own work: decoder plan 17 ms of
// builders, 27-32 ms measured; 17.8 ms with glibc told to keep its heap (MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_).
// A builder takes the arena for its scope (ArenaHold, FIRST local of the entry point: every array dies before it); arrays
// of at least kMinBytes are bump-allocated from it on the holder's thread, everything else - and everything while another
// builder holds the arena, and what does not fit - is plain malloc.  The block grows to 5/4 of what the last holder asked
// for, at the next acquire.  No array may outlive its hold.
struct HostArena {
    static constexpr size_t kMaxBytes = (size_t)1 << 30;
    static constexpr size_t kMinBytes = (size_t)256 << 10;
    std::mutex lock;
    std::atomic<char*> base{nullptr};  // (read by arena_owns on any thread, without the lock: see the order of the stores in ArenaHold)
    std::atomic<size_t> bytes{0};
    size_t want = 0;                   // what the block should hold at the next acquire
    std::atomic<size_t> used{0};       // bump pointer of the current hold
    std::atomic<size_t> asked{0};      // bytes requested during the current hold (served or not)
};
inline HostArena& host_arena() {
    static HostArena arena;
    return arena;
}
inline HostArena*& arena_of_this_thread() {
    static thread_local HostArena* held = nullptr;
    return held;
}
struct ArenaHold {
    bool held = false;
    ArenaHold() {
        HostArena& a = host_arena();
        if (arena_of_this_thread() != nullptr || !a.lock.try_lock()) return;     // (nested, or another builder has it: malloc)
        held = true;
        if (a.want > a.bytes.load()) {
            // a thread without the arena may be asking arena_owns() about a pointer of its own right now: it reads `bytes`, then
            // `base` - the size goes to zero before the block changes and comes back after it, so that no mix of old and new spans
            // memory that is not the block's
            char* old = a.base.load();
            a.bytes.store(0);
            a.base.store(nullptr);
            std::free(old);
            char* fresh = static_cast<char*>(std::malloc(a.want));
            a.base.store(fresh);
            a.bytes.store(fresh ? a.want : 0);
        }
        a.used.store(0); a.asked.store(0);
        arena_of_this_thread() = &a;
    }
    ArenaHold(const ArenaHold&) = delete;
    ArenaHold& operator=(const ArenaHold&) = delete;
    ~ArenaHold() {
        if (!held) return;
        HostArena& a = host_arena();
        arena_of_this_thread() = nullptr;
        const size_t asked = a.asked.load();
        a.want = std::max(a.want, std::min(HostArena::kMaxBytes, asked + asked / 4));
        a.lock.unlock();
    }
};
inline void* arena_allocate(size_t bytes) {
    HostArena* a = arena_of_this_thread();
    if (a == nullptr || bytes < HostArena::kMinBytes) return nullptr;
    const size_t padded = (bytes + 63) & ~(size_t)63;
    a->asked.fetch_add(padded);
    const size_t at = a->used.fetch_add(padded);
    if (at + padded > a->bytes.load()) { a->used.fetch_sub(padded); return nullptr; }
    return a->base.load() + at;
}
inline bool arena_owns(const void* p) {
    const HostArena& a = host_arena();
    const size_t bytes = a.bytes.load();
    const char* base = a.base.load();
    return base != nullptr && static_cast<const char*>(p) >= base && static_cast<const char*>(p) < base + bytes;
}

// A vector whose resize() leaves new elements uninitialised (the builders' large arrays are written whole by the parallel
// passes that follow: a value-initialising resize was a serial walk - and the first touch - of every page), and whose large
// blocks come from the arena above while the calling thread holds it.
template <typename T>
struct DefaultInit : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInit<U>; };
    template <typename U> void construct(U* ptr) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new (static_cast<void*>(ptr)) U; }
    template <typename U, typename... A> void construct(U* ptr, A&&... a) { ::new (static_cast<void*>(ptr)) U(std::forward<A>(a)...); }
    T* allocate(size_t n) {
        if (void* p = arena_allocate(n * sizeof(T))) return static_cast<T*>(p);
        return std::allocator<T>::allocate(n);
    }
    void deallocate(T* p, size_t n) {
        if (arena_owns(p)) return;                           // (the block is reused whole by the next holder)
        std::allocator<T>::deallocate(p, n);
    }
};
template <typename T>
using RawVec = std::vector<T, DefaultInit<T>>;

// v = n copies of `value`, written (and first touched) by the builder threads
template <typename V, typename T>
inline void parallel_assign(V& v, size_t n, T value) {
    v.resize(n);
    parallel_for((int64_t)n, 1 << 16, [&](int64_t i0, int64_t i1) { std::fill(v.begin() + i0, v.begin() + i1, value); });
}

}  // namespace gn

namespace gn_layout {

// Stage times of the builders on stderr when compiled with -DGN_LAYOUT_TIMES (tools/probes/plan_host_time.cpp); nothing otherwise.
#ifdef GN_LAYOUT_TIMES
inline void lap(const char* what) {
    static thread_local double last = 0.0;
    const double t = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    if (what) std::fprintf(stderr, "    %-34s %7.2f ms\n", what, 1e3 * (t - last));
    last = t;
}
#define GN_LAP(what) gn_layout::lap(what)
#else
#define GN_LAP(what) ((void)0)
#endif

constexpr uint32_t kNoMirror = 0xffffffffu;
constexpr int kRelDwItemEdges = 512;  // rgcn_basis.hip: edges of one work item of the general relational weight gradient
constexpr int kBasisHeavyEdges = 512; // rgcn_basis.hip: destination rows with more incoming edges are walked by a whole workgroup
constexpr int kClsDCache = 64;        // relation rows of D a workgroup of k_distmult_class keeps in LDS
constexpr int kClsSlack = 64;         // readable batches behind the last one (the kernel's prefetches run ahead unclamped)
constexpr int kClsMaxWalks = 8;       // position sub-ranges an XCD's workgroups walk one after the other (k_distmult_class)
constexpr int64_t kClsWindowBytes = 1 << 20;   // scores of one sub-range: what an XCD's 4 MB L2 holds half-written next to the streams

// ---- DistMult decoder on a static list (distmult_plan.hip) -------------------------------------------------------------
// Triples with the same unordered node pair and relation have the same score (the reference's positive list holds every
// edge in both directions, utils.py:132-138): they are paired up, the first of a pair is scored and writes both positions.
// mirror_of[e] = the later copy that takes e's score (-1: none); covered[e] = e is such a later copy.
// (the serial form: any order of relations)
template <typename V>
inline void pair_mirrors_serial(const V& hu, const V& hv, const V& hr, int node_bits,
                                gn::RawVec<int64_t>& mirror_of, gn::RawVec<char>& covered);

// Round 6: a type-sorted list (the reference's layout, utils.py:168-198) pairs up inside every relation on its own - the
// relations are dealt to the builder threads in contiguous runs of about equal edge counts, each thread with one small
// open-addressing table that it wipes by the slots it touched.  Same pairs as the serial pass (within a relation the
// edges are visited in list order).  2 M edges: 92 -> 14 ms on eight threads.
// (V: a vector of int64_t - the reference's index type - or of a narrower unsigned type the caller narrowed the validated ids to on the device)
template <typename V>
inline void pair_mirrors(const V& hu, const V& hv, const V& hr, int node_bits,
                         gn::RawVec<int64_t>& mirror_of, gn::RawVec<char>& covered) {
    const int64_t E = (int64_t)hu.size();
    GN_LAP(nullptr);
    // sorted by relation?  and the first edge of every run of equal relation ids: 64 slices of the list on the builder threads
    constexpr int kSlices = 64;
    std::vector<std::vector<int64_t>> slice_starts(kSlices);
    std::vector<char> slice_unsorted(kSlices, 0);
    gn::parallel_for(kSlices, 1, [&](int64_t s0, int64_t s1) {
        for (int64_t sl = s0; sl < s1; ++sl)
            for (int64_t e = E * sl / kSlices; e < E * (sl + 1) / kSlices; ++e)
                if (e == 0 || hr[e] != hr[e - 1]) {
                    slice_starts[(size_t)sl].push_back(e);
                    if (e > 0 && hr[e - 1] > hr[e]) slice_unsorted[(size_t)sl] = 1;
                }
    });
    bool sorted = true;
    for (char c : slice_unsorted) sorted = sorted && !c;
    if (!sorted || E < (1 << 16)) { pair_mirrors_serial(hu, hv, hr, node_bits, mirror_of, covered); return; }
    mirror_of.resize((size_t)E);                                 // (every task below wipes its own range first)
    covered.resize((size_t)E);
    std::vector<int64_t> rel_start;                              // first edge of every run of equal relation ids, then E
    for (const auto& v : slice_starts) rel_start.insert(rel_start.end(), v.begin(), v.end());
    rel_start.push_back(E);
    const int64_t runs = (int64_t)rel_start.size() - 1;
    // tasks: contiguous runs of relations of ~E / 64 edges each (a relation is never cut)
    std::vector<int64_t> task_first(1, 0);
    {
        const int64_t want = std::max<int64_t>(1, E / 64);
        int64_t acc = 0;
        for (int64_t r = 0; r < runs; ++r) {
            acc += rel_start[r + 1] - rel_start[r];
            if (acc >= want && r + 1 < runs) { task_first.push_back(r + 1); acc = 0; }
        }
        task_first.push_back(runs);
    }
    GN_LAP("mirrors: runs + tasks");
    gn::parallel_for((int64_t)task_first.size() - 1, 1, [&](int64_t t0, int64_t t1) {
        std::vector<uint64_t> keys;
        std::vector<int64_t> vals;
        std::vector<uint32_t> touched;
        for (int64_t t = t0; t < t1; ++t) {
            std::fill(mirror_of.begin() + rel_start[task_first[t]], mirror_of.begin() + rel_start[task_first[t + 1]], (int64_t)-1);
            std::fill(covered.begin() + rel_start[task_first[t]], covered.begin() + rel_start[task_first[t + 1]], (char)0);
            for (int64_t r = task_first[t]; r < task_first[t + 1]; ++r) {
                const int64_t lo_e = rel_start[r], hi_e = rel_start[r + 1];
                size_t cap = 16;
                while (cap < (size_t)(hi_e - lo_e) * 2 + 16) cap <<= 1;
                if (keys.size() < cap) { keys.assign(cap, ~(uint64_t)0); vals.assign(cap, -1); }
                const size_t mask = cap - 1;
                touched.clear();
                for (int64_t e = lo_e; e < hi_e; ++e) {
                    const uint64_t lo = (uint64_t)std::min(hu[e], hv[e]), hi = (uint64_t)std::max(hu[e], hv[e]);
                    const uint64_t key = (lo << node_bits) | hi;
                    size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & mask;
                    while (keys[h] != ~(uint64_t)0 && keys[h] != key) h = (h + 1) & mask;
                    if (keys[h] == key && vals[h] >= 0) {        // the open copy of this triple: pair up
                        mirror_of[vals[h]] = e;
                        covered[e] = 1;
                        vals[h] = -1;
                    } else {                                    // first (or third, fifth, ...) copy: stays open
                        if (keys[h] != key) touched.push_back((uint32_t)h);
                        keys[h] = key;
                        vals[h] = e;
                    }
                }
                for (uint32_t h : touched) { keys[h] = ~(uint64_t)0; vals[h] = -1; }
            }
        }
    });
    GN_LAP("mirrors: tables (parallel)");
}

template <typename V>
inline void pair_mirrors_serial(const V& hu, const V& hv, const V& hr, int node_bits,
                                gn::RawVec<int64_t>& mirror_of, gn::RawVec<char>& covered) {
    const int64_t E = (int64_t)hu.size();
    mirror_of.assign((size_t)E, -1);
    covered.assign((size_t)E, 0);
    // open addressing on a power-of-two table (keys are unique per open triple; an erased slot keeps its key with
    // value -1 so that probe chains stay intact)
    size_t cap = 1;
    while (cap < (size_t)E * 2 + 16) cap <<= 1;
    std::vector<uint64_t> keys(cap, ~(uint64_t)0);
    std::vector<int64_t> vals(cap, -1);
    for (int64_t e = 0; e < E; ++e) {
        const uint64_t lo = (uint64_t)std::min(hu[e], hv[e]), hi = (uint64_t)std::max(hu[e], hv[e]);
        const uint64_t key = ((uint64_t)hr[e] << (2 * node_bits)) | (lo << node_bits) | hi;
        size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
        while (keys[h] != ~(uint64_t)0 && keys[h] != key) h = (h + 1) & (cap - 1);
        if (keys[h] == key && vals[h] >= 0) {                // the open copy of this triple: pair up
            mirror_of[vals[h]] = e;
            covered[e] = 1;
            vals[h] = -1;
        } else {                                            // first (or third, fifth, ...) copy: stays open
            keys[h] = key;
            vals[h] = e;
        }
    }
}

// The edges the decoder scores (the others are written as their pair's mirror), in list order.
inline gn::RawVec<int64_t> scored_edges(const gn::RawVec<char>& covered) {
    constexpr int kSlices = 64;
    const int64_t E = (int64_t)covered.size();
    std::vector<int64_t> first(kSlices + 1, 0);
    gn::parallel_for(kSlices, 1, [&](int64_t s0, int64_t s1) {
        for (int64_t sl = s0; sl < s1; ++sl) {
            int64_t c = 0;
            for (int64_t e = E * sl / kSlices; e < E * (sl + 1) / kSlices; ++e) c += !covered[(size_t)e];
            first[(size_t)sl + 1] = c;
        }
    });
    for (int sl = 0; sl < kSlices; ++sl) first[(size_t)sl + 1] += first[(size_t)sl];
    gn::RawVec<int64_t> scored((size_t)first[kSlices]);
    gn::parallel_for(kSlices, 1, [&](int64_t s0, int64_t s1) {
        for (int64_t sl = s0; sl < s1; ++sl) {
            int64_t at = first[(size_t)sl];
            for (int64_t e = E * sl / kSlices; e < E * (sl + 1) / kSlices; ++e)
                if (!covered[(size_t)e]) scored[(size_t)at++] = e;
        }
    });
    return scored;
}

// A bucket of the dealers below: at most 64 entries, no allocation (with std::vector buckets a deal of 64 pairs cost ~50 us -
// sixteen vectors grown by push_back - and the decoder plan of pose0-syn spent 100 ms of eight threads in them).
struct SmallStack {
    int v[64];
    int n = 0;
    void push_back(int x) { v[n++] = x; }
    int back() const { return v[n - 1]; }
    void pop_back() { --n; }
    size_t size() const { return (size_t)n; }
    bool empty() const { return n == 0; }
};

// Deals the (up to) 64 edges of a batch to its slots.  Lane l of the wave holds slot l; wave step S works on the
// slots 4 q + S of the 16 quads q, and ds_read_b128 serves the quads in four access groups.  A cell = (step,
// access group) = four slots that hit the LDS together: its edges should have four different u % 4 and four
// different v % 4 (the bank slot of a row is (row * odd stride) % 4).
inline void deal_batch(const int64_t* u, const int64_t* v, int count, int* slot_of_edge) {
    static const int kGroupQuads[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    static const int kPerms[24][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {0, 3, 2, 1},
                                      {1, 0, 2, 3}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 0, 2}, {1, 3, 2, 0},
                                      {2, 0, 1, 3}, {2, 0, 3, 1}, {2, 1, 0, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {2, 3, 1, 0},
                                      {3, 0, 1, 2}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 1, 2, 0}, {3, 2, 0, 1}, {3, 2, 1, 0}};
    SmallStack bucket[4][4];                         // edges by (u % 4, v % 4)
    for (int e = 0; e < count; ++e) bucket[u[e] & 3][v[e] & 3].push_back(e);
    int left = count;
    for (int cell = 0; cell < 16; ++cell) {
        const int S = cell & 3, g = cell >> 2;
        int chosen[4] = {-1, -1, -1, -1};
        if (left > 0) {
            // a full cell: one edge from each (c, sigma(c)) for the permutation whose scarcest bucket is fullest
            int best = -1, best_min = 0;
            for (int p = 0; p < 24; ++p) {
                int mn = 1 << 30;
                for (int c = 0; c < 4; ++c) mn = std::min(mn, (int)bucket[c][kPerms[p][c]].size());
                if (mn > best_min) { best_min = mn; best = p; }
            }
            if (best >= 0) {
                for (int c = 0; c < 4; ++c) { auto& bk = bucket[c][kPerms[best][c]]; chosen[c] = bk.back(); bk.pop_back(); }
            } else {
                // no conflict-free quadruple left: take edges one by one, preferring unused u and v classes
                unsigned used_u = 0, used_v = 0;
                for (int k = 0; k < 4; ++k) {
                    int bc = -1, bd = -1, bscore = -1;
                    for (int c = 0; c < 4; ++c)
                        for (int dd = 0; dd < 4; ++dd) {
                            if (bucket[c][dd].empty()) continue;
                            const int score = 2 * (!((used_u >> c) & 1) + !((used_v >> dd) & 1)) * 64 + (int)bucket[c][dd].size();
                            if (score > bscore) { bscore = score; bc = c; bd = dd; }
                        }
                    if (bc < 0) break;
                    chosen[k] = bucket[bc][bd].back();
                    bucket[bc][bd].pop_back();
                    used_u |= 1u << bc; used_v |= 1u << bd;
                }
            }
        }
        for (int k = 0; k < 4; ++k)
            if (chosen[k] >= 0) { slot_of_edge[chosen[k]] = 4 * kGroupQuads[g][k] + S; --left; }
    }
}

// Cells of four pairs for one run of pairs that share class and relation: the four pairs of a cell are read by one
// 16-lane access group of ds_read_b128, so they should have four different (local row of u) % 4 and four different
// (local row of v) % 4 - the 64-byte bank slot of a row is (row * odd stride) % 4.  Cells fill whole steps first
// (step = cell / 4): a run is padded to a multiple of 16 pairs, not 64.  order[cell * 4 + k] = pair of the run, or -1.
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define GN_DEAL_SSSE3 1
// The permutation search of deal_run in six byte shuffles: sz = the sixteen stack sizes (bytes, index 4 c + d); returns the first
// of the 24 permutations (kPerms order) whose scarcest bucket is fullest, -1 when every permutation has an empty bucket.
__attribute__((target("ssse3"))) inline int best_permutation_ssse3(const uint8_t* sz, const uint8_t (*idx)[16]) {
    const __m128i s = _mm_loadu_si128(reinterpret_cast<const __m128i*>(sz));   // (sz: sixteen bytes the caller keeps 16-byte stores to)
    const __m128i lo = _mm_min_epu8(_mm_min_epu8(_mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[0]))),
                                                 _mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[1])))),
                                    _mm_min_epu8(_mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[2]))),
                                                 _mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[3])))));
    const __m128i hi = _mm_min_epu8(_mm_min_epu8(_mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[4]))),
                                                 _mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[5])))),
                                    _mm_min_epu8(_mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[6]))),
                                                 _mm_shuffle_epi8(s, _mm_loadu_si128(reinterpret_cast<const __m128i*>(idx[7])))));
    __m128i m = _mm_max_epu8(lo, hi);
    m = _mm_max_epu8(m, _mm_srli_si128(m, 8));
    m = _mm_max_epu8(m, _mm_srli_si128(m, 4));
    m = _mm_max_epu8(m, _mm_srli_si128(m, 2));
    m = _mm_max_epu8(m, _mm_srli_si128(m, 1));
    const int best_min = _mm_cvtsi128_si32(m) & 0xff;
    if (best_min == 0) return -1;
    const __m128i all = _mm_set1_epi8((char)best_min);
    const int first_lo = _mm_movemask_epi8(_mm_cmpeq_epi8(lo, all));
    if (first_lo) return __builtin_ctz((unsigned)first_lo);
    return 16 + __builtin_ctz((unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(hi, all)));
}
__attribute__((target("ssse3"))) inline void take_permutation_ssse3(uint8_t* szv, const uint8_t* take) {
    _mm_store_si128(reinterpret_cast<__m128i*>(szv), _mm_sub_epi8(_mm_load_si128(reinterpret_cast<const __m128i*>(szv)),
                                                                 _mm_load_si128(reinterpret_cast<const __m128i*>(take))));
}
#endif

inline void deal_run(const int* lu, const int* lv, int count, std::vector<int>& order) {
    static const int kPerms[24][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {0, 3, 2, 1},
                                      {1, 0, 2, 3}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 0, 2}, {1, 3, 2, 0},
                                      {2, 0, 1, 3}, {2, 0, 3, 1}, {2, 1, 0, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {2, 3, 1, 0},
                                      {3, 0, 1, 2}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 1, 2, 0}, {3, 2, 0, 1}, {3, 2, 1, 0}};
#ifdef GN_DEAL_SSSE3
    // shuffle controls: idx[c] = the bucket 4 c + sigma_p(c) of permutations p = 0..15, idx[4 + c] of p = 16..23 (then 0x80: a zero)
    struct ShuffleTable {
        uint8_t idx[8][16];
        alignas(16) uint8_t take[24][16];             // 1 at the four buckets of a permutation
        ShuffleTable() {
            for (int c = 0; c < 4; ++c)
                for (int p = 0; p < 32; ++p) idx[(p >> 4) * 4 + c][p & 15] = p < 24 ? (uint8_t)(4 * c + kPerms[p][c]) : (uint8_t)0x80;
            for (int p = 0; p < 24; ++p) {
                for (int b = 0; b < 16; ++b) take[p][b] = 0;
                for (int c = 0; c < 4; ++c) take[p][4 * c + kPerms[p][c]] = 1;
            }
        }
    };
    static const ShuffleTable table;
    static const bool use_ssse3 = __builtin_cpu_supports("ssse3");
#endif
    const int steps = (count + 15) / 16;
    order.assign((size_t)steps * 16, -1);
    // sixteen stacks by (lu % 4, lv % 4), their sizes side by side in sixteen bytes: the search below reads nothing else
    int stack[16][64];
    uint8_t sz[16] = {0};
    for (int e = count - 1; e >= 0; --e) { const int b = (lu[e] & 3) * 4 + (lv[e] & 3); stack[b][sz[b]++] = e; }   // (popped from the back: list order)
#ifdef GN_DEAL_SSSE3
    alignas(16) uint8_t szv[16];                      // the sizes again, only ever stored whole (a vector load behind byte stores stalls)
    std::memcpy(szv, sz, 16);
#endif
    int left = count;
    for (int cell = 0; cell < steps * 4 && left > 0; ++cell) {
        int chosen[4] = {-1, -1, -1, -1};
        // a full cell: one pair from each (c, sigma(c)) for the permutation whose scarcest bucket is fullest (the first of equals)
        int best = -1;
#ifdef GN_DEAL_SSSE3
        if (use_ssse3) {
            best = best_permutation_ssse3(szv, table.idx);
        } else
#endif
        {
            int best_min = 0;
            for (int p = 0; p < 24; ++p) {
                const int mn = std::min(std::min((int)sz[kPerms[p][0]], (int)sz[4 + kPerms[p][1]]), std::min((int)sz[8 + kPerms[p][2]], (int)sz[12 + kPerms[p][3]]));
                if (mn > best_min) { best_min = mn; best = p; }
            }
        }
        if (best >= 0) {
            for (int c = 0; c < 4; ++c) { const int b = c * 4 + kPerms[best][c]; chosen[c] = stack[b][--sz[b]]; }
#ifdef GN_DEAL_SSSE3
            if (use_ssse3) take_permutation_ssse3(szv, table.take[best]);
#endif
        } else {
            // no conflict-free quadruple left: pairs one by one, preferring unused u and v classes
            unsigned used_u = 0, used_v = 0;
            for (int k = 0; k < 4; ++k) {
                int bb = -1, bscore = -1;
                for (int b = 0; b < 16; ++b) {
                    if (sz[b] == 0) continue;
                    const int score = 2 * (!((used_u >> (b >> 2)) & 1) + !((used_v >> (b & 3)) & 1)) * 64 + (int)sz[b];
                    if (score > bscore) { bscore = score; bb = b; }
                }
                if (bb < 0) break;
                chosen[k] = stack[bb][--sz[bb]];
                used_u |= 1u << (bb >> 2); used_v |= 1u << (bb & 3);
            }
#ifdef GN_DEAL_SSSE3
            std::memcpy(szv, sz, 16);
#endif
        }
        for (int k = 0; k < 4; ++k)
            if (chosen[k] >= 0) { order[(size_t)cell * 4 + k] = chosen[k]; --left; }
    }
}

// The row-class encoding of the scored pairs (see k_distmult_class in distmult_plan.hip).  ok = false when a class's
// rows do not fit the LDS with `features` columns, or a workgroup's batches name more relations than its D cache holds.
struct ClassLayout {
    bool ok = false;
    int groups = 0;
    int walks = 1;                      // batch ranges per workgroup (descriptor: 4 + 4 walks ints)
    int64_t batches = 0;
    gn::RawVec<uint32_t> packed, own, mirror;
    std::vector<uint32_t> rel32;
    std::vector<int32_t> wg;
};

template <typename V>
inline ClassLayout build_class_layout(const V& hu, const V& hv, const V& hr,
                                      const gn::RawVec<int64_t>& scored, const gn::RawVec<int64_t>& mirror_of, int64_t n,
                                      int64_t features, int cus, int64_t window_bytes = kClsWindowBytes) {
    static const int kGroupQuads[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
#ifdef GN_LAYOUT_TIMES
    struct ExitLap { ~ExitLap() { GN_LAP("class: locals freed"); } } exit_lap;
#endif
    ClassLayout L;
    if (features < 16 || features % 16 != 0 || features > 128 || scored.empty() || n < 1) return L;
    const int J = (int)(features / 16), str4 = (J & 1) ? 4 * J : 4 * J + 4;
    const int64_t rows_fit = ((int64_t)160 * 1024 - (int64_t)kClsDCache * 4 * J * 16) / ((int64_t)str4 * 16);
    int nblocks = 1;
    int64_t blk = n;
    if (n > rows_fit) {
        blk = gn::ceil_div(n, 3);
        if (2 * blk > rows_fit) return L;
        nblocks = 3;
    }
    if (n > 65535) return L;
    const int nclasses = nblocks == 1 ? 1 : 3;
    auto bstart = [&](int b) { return std::min<int64_t>(n, (int64_t)b * blk); };
    auto bsize = [&](int b) { return bstart(b + 1) - bstart(b); };
    auto local_b = [&](int64_t node, int b, int k) {          // local row of `node` (of block b) in class k's table
        return (int)(b == k ? node - bstart(k) : bsize(k) + node - bstart((k + 1) % 3));
    };
    // Position parts.  A 64-byte line of the score vector holds sixteen consecutive edges of one relation - pairs of all
    // three classes - so three workgroups write it, a third each.  The list is cut into eight position ranges, one per
    // XCD (workgroup b runs on XCD b % 8), and inside a range each class gets its share of that XCD's compute units: the
    // three writers of a line share an L2, which holds the range's whole share of the scores.
    // When an XCD's share of the score vector exceeds what its L2 keeps half-written (pose2-syn: 4.2 MB of 33.5: 97 MB were
    // written for them, the lines leaving L2 a third at a time), the XCD's range is cut into `walks` sub-ranges that its
    // workgroups walk one after the other: the live window is one sub-range.
    const int64_t S = (int64_t)scored.size();
    GN_LAP(nullptr);
    // The scored pairs' endpoints, relation and block ids as compact arrays in list order (round 6): every pass below walks
    // THESE (8 bytes per pair) instead of chasing scored[] into four int64 arrays of the whole list (64 MB at pose0-syn: the
    // builder was bound by cache misses, 120 ms on eight threads).
    gn::RawVec<uint16_t> su((size_t)S), sv((size_t)S), sr((size_t)S);
    gn::RawVec<uint8_t> sbu((size_t)S), sbv((size_t)S);
    gn::RawVec<uint32_t> sm((size_t)S);                          // the pair's mirror position (kNoMirror: none)
    // (the passes over the pairs below run on the builder threads in FIXED chunks (2^15 pairs, or a 64th of the list) - what a chunk computes does
    // not depend on the thread count - or one position part per task)
    const int64_t kChunkPairs = std::max<int64_t>(1 << 15, gn::ceil_div(S, 64));   // (at most 64 chunks: their histograms stay small)
    const int64_t nchunks = gn::ceil_div(S, kChunkPairs);
    std::vector<int64_t> chunk_max_rel((size_t)nchunks, 0);
    gn::parallel_for(nchunks, 1, [&](int64_t c0, int64_t c1) {
        for (int64_t c = c0; c < c1; ++c) {
            int64_t mx = 0;
            for (int64_t i = c * kChunkPairs; i < std::min(S, (c + 1) * kChunkPairs); ++i) {
                const int64_t e = scored[(size_t)i];
                sm[(size_t)i] = mirror_of[(size_t)e] >= 0 ? (uint32_t)mirror_of[(size_t)e] : kNoMirror;
                su[(size_t)i] = (uint16_t)hu[(size_t)e]; sv[(size_t)i] = (uint16_t)hv[(size_t)e]; sr[(size_t)i] = (uint16_t)hr[(size_t)e];
                sbu[(size_t)i] = (uint8_t)(hu[(size_t)e] / blk); sbv[(size_t)i] = (uint8_t)(hv[(size_t)e] / blk);
                mx = std::max<int64_t>(mx, (int64_t)sr[(size_t)i] + 1);
            }
            chunk_max_rel[(size_t)c] = mx;
        }
    });
    GN_LAP("class: compact arrays (parallel)");
    const int parts = (cus % 8 == 0 && cus >= 24 && S >= (int64_t)64 * 4 * cus) ? 8 : 1;
    const int64_t list_bytes = (int64_t)hu.size() * 4;
    const int walks = parts == 8 ? (int)std::max<int64_t>(1, std::min<int64_t>(kClsMaxWalks, gn::ceil_div(list_bytes / 8, window_bytes))) : 1;
    const int nparts = parts * walks;                            // part p = XCD (p / walks), walk (p % walks)
    const int ngroups = nparts * nclasses;
    // The class of a pair INSIDE one block is free between the two classes that hold the block.  With eight position ranges an
    // XCD's 32 compute units go to the three classes as 11 + 11 + 10, and a class that gets ten for a third of the batches
    // has 65 per workgroup where the others have 59 (pose0-syn: 59-70 over the 256 workgroups, and the launch ends with the
    // fullest - `tools/dm_stamps.py`: last waves done 11.9-16.0 us; evened out, 61-63 and 12.8-15.2 us, the step 1.2 us
    // shorter on the same box).  So the free pairs are dealt per (part, relation, block) - a relation's pairs of a block stay
    // one run - to whichever of the two classes is further below its share of the part: class loads in the ratio of the units
    // they will get.  (What counts is a compute unit's batches, not its waves' trips: see the walks below.)
    // The position parts themselves are cut by BATCHES, not by pairs: a relation's pairs of a (part, class) are a run padded to
    // steps of sixteen slots - about eight slots per run and class - so a range of many small relations (the tail of the
    // type-sorted list) has more batches per pair than the head's few large ones (pose0-syn: 2,083 against 1,954 with
    // equal pair counts, 65 batches per workgroup against 61).  Every pair weighs 1 + 24 / (its relation's pairs).
    // (Walks of equal weight: cutting them in whole wave trips - 17 trips of sixteen batches per workgroup at pose2-syn instead
    // of 5 x 4 - changed nothing, 47.4 us either way: the loop follows a compute unit's batches, not its waves' trips.)
    gn::RawVec<int32_t> part_of((size_t)S);
    int64_t n_rel = 0;
    for (int64_t c = 0; c < nchunks; ++c) n_rel = std::max(n_rel, chunk_max_rel[(size_t)c]);
    std::vector<int64_t> part_first((size_t)nparts + 1, S);      // part p = pairs [part_first[p], part_first[p + 1]) (list order: monotone)
    {
        // a relation's pair count: per-chunk histograms, added up
        std::vector<int32_t> hist((size_t)nchunks * (size_t)n_rel, 0);
        gn::parallel_for(nchunks, 1, [&](int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; ++c) {
                int32_t* h = hist.data() + (size_t)c * (size_t)n_rel;
                for (int64_t i = c * kChunkPairs; i < std::min(S, (c + 1) * kChunkPairs); ++i) h[sr[(size_t)i]]++;
            }
        });
        std::vector<int64_t> rel_cnt((size_t)n_rel, 0);
        for (int64_t c = 0; c < nchunks; ++c)
            for (int64_t r = 0; r < n_rel; ++r) rel_cnt[(size_t)r] += hist[(size_t)c * (size_t)n_rel + (size_t)r];
        double total_w = 0.0;
        std::vector<double> rel_w((size_t)n_rel, 0.0);
        for (int64_t r = 0; r < n_rel; ++r) {
            total_w += rel_cnt[(size_t)r] > 0 ? (double)rel_cnt[(size_t)r] + 24.0 : 0.0;
            rel_w[(size_t)r] = rel_cnt[(size_t)r] > 0 ? 1.0 + 24.0 / (double)rel_cnt[(size_t)r] : 0.0;
        }
        // the weight before every chunk (from its histogram), then the chunks on their own
        std::vector<double> chunk_cum((size_t)nchunks + 1, 0.0);
        for (int64_t c = 0; c < nchunks; ++c) {
            double w = 0.0;
            for (int64_t r = 0; r < n_rel; ++r) w += (double)hist[(size_t)c * (size_t)n_rel + (size_t)r] * rel_w[(size_t)r];
            chunk_cum[(size_t)c + 1] = chunk_cum[(size_t)c] + w;
        }
        std::vector<int32_t> chunk_last((size_t)nchunks, 0);
        gn::parallel_for(nchunks, 1, [&](int64_t c0, int64_t c1) {
            for (int64_t c = c0; c < c1; ++c) {
                double cum = chunk_cum[(size_t)c];
                int32_t run_max = 0;
                for (int64_t i = c * kChunkPairs; i < std::min(S, (c + 1) * kChunkPairs); ++i) {
                    const double w = rel_w[sr[(size_t)i]];
                    run_max = std::max(run_max, (int32_t)std::min<int64_t>(nparts - 1, (int64_t)((cum + 0.5 * w) * nparts / total_w)));
                    part_of[(size_t)i] = run_max;
                    cum += w;
                }
                chunk_last[(size_t)c] = run_max;
            }
        });
        // monotone over the chunks' borders too: a chunk starts no lower than the one before it ended
        for (int64_t c = 1; c < nchunks; ++c) {
            const int32_t carry = chunk_last[(size_t)c - 1];
            chunk_last[(size_t)c] = std::max(chunk_last[(size_t)c], carry);
            for (int64_t i = c * kChunkPairs; i < std::min(S, (c + 1) * kChunkPairs) && part_of[(size_t)i] < carry; ++i) part_of[(size_t)i] = carry;
        }
        for (int p = 0; p <= nparts; ++p)
            part_first[(size_t)p] = p == nparts ? S : std::lower_bound(part_of.begin(), part_of.end(), (int32_t)p) - part_of.begin();
    }
    GN_LAP("class: position parts");
    // (A group of a few hundred pairs or more - the head of the list is one or two relations - is CUT between its two classes
    // where that evens them out: the first free_cut pairs of the group, in list order, go to the block's own class.)
    std::vector<int32_t> free_cut;                               // [part][relation][block]: pairs of the group that go to class `block`
    if (nblocks == 3 && parts == 8) {
        free_cut.assign((size_t)nparts * n_rel * 3, 0);
        const int W = cus / 8;
        std::vector<int64_t> fixed((size_t)nparts * 3, 0), flex((size_t)nparts * n_rel * 3, 0);
        gn::parallel_for(nparts, 1, [&](int64_t p0, int64_t p1) {
            for (int64_t part = p0; part < p1; ++part)
                for (int64_t i = part_first[(size_t)part]; i < part_first[(size_t)part + 1]; ++i) {
                    const int bu = sbu[(size_t)i], bv = sbv[(size_t)i];
                    if (bu != bv) fixed[(size_t)part * 3 + ((bu + 1) % 3 == bv ? bu : bv)]++;
                    else flex[((size_t)part * n_rel + sr[(size_t)i]) * 3 + bu]++;
                }
        });
        gn::parallel_for(nparts, 1, [&](int64_t p0, int64_t p1) {
        for (int part = (int)p0; part < (int)p1; ++part) {
            // the units of the classes: as even as W allows, the smaller shares to the classes with the least fixed load - of the
            // whole RANGE (its workgroups keep their class through all its walks)
            const int x0 = part / walks * walks;
            int64_t fixed_x[3] = {0, 0, 0};
            for (int wk = 0; wk < walks; ++wk)
                for (int c = 0; c < 3; ++c) fixed_x[c] += fixed[(size_t)(x0 + wk) * 3 + c];
            int order3[3] = {0, 1, 2};
            std::sort(order3, order3 + 3, [&](int a, int b) { return fixed_x[a] != fixed_x[b] ? fixed_x[a] > fixed_x[b] : a < b; });
            double share[3];
            for (int k = 0; k < 3; ++k) share[order3[k]] = (double)(W / 3 + (k < W % 3 ? 1 : 0));
            double load[3] = {(double)fixed[(size_t)part * 3], (double)fixed[(size_t)part * 3 + 1], (double)fixed[(size_t)part * 3 + 2]};
            // largest groups first (ties in (relation, block) order)
            std::vector<std::pair<int64_t, int32_t>> groups;
            for (int64_t r = 0; r < n_rel; ++r)
                for (int b = 0; b < 3; ++b)
                    if (flex[((size_t)part * n_rel + r) * 3 + b] > 0) groups.push_back({-flex[((size_t)part * n_rel + r) * 3 + b], (int32_t)(r * 3 + b)});
            std::sort(groups.begin(), groups.end());
            for (const auto& gq : groups) {
                const int b = gq.second % 3, c0 = b, c1 = (b + 2) % 3;            // the two classes that hold block b
                const double m = (double)-gq.first;
                // x pairs to c0 so that both end at the same load per unit: (load0 + x) / share0 = (load1 + m - x) / share1
                double x = (share[c0] * (load[c1] + m) - share[c1] * load[c0]) / (share[c0] + share[c1]);
                x = std::min(m, std::max(0.0, x));
                if (m < 512.0) x = x >= 0.5 * m ? m : 0.0;                       // a small group stays one run
                else x = std::min(m, std::floor(x / 64.0 + 0.5) * 64.0);
                load[c0] += x; load[c1] += m - x;
                free_cut[(size_t)part * n_rel * 3 + gq.second] = (int32_t)x;
            }
        }
        });
    }
    GN_LAP("class: free cuts");
    // scored pairs by (part, class, relation), list order inside: a stable counting sort, one position part per task (the
    // buckets of a part are its own; round 6: std::stable_sort with a comparator over 10^6 indices was the builder's longest
    // serial stretch, then the serial counting sort was)
    gn::RawVec<uint32_t> key((size_t)S), idx((size_t)S);
    std::vector<int32_t> free_seen(free_cut.size(), 0);
    const size_t nbuckets = (size_t)ngroups * (size_t)n_rel;
    std::vector<int64_t> first(nbuckets + 1, 0);
    auto bucket = [&](int64_t i) { return (size_t)(key[(size_t)i] >> 16) * (size_t)n_rel + (size_t)(key[(size_t)i] & 0xffffu); };
    gn::parallel_for(nparts, 1, [&](int64_t p0, int64_t p1) {
        for (int64_t part = p0; part < p1; ++part)
            for (int64_t i = part_first[(size_t)part]; i < part_first[(size_t)part + 1]; ++i) {
                const int bu = sbu[(size_t)i], bv = sbv[(size_t)i], rel = sr[(size_t)i];
                int c = 0;                                       // (cls_of, on the compact arrays)
                if (nblocks != 1) c = bu != bv ? ((bu + 1) % 3 == bv ? bu : bv) : ((rel & 1) ? (bu + 2) % 3 : bu);
                if (!free_cut.empty() && bu == bv) {
                    const size_t cell = ((size_t)part * n_rel + rel) * 3 + (size_t)bu;
                    c = free_seen[cell]++ < free_cut[cell] ? bu : (bu + 2) % 3;
                }
                key[(size_t)i] = (uint32_t)(part * nclasses + c) << 16 | (uint32_t)rel;
                first[bucket(i) + 1]++;
            }
    });
    GN_LAP("class: keys");
    for (size_t b = 1; b < first.size(); ++b) first[b] += first[b - 1];
    gn::parallel_for(nparts, 1, [&](int64_t p0, int64_t p1) {
        for (int64_t part = p0; part < p1; ++part)
            for (int64_t i = part_first[(size_t)part]; i < part_first[(size_t)part + 1]; ++i) idx[(size_t)first[bucket(i)]++] = (uint32_t)i;
    });
    GN_LAP("class: counting sort");
    // runs (the non-empty buckets: first[b] is now bucket b's end) -> steps of 16 slots
    struct Run { int64_t lo, hi; int grp, rel; int64_t step0; };
    std::vector<Run> runs;
    std::vector<int64_t> grp_steps(ngroups, 0);
    for (size_t b = 0; b < nbuckets; ++b) {
        const int64_t lo = b ? first[b - 1] : 0, hi = first[b];
        if (hi <= lo) continue;
        const int g = (int)(b / (size_t)n_rel);
        runs.push_back({lo, hi, g, (int)(b % (size_t)n_rel), grp_steps[g]});
        grp_steps[g] += gn::ceil_div(hi - lo, 16);
    }
    std::vector<int64_t> grp_batch0(ngroups + 1, 0);
    for (int g = 0; g < ngroups; ++g) grp_batch0[g + 1] = grp_batch0[g] + gn::ceil_div(grp_steps[g], 4);
    const int64_t NB = grp_batch0[ngroups], NBA = NB + kClsSlack;
    GN_LAP("class: runs");
    gn::RawVec<uint32_t> packed, own, mirror;
    gn::parallel_assign(packed, (size_t)NBA * 64, 0u);
    gn::parallel_assign(own, (size_t)NBA * 64, kNoMirror);
    gn::parallel_assign(mirror, (size_t)NBA * 64, kNoMirror);
    std::vector<uint16_t> rel16((size_t)NBA * 4, 0);
    GN_LAP("class: output arrays");
    // tasks of about equal PAIR counts (contiguous runs): the runs of the list's head are two orders of magnitude longer than
    // those of its tail, and equal numbers of runs per builder thread left one thread with most of the pairs
    std::vector<int64_t> task_first(1, 0);
    {
        const int64_t want = std::max<int64_t>(1, S / 256);
        int64_t acc = 0;
        for (size_t ri = 0; ri < runs.size(); ++ri) {
            acc += runs[ri].hi - runs[ri].lo;
            if (acc >= want && ri + 1 < runs.size()) { task_first.push_back((int64_t)ri + 1); acc = 0; }
        }
        task_first.push_back((int64_t)runs.size());
    }
    GN_LAP("class: tasks");
    gn::parallel_for((int64_t)task_first.size() - 1, 1, [&](int64_t t0, int64_t t1) {
        std::vector<int> lu, lv, order;
        for (int64_t ri = task_first[(size_t)t0]; ri < task_first[(size_t)t1]; ++ri) {
            const Run& run = runs[ri];
            const int count = (int)(run.hi - run.lo), cls = run.grp % nclasses;
            lu.resize(count); lv.resize(count);
            for (int k = 0; k < count; ++k) {
                const size_t i = (size_t)idx[run.lo + k];
                lu[k] = local_b(su[i], sbu[i], cls); lv[k] = local_b(sv[i], sbv[i], cls);
            }
            // dealt 64 consecutive pairs (one batch, one store instruction per lane) at a time: the 64 scores of a batch then
            // land inside a window of ~200 list positions.  Dealt over the whole run - more freedom for conflict-free
            // LDS cells - a batch's scores were spread over the relation's whole block and every lane's store became its own
            // 32-byte memory write: 202 MB written for the 33.5 MB of scores of pose2-syn (WRITE_SIZE), 71 us instead of 51
            order.clear();
            {
                std::vector<int> part;
                for (int c0 = 0; c0 < count; c0 += 64) {
                    const int cn = std::min(64, count - c0);
                    deal_run(lu.data() + c0, lv.data() + c0, cn, part);
                    for (int v : part) order.push_back(v >= 0 ? v + c0 : -1);
                }
            }
            const int steps = (int)(order.size() / 16);
            for (int t = 0; t < steps; ++t) {
                const int64_t gstep = grp_batch0[run.grp] * 4 + run.step0 + t;
                const int64_t bat = gstep >> 2;
                const int s_in = (int)(gstep & 3);
                rel16[(size_t)gstep] = (uint16_t)run.rel;
                for (int gq = 0; gq < 4; ++gq)
                    for (int k = 0; k < 4; ++k) {
                        const int pr = order[(size_t)t * 16 + gq * 4 + k];
                        const size_t slot = (size_t)bat * 64 + 4 * kGroupQuads[gq][k] + s_in;
                        const int src = pr >= 0 ? pr : 0;                       // padding repeats the run's first pair, writes nothing
                        packed[slot] = (uint32_t)lu[src] | (uint32_t)lv[src] << 16;
                        if (pr >= 0) {
                            const size_t i = (size_t)idx[run.lo + pr];
                            own[slot] = (uint32_t)scored[i];
                            mirror[slot] = sm[i];
                        }
                    }
            }
        }
    });
    GN_LAP("class: deal (parallel)");
    // steps that pad a group to whole batches: the relation of the step before them (no reload), pair (0, 0), no positions
    for (int g = 0; g < ngroups; ++g)
        for (int64_t gstep = grp_batch0[g] * 4 + grp_steps[g]; gstep < grp_batch0[g + 1] * 4; ++gstep)
            rel16[(size_t)gstep] = gstep > 0 ? rel16[(size_t)gstep - 1] : 0;
    for (int64_t gstep = NB * 4; gstep < NBA * 4; ++gstep) rel16[(size_t)gstep] = NB > 0 ? rel16[(size_t)NB * 4 - 1] : 0;
    // workgroups: inside an XCD's range, a share of its compute units per class in proportion to the class's batches (over
    // all the range's walks); a workgroup takes the same slice of its class's batches in every walk, contiguous batch
    // ranges; with eight ranges workgroup 8 l + x is the l-th of range x
    const int per_part = parts == 8 ? cus / 8 : (int)std::min<int64_t>(cus, std::max<int64_t>(NB, 1));
    struct Wg { int cls; int share, k; };
    std::vector<std::vector<Wg>> part_wgs(parts);
    auto group_of = [&](int x, int walk, int c) { return (x * walks + walk) * nclasses + c; };
    for (int x = 0; x < parts; ++x) {
        int64_t nb_part = 0;
        int live = 0;
        std::vector<int64_t> nb_cls(nclasses, 0);
        for (int c = 0; c < nclasses; ++c) {
            for (int wk = 0; wk < walks; ++wk) { const int g = group_of(x, wk, c); nb_cls[c] += grp_batch0[g + 1] - grp_batch0[g]; }
            nb_part += nb_cls[c];
            live += nb_cls[c] > 0;
        }
        const int W = parts == 8 ? per_part : std::max(std::min<int>(per_part, (int)std::max<int64_t>(nb_part, 1)), live);
        if (W < live) return L;
        std::vector<int> share(nclasses, 0);
        std::vector<double> frac(nclasses, 0.0);
        int given = 0;
        for (int c = 0; c < nclasses; ++c) {
            if (nb_cls[c] == 0) continue;
            const double want = (double)W * nb_cls[c] / std::max<int64_t>(nb_part, 1);
            share[c] = std::max(1, (int)want);
            frac[c] = want - share[c];
            given += share[c];
        }
        while (given < W && live > 0) { int best = -1; for (int c = 0; c < nclasses; ++c) if (share[c] && (best < 0 || frac[c] > frac[best])) best = c; share[best]++; frac[best] -= 1.0; ++given; }
        while (given > W) { int best = -1; for (int c = 0; c < nclasses; ++c) if (share[c] > 1 && (best < 0 || frac[c] < frac[best])) best = c; if (best < 0) break; share[best]--; frac[best] += 1.0; --given; }
        for (int c = 0; c < nclasses; ++c)
            for (int k = 0; k < share[c]; ++k) part_wgs[x].push_back({c, share[c], k});
        while (parts == 8 && (int)part_wgs[x].size() < W) part_wgs[x].push_back({0, 0, 0});      // (a range without work for all its units)
    }
    int G = 0;
    for (int x = 0; x < parts; ++x) G += (int)part_wgs[x].size();
    if (G < 1) return L;
    const int dstride = 4 + 4 * walks;
    std::vector<int32_t> wg((size_t)G * dstride, 0);
    for (int x = 0; x < parts; ++x)
        for (size_t l = 0; l < part_wgs[x].size(); ++l) {
            const Wg& w = part_wgs[x][l];
            const int c = w.cls;
            int32_t* d = wg.data() + (parts == 8 ? (size_t)(8 * l + x) : l) * dstride;
            if (nblocks == 1) { d[0] = 0; d[1] = (int32_t)n; d[2] = 0; d[3] = 0; }
            else { d[0] = (int32_t)bstart(c); d[1] = (int32_t)bsize(c); d[2] = (int32_t)bstart((c + 1) % 3); d[3] = (int32_t)bsize((c + 1) % 3); }
            for (int wk = 0; wk < walks; ++wk) {
                const int g = group_of(x, wk, c);
                const int64_t nb = grp_batch0[g + 1] - grp_batch0[g];
                const int64_t lo = w.share ? grp_batch0[g] + nb * w.k / w.share : 0, hi = w.share ? grp_batch0[g] + nb * (w.k + 1) / w.share : 0;
                int rlo = 1 << 30, rhi = -1;
                for (int64_t gstep = lo * 4; gstep < hi * 4; ++gstep) { rlo = std::min<int>(rlo, rel16[(size_t)gstep]); rhi = std::max<int>(rhi, rel16[(size_t)gstep]); }
                if (hi <= lo) { rlo = 0; rhi = 0; }
                if (rhi - rlo + 1 > kClsDCache) return L;                         // (the column-phase kernel serves such a list)
                d[4 + 4 * wk] = (int32_t)lo; d[5 + 4 * wk] = (int32_t)hi; d[6 + 4 * wk] = rlo; d[7 + 4 * wk] = rhi - rlo + 1;
            }
        }
    std::vector<uint32_t> rel32((size_t)NBA * 2);
    for (size_t i = 0; i < rel32.size(); ++i) rel32[i] = (uint32_t)rel16[2 * i] | (uint32_t)rel16[2 * i + 1] << 16;
    GN_LAP("class: workgroups + rel32");
    L.packed.swap(packed); L.own.swap(own); L.mirror.swap(mirror); L.rel32.swap(rel32); L.wg.swap(wg);
    L.groups = G; L.batches = NB; L.walks = walks;
    L.ok = true;
    return L;
}


// ---- relational layer of any size (rgcn_basis.hip): the rows' degree order and the weight gradient's work items ---------------------
// Rows by in-degree, largest first (a counting sort; ties by row id), and the number of rows a whole workgroup walks.
inline void degree_order(const std::vector<int32_t>& rp, std::vector<int32_t>& order, int64_t& heavy_rows) {
    const int64_t N = (int64_t)rp.size() - 1;
    order.assign((size_t)std::max<int64_t>(N, 0), 0);
    heavy_rows = 0;
    if (N <= 0) return;
    int64_t max_deg = 0;
    for (int64_t i = 0; i < N; ++i) max_deg = std::max<int64_t>(max_deg, rp[i + 1] - rp[i]);
    std::vector<int64_t> first((size_t)max_deg + 2, 0);
    for (int64_t i = 0; i < N; ++i) ++first[(size_t)(max_deg - (rp[i + 1] - rp[i]) + 1)];
    for (size_t d = 1; d < first.size(); ++d) first[d] += first[d - 1];
    for (int64_t i = 0; i < N; ++i) {
        order[(size_t)first[(size_t)(max_deg - (rp[i + 1] - rp[i]))]++] = (int32_t)i;
        heavy_rows += (rp[i + 1] - rp[i]) > kBasisHeavyEdges ? 1 : 0;
    }
}

// The general weight gradient's work items: every relation's share [max(start, lo), min(end, hi)) of the shard's edges, cut
// into items of at most kRelDwItemEdges edges - (relation, first edge, end edge, slot): slot = -1 for a relation's only item,
// else the item's slot among the parts that meet in a workspace; `multi`: (relation, first slot, parts, 0) of every relation of
// several items.  ok = false when a relation would need more than 65,535 parts.
struct RelDwItems {
    bool ok = true;
    std::vector<int32_t> items, multi;
    int64_t parts = 0;
};

inline RelDwItems build_rel_dw_items(const std::vector<int64_t>& ranges, int64_t lo, int64_t hi) {
    RelDwItems L;
    const int64_t R = (int64_t)ranges.size() / 2;
    for (int64_t r = 0; r < R; ++r) {
        const int64_t a = std::max<int64_t>(ranges[2 * r], lo), b = std::min<int64_t>(ranges[2 * r + 1], hi);
        if (b <= a) continue;
        const int64_t parts = gn::ceil_div(b - a, kRelDwItemEdges);
        if (parts > 65535) { L.ok = false; L.items.clear(); L.multi.clear(); L.parts = 0; return L; }
        if (parts > 1) { L.multi.push_back((int32_t)r); L.multi.push_back((int32_t)L.parts); L.multi.push_back((int32_t)parts); L.multi.push_back(0); }
        for (int64_t k = 0; k < parts; ++k) {
            L.items.push_back((int32_t)r);
            L.items.push_back((int32_t)(a + k * kRelDwItemEdges));
            L.items.push_back((int32_t)std::min<int64_t>(b, a + (k + 1) * kRelDwItemEdges));
            L.items.push_back((int32_t)(parts > 1 ? (L.parts + k) : -1));
        }
        if (parts > 1) L.parts += parts;
    }
    return L;
}

// ---- relational layer, destination-major kernel (rgcn_pair.hip) -----------------------------------------------------------
constexpr int kPairWaves = 16;             // waves of a workgroup
constexpr int kPairRowBytes = 128;         // LDS stride of an att row
constexpr int kPairMaxD = 3;               // destination rows per workgroup
constexpr int kPairSectionCap = 64;        // blocks of a section inside one unit
constexpr int kPairSlackBlocks = 192;      // readable blocks behind the last wave's stream (the window reads ahead)

// The blocks of one section: four lists of relation ids (one per lane group), `nb` blocks of four positions each.
// Lane groups 0/1 and 2/3 share the 32 lanes of one LDS access: rows of equal parity sit in the same banks, so the
// lists of a group pair are laid out even rows first / odd rows last against odd rows first / even rows last, and a
// padded position names the zero row of the parity its partner does not use.
inline void lay_out_section(const uint32_t* const (&list)[4], const int (&len)[4], int nb, uint32_t R, std::vector<uint32_t>& out) {
    const int P = 4 * nb;
    const uint32_t none = 0xffffffffu;
    // (on the stack, and without data-dependent branches: a plan lays out 10^5 sections of a dozen rows each, and the parity
    // of a relation id is a coin flip - with a branch per row the mispredictions were most of the layout's time)
    uint32_t pos[4][4 * kPairSectionCap], lead[4 * kPairSectionCap + 1], trail[4 * kPairSectionCap + 1];
    for (int k = 0; k < 4; ++k) {
        const uint32_t lead_parity = (k & 1) ? 1u : 0u;
        // first the rows of the leading parity, left aligned, in list order; then the others, right aligned, in list order
        int nl = 0, nt = 0;
        for (int i = 0; i < len[k]; ++i) {
            const uint32_t v = list[k][i];
            const int is_lead = (v & 1u) == lead_parity;
            lead[nl] = v; trail[nt] = v;
            nl += is_lead; nt += 1 - is_lead;
        }
        uint32_t* pk = pos[k];
        for (int i = 0; i < nl; ++i) pk[i] = lead[i];
        for (int i = nl; i < P - nt; ++i) pk[i] = none;
        for (int i = 0; i < nt; ++i) pk[P - nt + i] = trail[i];
    }
    const uint32_t zero_even = (R & 1u) ? R + 1 : R, zero_odd = (R & 1u) ? R : R + 1;
    const size_t base = out.size();
    out.resize(base + (size_t)nb * 16);
    uint32_t* o = out.data() + base;
    for (int k = 0; k < 4; ++k) {
        const uint32_t* mine = pos[k];
        const uint32_t* theirs = pos[k ^ 1];
        const uint32_t both_padded = (k & 1) ? zero_odd : zero_even;             // two padded partners: one of each
        for (int i = 0; i < P; ++i) {
            const uint32_t row = mine[i], other = theirs[i];
            uint32_t pad = (other & 1u) == 0u ? zero_odd : zero_even;             // the zero row of the parity the partner does not use
            pad = other == none ? both_padded : pad;
            o[(size_t)(i >> 2) * 16 + k * 4 + (i & 3)] = (row == none ? pad : row) * (uint32_t)kPairRowBytes;
        }
    }
}

// Units, per-wave streams and descriptors of the destination-major plan.  rp: [N * kpad + 1] first edge of every
// (destination, K position) cell of the edge list sorted by that key; rels: the relation of every sorted edge; perm: the
// source node at every K position (N: none).  G workgroups, up to D rows each.
struct PairLayout {
    bool ok = false;
    int64_t blocks = 0;
    gn::RawVec<uint32_t> stream;
    std::vector<uint32_t> wave_first, desc, wave_units, wave_desc;
    std::vector<int32_t> wg_dst;
};

inline PairLayout build_pair_layout(int64_t N, int64_t R, int chunks, int kpad, int G, int D, const std::vector<int32_t>& rp,
                                    const std::vector<uint32_t>& rels, const std::vector<int32_t>& perm) {
    PairLayout L;
    // A unit = (destination, chunk, slice j of <= kPairSectionCap blocks per section).  Blocks of a section = the longest of
    // its four pairs, in fours, at least one; a (destination, chunk) without any edge is no unit at all.
    // K order PER DESTINATION: its (destination, source) pairs by edge count, longest first, four consecutive ones to the
    // four lane groups of a section - lock-step partners then have (nearly) equal runs and what is left of the padding is
    // the rounding to blocks of four (pose0-syn: 1.83 -> 1.32 x the edges, tools/pair_sim.py).  kord[i][pos] = the global
    // K position (cell of `rp`) that sits at operand position pos = 32 chunk + 8 group + t of destination i; the sources
    // of a (destination, chunk) are a row of `perm2` (the kernel reads its x rows through it).
    GN_LAP(nullptr);
    std::vector<int32_t> kord((size_t)N * kpad);
    std::vector<int32_t> perm2((size_t)N * kpad);
    auto cell = [&](int64_t i, int pos) { return (size_t)i * kpad + kord[(size_t)i * kpad + pos]; };
    auto pair_len = [&](int64_t i, int pos) { const size_t c = cell(i, pos); return rp[c + 1] - rp[c]; };
    auto chunk_empty = [&](int64_t i, int ch) { return pair_len(i, 32 * ch) == 0; };   // (position 32 ch holds the chunk's longest pair)
    auto section_blocks = [&](int64_t i, int ch, int t) {
        int longest = 0;
        for (int k = 0; k < 4; ++k) longest = std::max(longest, pair_len(i, 32 * ch + 8 * k + t));
        return std::max(1, (longest + 3) / 4);
    };
    std::vector<int64_t> cost(N, 0);
    gn::parallel_for(N, 8, [&](int64_t b, int64_t e) {
        std::vector<int32_t> idx(kpad);
        std::vector<uint64_t> keyed(kpad);
        for (int64_t i = b; i < e; ++i) {
            const int32_t* r = rp.data() + (size_t)i * kpad;
            // longest first, equal lengths in K order: a counting sort when the lengths are small (they are: a few edges per
            // (destination, source) pair), else one sort of (complement of the length, position) words
            int longest = 0;
            for (int q = 0; q < kpad; ++q) longest = std::max(longest, r[q + 1] - r[q]);
            if (longest < 1024) {
                int32_t start[1025];
                std::fill(start, start + longest + 2, 0);
                for (int q = 0; q < kpad; ++q) start[longest - (r[q + 1] - r[q]) + 1]++;
                for (int l = 0; l <= longest; ++l) start[l + 1] += start[l];
                for (int q = 0; q < kpad; ++q) idx[start[longest - (r[q + 1] - r[q])]++] = q;
            } else {
                for (int q = 0; q < kpad; ++q) keyed[q] = (uint64_t)(0x7fffffff - (r[q + 1] - r[q])) << 32 | (uint32_t)q;
                std::sort(keyed.begin(), keyed.end());
                for (int q = 0; q < kpad; ++q) idx[q] = (int32_t)(uint32_t)keyed[q];
            }
            for (int q = 0; q < kpad; ++q) {
                const int ch = q >> 5, t = (q & 31) >> 2, k = q & 3;
                const size_t pos = (size_t)i * kpad + 32 * ch + 8 * k + t;
                kord[pos] = idx[q];
                // a pair without edges names no source: its x row is not read and counts as zero, so a non-finite x[s]
                // reaches only the destinations s has an edge to (0 . inf would be NaN), as in the reference's edge sum
                perm2[pos] = r[idx[q] + 1] > r[idx[q]] ? perm[idx[q]] : (int32_t)N;
            }
            // the row's cost (its K order is known now): blocks of all its sections, + a unit's split and matrix products
            int64_t blocks = 0;
            for (int ch = 0; ch < chunks; ++ch) {
                if (chunk_empty(i, ch)) continue;
                int deepest = 1;
                for (int t = 0; t < 8; ++t) {
                    const int nb = section_blocks(i, ch, t);
                    deepest = std::max(deepest, nb);
                    blocks += nb;
                }
                blocks += 12 * gn::ceil_div(deepest, kPairSectionCap);             // in block times
            }
            cost[i] = blocks;
        }
    });
    GN_LAP("pair: K order + costs (parallel)");
    // destinations to workgroups: longest first, each to the least loaded workgroup that still has room
    std::vector<std::vector<int32_t>> wg_rows(G);
    {
        std::vector<int32_t> order(N);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return cost[x] > cost[y]; });
        std::vector<int64_t> load(G, 0);
        for (int32_t i : order) {
            int best = -1;
            for (int gg = 0; gg < G; ++gg)
                if ((int)wg_rows[gg].size() < D && (best < 0 || load[gg] < load[best])) best = gg;
            wg_rows[best].push_back(i);
            load[best] += cost[i];
        }
    }
    // per workgroup: every destination row gets a share of the sixteen waves in proportion to its cost (at least one), a
    // wave a contiguous run of its row's units (chunk order) of equal cost; per wave the descriptors (eight dwords a unit,
    // pages of eight units) and the stream
    GN_LAP("pair: rows to workgroups");
    std::vector<std::vector<uint32_t>> wg_stream((size_t)G * kPairWaves), wg_desc((size_t)G * kPairWaves);
    std::vector<uint32_t> wave_units((size_t)G * kPairWaves, 0u);
    std::vector<int32_t> wg_dst((size_t)G * 4, -1);
    gn::parallel_for(G, 1, [&](int64_t b, int64_t e) {
        struct Unit { int32_t ch, slice; int64_t cost; };
        std::vector<Unit> units;
        for (int64_t gg = b; gg < e; ++gg) {
            const std::vector<int32_t>& rows = wg_rows[gg];
            const int nd = (int)rows.size();
            for (int d = 0; d < nd; ++d) wg_dst[gg * 4 + d] = rows[d];
            // waves per row: largest remainders of the proportional share
            int share[kPairMaxD] = {0, 0, 0};
            {
                int64_t total = 0;
                for (int d = 0; d < nd; ++d) total += std::max<int64_t>(cost[rows[d]], 1);
                int given = 0;
                double frac[kPairMaxD] = {0, 0, 0};
                for (int d = 0; d < nd; ++d) {
                    const double want = (double)kPairWaves * std::max<int64_t>(cost[rows[d]], 1) / total;
                    share[d] = std::max(1, (int)want);
                    frac[d] = want - share[d];
                    given += share[d];
                }
                while (given < kPairWaves) { int best = 0; for (int d = 1; d < nd; ++d) if (frac[d] > frac[best]) best = d; share[best]++; frac[best] -= 1.0; ++given; }
                while (given > kPairWaves) { int best = -1; for (int d = 0; d < nd; ++d) if (share[d] > 1 && (best < 0 || frac[d] < frac[best])) best = d; share[best]--; frac[best] += 1.0; --given; }
            }
            int wave0 = 0;
            uint32_t starts = 0;
            for (int d = 0; d < nd; ++d) {
                if (d == 1) starts |= (uint32_t)wave0;
                if (d == 2) starts |= (uint32_t)wave0 << 8;
                const int64_t i = rows[d];
                units.clear();
                int64_t total = 0;
                for (int ch = 0; ch < chunks; ++ch) {
                    if (chunk_empty(i, ch)) continue;
                    int nb[8], deepest = 1;
                    for (int t = 0; t < 8; ++t) { nb[t] = section_blocks(i, ch, t); deepest = std::max(deepest, nb[t]); }
                    for (int j = 0; j * kPairSectionCap < deepest; ++j) {
                        int64_t c = 16;                                        // x chunk, split, matrix products: in block times
                        for (int t = 0; t < 8; ++t) c += std::max(1, std::min(kPairSectionCap, nb[t] - j * kPairSectionCap));
                        units.push_back({ch, j, c});
                        total += c;
                    }
                }
                int64_t seen = 0;
                for (const Unit& un : units) {
                    // the wave of this row whose share of the cost line holds this unit's midpoint
                    const int wv = wave0 + (total > 0 ? (int)std::min<int64_t>(share[d] - 1, (2 * seen + un.cost) * share[d] / (2 * total)) : 0);
                    seen += un.cost;
                    std::vector<uint32_t>& out = wg_stream[gg * kPairWaves + wv];
                    std::vector<uint32_t>& dv = wg_desc[gg * kPairWaves + wv];
                    const size_t at = dv.size();
                    dv.resize(at + 32, 0u);
                    for (int q = 0; q < 16; ++q) {                             // the chunk's sources, 16 bits each (N: none)
                        const int32_t* ids = perm2.data() + ((size_t)i * chunks + un.ch) * 32 + 2 * q;
                        dv[at + 8 + q] = (uint32_t)ids[0] | (uint32_t)ids[1] << 16;
                    }
                    for (int t = 0; t < 8; ++t) {
                        const uint32_t* list[4];
                        int len[4], longest = 0;
                        for (int k = 0; k < 4; ++k) {
                            const size_t key_id = cell(i, 32 * un.ch + 8 * k + t);
                            const int full = rp[key_id + 1] - rp[key_id];
                            const int from = std::min(full, un.slice * kPairSectionCap * 4);
                            list[k] = rels.data() + rp[key_id] + from;
                            len[k] = std::min(full - from, kPairSectionCap * 4);
                            longest = std::max(longest, len[k]);
                        }
                        const int nb = std::max(1, (longest + 3) / 4);
                        dv[at + (t >> 2)] |= (uint32_t)nb << (8 * (t & 3));
                        lay_out_section(list, len, nb, (uint32_t)R, out);
                    }
                    wave_units[gg * kPairWaves + wv] += 1;
                }
                wave0 += share[d];
            }
            if (nd < 2) starts |= (uint32_t)kPairWaves;
            if (nd < 3) starts |= (uint32_t)kPairWaves << 8;
            wg_dst[gg * 4 + 3] = (int32_t)starts;
            for (int wv = 0; wv < kPairWaves; ++wv) {                              // whole pages
                std::vector<uint32_t>& dv = wg_desc[gg * kPairWaves + wv];
                dv.resize((dv.size() + 63) / 64 * 64, 0u);
            }
        }
    });
    GN_LAP("pair: streams (parallel)");
    std::vector<uint32_t> wave_desc((size_t)G * kPairWaves);
    std::vector<uint32_t> desc;
    for (size_t i = 0; i < wg_desc.size(); ++i) {
        wave_desc[i] = (uint32_t)(desc.size() / 32);
        desc.insert(desc.end(), wg_desc[i].begin(), wg_desc[i].end());
    }
    desc.resize(desc.size() + 128, 0u);                                         // a wave without units still reads a page (and the one after)
    std::vector<uint32_t> first((size_t)G * kPairWaves);
    size_t total = 0;
    for (size_t i = 0; i < wg_stream.size(); ++i) { first[i] = (uint32_t)(total / 16); total += wg_stream[i].size(); }
    if (total / 16 + kPairSlackBlocks >= ((size_t)1 << 31)) return L;
    gn::RawVec<uint32_t> stream(total + (size_t)kPairSlackBlocks * 16);         // (the waves' streams tile [0, total): only the slack is filled)
    std::fill(stream.begin() + (std::ptrdiff_t)total, stream.end(), (uint32_t)R * kPairRowBytes);
    gn::parallel_for((int64_t)wg_stream.size(), 64, [&](int64_t b, int64_t e) {
        for (int64_t i = b; i < e; ++i)
            if (!wg_stream[i].empty()) memcpy(stream.data() + (size_t)first[i] * 16, wg_stream[i].data(), wg_stream[i].size() * sizeof(uint32_t));
    });
    GN_LAP("pair: concatenate");
    L.blocks = (int64_t)(total / 16);
    L.stream.swap(stream); L.wave_first.swap(first); L.desc.swap(desc); L.wave_units.swap(wave_units); L.wave_desc.swap(wave_desc);
    L.wg_dst.swap(wg_dst);
    L.ok = true;
    return L;
}


// ---- gene layers, LDS-staged gather (gcn_blocked.hip) -----------------------------------------------------------------------
constexpr int kColLayoutWaves = 16;        // waves of a k_col_gather workgroup
constexpr int kColLayoutSlack = 32;        // spare iterations behind the id stream

// Destination rows -> ranges -> 16-row tiles -> (iteration, slot) of every edge, chosen for conflict-free LDS reads.
// rp / col: the destination-major CSR; dis: deg^-1/2 per node (zero padded).  R ranges of destination rows.
struct BlockedLayout {
    bool ok = false, failed = false;
    int64_t iters_total = 0;
    std::vector<int32_t> tile_off, tile_rows, cell;
    std::vector<float> tile_dis;
    gn::RawVec<uint16_t> ids;
};

inline BlockedLayout build_blocked_layout(int64_t N, int R, const std::vector<int32_t>& rp, const std::vector<int32_t>& col,
                                          const std::vector<float>& dis_host) {
    BlockedLayout L;
    GN_LAP(nullptr);
    // destination rows by degree (descending, stable), dealt to the ranges in a snake: every range gets the same number
    // of edges (to within a row) and rows of every degree; inside a range the rows stay in degree order, so that the 16
    // rows of a tile have similar lengths
    std::vector<int32_t> order(N);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return rp[x + 1] - rp[x] > rp[y + 1] - rp[y]; });
    std::vector<std::vector<int32_t>> range_rows(R);
    for (int64_t k = 0; k < N; ++k) {
        const int64_t lap = k / R, pos = k % R;
        range_rows[(lap & 1) ? R - 1 - pos : pos].push_back(order[k]);
    }
    // tiles of 16 rows; a row's edges are dealt to the 4 lanes of its quad, 4 ids per lane and iteration.  Which edge
    // goes into which (iteration, slot) is free (the order of a sum), so it is chosen for the LDS: ds_read_b64 (and
    // b32) serves lanes 0-31 and 32-63 as two access groups, conflict-free when the ids of a group differ mod 32.
    // (the ranges are scheduled independently of each other, on the plan builders' threads, and concatenated in order)
    GN_LAP("blocked: rows by degree, ranges");
    struct RangeOut { std::vector<int32_t> tile_iters, tile_rows; std::vector<uint16_t> ids; bool failed = false; };
    std::vector<RangeOut> built(R);
    const uint16_t zero_id = (uint16_t)N;
    gn::parallel_for(R, 1, [&](int64_t r0, int64_t r1) {
        // a row's ids by (id mod 32): thirty-two stacks in one flat array (filled in CSR order, popped from the back), their
        // live sizes in cnt[row][class] - the scheduler's inner loop is "the fullest class of this row that this instruction's
        // access group has not used yet", a scan of 32 counters (round 6: with a std::vector per stack the scan chased 64
        // pointers and the gene plan spent 10 ms of sixteen threads here)
        std::vector<uint16_t> flat;
        int32_t cnt[16][32], first[16][32];
        for (int64_t r = r0; r < r1; ++r) {
            const std::vector<int32_t>& rows = range_rows[r];
            RangeOut& o = built[r];
            const int tiles_r = (int)gn::ceil_div((int64_t)rows.size(), 16);
            o.tile_rows.reserve((size_t)tiles_r * 16);
            o.tile_iters.reserve((size_t)tiles_r);
            for (int tl = 0; tl < tiles_r; ++tl) {
                int32_t trow[16], rem[16];
                int iters = 0;
                size_t total = 0;
                for (int qi = 0; qi < 16; ++qi) {
                    const size_t k = (size_t)tl * 16 + qi;
                    trow[qi] = k < rows.size() ? rows[k] : -1;
                    rem[qi] = trow[qi] < 0 ? 0 : rp[trow[qi] + 1] - rp[trow[qi]];
                    total += (size_t)rem[qi];
                    iters = std::max(iters, (rem[qi] + 15) / 16);
                }
                if (flat.size() < total) flat.resize(total);
                size_t at = 0;
                for (int qi = 0; qi < 16; ++qi) {
                    for (int c = 0; c < 32; ++c) cnt[qi][c] = 0;
                    if (trow[qi] < 0) { for (int c = 0; c < 32; ++c) first[qi][c] = 0; continue; }
                    const int32_t p0 = rp[trow[qi]], p1 = rp[trow[qi] + 1];
                    for (int32_t p = p0; p < p1; ++p) cnt[qi][col[p] & 31]++;
                    for (int c = 0; c < 32; ++c) { first[qi][c] = (int32_t)at; at += (size_t)cnt[qi][c]; cnt[qi][c] = 0; }
                    for (int32_t p = p0; p < p1; ++p) { const int c = col[p] & 31; flat[(size_t)first[qi][c] + cnt[qi][c]++] = (uint16_t)col[p]; }
                }
                for (int qi = 0; qi < 16; ++qi) o.tile_rows.push_back(trow[qi]);
                const size_t base = o.ids.size();
                o.ids.resize(base + (size_t)iters * 256, zero_id);
                uint16_t* out_ids = o.ids.data() + base;
                for (int itn = 0; itn < iters; ++itn)
                    for (int s = 0; s < 4; ++s)                               // one LDS instruction: slot s of every lane
                        for (int half = 0; half < 2; ++half) {                // its two access groups: rows 0-7, rows 8-15
                            int32_t open_mask[32];                            // -1: class not used by this access group yet
                            for (int c = 0; c < 32; ++c) open_mask[c] = -1;
                            int rows_by_need[8];
                            for (int k = 0; k < 8; ++k) rows_by_need[k] = half * 8 + k;
                            std::sort(rows_by_need, rows_by_need + 8, [&](int x, int y) { return rem[x] > rem[y]; });
                            const int left = (iters - itn) * 4 - s;           // instructions left, this one included
                            for (int k = 0; k < 8; ++k) {
                                const int qi = rows_by_need[k];
                                int32_t* cq = cnt[qi];
                                for (int jl = 0; jl < 4; ++jl) {
                                    if (rem[qi] == 0) break;
                                    // must this lane take an edge now?  (4 lanes x (left - 1) instructions remain after this one)
                                    const bool must = rem[qi] > (left - 1) * 4 + (3 - jl);
                                    // the fullest open class, the lowest of equals: the largest of (count << 5 | 31 - class)
                                    int32_t bestkey = 0;
                                    for (int c = 0; c < 32; ++c) bestkey = std::max(bestkey, ((cq[c] << 5) | (31 - c)) & open_mask[c]);
                                    int best = bestkey >> 5 ? 31 - (bestkey & 31) : -1;
                                    if (best < 0) {
                                        if (!must) continue;                  // sits this slot out: the zero row
                                        bestkey = 0;
                                        for (int c = 0; c < 32; ++c) bestkey = std::max(bestkey, (cq[c] << 5) | (31 - c));
                                        best = 31 - (bestkey & 31);
                                    }
                                    out_ids[((size_t)itn * 64 + qi * 4 + jl) * 4 + s] = flat[(size_t)first[qi][best] + --cq[best]];
                                    open_mask[best] = 0;
                                    --rem[qi];
                                }
                            }
                        }
                for (int qi = 0; qi < 16; ++qi)
                    if (rem[qi] != 0) o.failed = true;
                o.tile_iters.push_back(iters);
            }
        }
    });
    GN_LAP("blocked: tiles (parallel)");
    std::vector<int32_t> tile_off(1, 0), tile_rows, cell;
    // the ranges' id streams one after the other: sized once, copied on the builder threads (6 MB at pose0-syn; appended
    // range by range on one thread this was a third of the schedule's time)
    std::vector<size_t> ids_first((size_t)R + 1, 0);
    for (int r = 0; r < R; ++r) {
        if (built[r].failed) { L.failed = true; return L; }
        ids_first[(size_t)r + 1] = ids_first[(size_t)r] + built[r].ids.size();
    }
    gn::RawVec<uint16_t> ids(ids_first[(size_t)R] + (size_t)kColLayoutSlack * 256);
    std::fill(ids.begin() + (std::ptrdiff_t)ids_first[(size_t)R], ids.end(), zero_id);
    gn::parallel_for(R, 1, [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r)
            if (!built[r].ids.empty()) memcpy(ids.data() + ids_first[(size_t)r], built[r].ids.data(), built[r].ids.size() * sizeof(uint16_t));
    });
    for (int r = 0; r < R; ++r) {
        RangeOut& o = built[r];
        const int tiles_r = (int)o.tile_iters.size();
        const int first_tile = (int)tile_off.size() - 1;
        for (int tl = 0; tl < tiles_r; ++tl) tile_off.push_back(tile_off.back() + o.tile_iters[tl]);
        tile_rows.insert(tile_rows.end(), o.tile_rows.begin(), o.tile_rows.end());
        // the range's tiles, cut into the contiguous ranges of the workgroup's waves by iterations (+ a cost per tile)
        auto cost_upto = [&](int tl) { return (int64_t)(tile_off[first_tile + tl] - tile_off[first_tile]) + 2 * (int64_t)tl; };
        int wt = 0;
        for (int wv = 0; wv < kColLayoutWaves; ++wv) {
            int wt1 = tiles_r;
            if (wv < kColLayoutWaves - 1) {
                const int64_t goal = cost_upto(tiles_r) * (wv + 1) / kColLayoutWaves;
                wt1 = wt;
                while (wt1 < tiles_r && cost_upto(wt1 + 1) <= goal) ++wt1;
            }
            cell.push_back(first_tile + wt); cell.push_back(first_tile + wt1);
            cell.push_back(tile_off[first_tile + wt]); cell.push_back(tile_off[first_tile + wt1]);
            for (int k = 1; k <= 5; ++k) cell.push_back(tile_off[std::min(first_tile + wt + k, first_tile + tiles_r)]);
            cell.push_back(0); cell.push_back(0); cell.push_back(0);
            wt = wt1;
        }
        o = RangeOut();
    }
    const int64_t iters_total = tile_off.back();
    for (int k = 0; k < 6; ++k) tile_off.push_back((int32_t)iters_total);
    for (int k = 0; k < 64; ++k) tile_rows.push_back(-1);
    std::vector<float> tile_dis(tile_rows.size(), 0.f);
    for (size_t k = 0; k < tile_rows.size(); ++k)
        if (tile_rows[k] >= 0) tile_dis[k] = dis_host[tile_rows[k]];
    if (ids.size() / 2 >= ((size_t)1 << 31)) return L;

    GN_LAP("blocked: concatenate");
    L.iters_total = iters_total;
    L.tile_off.swap(tile_off); L.tile_rows.swap(tile_rows); L.cell.swap(cell); L.tile_dis.swap(tile_dis); L.ids.swap(ids);
    L.ok = true;
    return L;
}

// ---- relational weight gradient (rel_grad.hip): dW_r = X^T Q_r, Q_r[s] = sum of the gradient rows of the edges s -> . of
//      relation r, from the (relation, source)-major CSR of the layer's edges ------------------------------------------
constexpr int kRelWaves = 16;              // waves of a k_rel_weight_grad workgroup
constexpr int kRelChunk = 8;               // destination ids of one lane group in one unit
constexpr int kRelRing = 4;                // units a wave has in flight: the units of a wave in a list entry are padded to a multiple
constexpr int kRelSlackUnits = 12;         // readable units behind the last one (ids and x are requested four units ahead, sources eight)
constexpr int kRelCostEntry = 50;          // cost of a list entry in units (the sixteen waves' sums through LDS, four barriers)
constexpr uint32_t kRelNoSource = 0xffffu;

// A UNIT is four CHUNKS of one relation, a chunk up to eight edges of one (relation, source) row: the source and eight
// 16-bit destination ids, padded with `n` (the table's zero row).  A row of more than eight edges is several chunks (the
// sums are linear), so every unit costs the same and nothing in the stream depends on what was loaded before.  A relation
// whose units exceed half a workgroup's fair share is cut into PARTS (contiguous unit ranges); parts are dealt to the
// workgroups longest first; the units of a part go round-robin to the sixteen waves, and every wave's units of all its
// workgroup's parts are contiguous in memory (one stream per wave for the whole launch).
struct RelGradLayout {
    std::vector<uint16_t> src;             // [units][4] source of every chunk (kRelNoSource: none)
    std::vector<uint16_t> ids;             // [units][4][8] destinations
    std::vector<int32_t> entry;            // 4 per list entry: relation, parts of the relation, part index, first scratch slot of the relation
    std::vector<int32_t> wave_cnt;         // [entries][16] units of every wave
    std::vector<int32_t> wg_off;           // [groups + 1] list entries of every workgroup
    std::vector<int32_t> wave_u0;          // [groups][16] first unit of every wave
    int groups = 0, scratch_slots = 0;
    int64_t units = 0;                     // without the slack
    bool ok = false;
};

inline RelGradLayout build_rel_grad_layout(const int32_t* rowptr, const int32_t* col, int64_t n, int64_t R, int groups) {
    RelGradLayout L;
    if (n < 1 || n >= (int64_t)kRelNoSource || R < 1 || groups < 1) return L;
    struct Part { int rel, index, parts, u0, u1, slot0; };
    // chunks of every relation: (source, first edge); units = chunks / 4
    std::vector<int64_t> chunk_off((size_t)R + 1, 0);
    for (int64_t r = 0; r < R; ++r) {
        const int32_t* rp = rowptr + r * n;
        int64_t c = 0;
        for (int64_t s = 0; s < n; ++s) c += gn::ceil_div(rp[s + 1] - rp[s], kRelChunk);
        chunk_off[(size_t)r + 1] = chunk_off[(size_t)r] + c;
    }
    std::vector<int32_t> chunk_src((size_t)chunk_off[(size_t)R]), chunk_first((size_t)chunk_off[(size_t)R]);
    gn::parallel_for(R, 8, [&](int64_t r0, int64_t r1) {
        for (int64_t r = r0; r < r1; ++r) {
            const int32_t* rp = rowptr + r * n;
            size_t o = (size_t)chunk_off[(size_t)r];
            for (int64_t s = 0; s < n; ++s)
                for (int32_t e = rp[s]; e < rp[s + 1]; e += kRelChunk) { chunk_src[o] = (int32_t)s; chunk_first[o] = e; ++o; }
        }
    });
    auto units_of = [&](int64_t r) { return gn::ceil_div(chunk_off[(size_t)r + 1] - chunk_off[(size_t)r], 4); };
    int64_t total = 0;
    for (int64_t r = 0; r < R; ++r) total += units_of(r) + kRelCostEntry;
    const int64_t share = std::max<int64_t>(1, total / groups);
    std::vector<Part> parts;
    int slots = 0;
    for (int64_t r = 0; r < R; ++r) {
        const int64_t nu = units_of(r);
        int np = (int)std::min<int64_t>(gn::ceil_div(2 * nu, share), std::max<int64_t>(1, nu / (2 * kRelWaves)));
        np = std::max(1, std::min(np, 64));
        const int slot0 = np > 1 ? slots : 0;
        for (int p = 0; p < np; ++p) parts.push_back(Part{(int)r, p, np, (int)(nu * p / np), (int)(nu * (p + 1) / np), slot0});
        if (np > 1) slots += np;
    }
    // longest first onto the least loaded workgroup
    std::vector<int> order(parts.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = (int)k;
    auto cost = [&](int k) { return (int64_t)(parts[(size_t)k].u1 - parts[(size_t)k].u0) + kRelCostEntry; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost(a) > cost(b); });
    std::vector<std::vector<int>> mine((size_t)groups);
    {
        std::vector<std::pair<int64_t, int>> heap;           // (-load, -group): the max-heap pops the least loaded, lowest index first
        for (int g = 0; g < groups; ++g) heap.emplace_back(0, -g);
        std::make_heap(heap.begin(), heap.end());
        for (int k : order) {
            std::pop_heap(heap.begin(), heap.end());
            auto top = heap.back();
            mine[(size_t)-top.second].push_back(k);
            top.first -= cost(k);
            heap.back() = top;
            std::push_heap(heap.begin(), heap.end());
        }
    }
    // emit: [workgroup][wave][entry][the wave's units of the entry]
    L.wg_off.assign((size_t)groups + 1, 0);
    for (int g = 0; g < groups; ++g) L.wg_off[(size_t)g + 1] = L.wg_off[(size_t)g] + (int32_t)mine[(size_t)g].size();
    const size_t entries = parts.size();
    L.entry.resize(entries * 4);
    L.wave_cnt.assign(entries * kRelWaves, 0);
    L.wave_u0.assign((size_t)groups * kRelWaves, 0);
    // first unit of every (workgroup, wave): a prefix sum, so that the streams can be written in parallel
    std::vector<int64_t> wave_first((size_t)groups * kRelWaves + 1, 0);
    for (int g = 0; g < groups; ++g) {
        for (size_t k = 0; k < mine[(size_t)g].size(); ++k) {
            const Part& p = parts[(size_t)mine[(size_t)g][k]];
            const size_t e = (size_t)L.wg_off[(size_t)g] + k;
            L.entry[4 * e] = p.rel; L.entry[4 * e + 1] = p.parts; L.entry[4 * e + 2] = p.index; L.entry[4 * e + 3] = p.slot0;
            for (int w = 0; w < kRelWaves; ++w) {
                // (padded with empty units to the depth of the kernel's ring of register sets: its loop body is four units)
                const int cnt = (int)(gn::ceil_div(std::max<int64_t>(0, gn::ceil_div((int64_t)(p.u1 - p.u0) - w, kRelWaves)), kRelRing) * kRelRing);
                L.wave_cnt[e * kRelWaves + w] = cnt;
                wave_first[(size_t)g * kRelWaves + w + 1] += cnt;
            }
        }
    }
    for (size_t k = 1; k < wave_first.size(); ++k) wave_first[k] += wave_first[k - 1];
    const int64_t units = wave_first.back();
    if ((units + kRelSlackUnits) * 4 * kRelChunk >= ((int64_t)1 << 31)) return L;
    L.src.assign((size_t)(units + kRelSlackUnits) * 4, (uint16_t)kRelNoSource);
    L.ids.assign((size_t)(units + kRelSlackUnits) * 4 * kRelChunk, (uint16_t)n);
    for (size_t k = 0; k + 1 < wave_first.size(); ++k) L.wave_u0[k] = (int32_t)wave_first[k];
    gn::parallel_for(groups, 1, [&](int64_t g0, int64_t g1) {
        for (int64_t g = g0; g < g1; ++g)
            for (int w = 0; w < kRelWaves; ++w) {
                int64_t u = wave_first[(size_t)g * kRelWaves + w];
                for (size_t k = 0; k < mine[(size_t)g].size(); ++k) {
                    const Part& p = parts[(size_t)mine[(size_t)g][k]];
                    const int32_t* rp = rowptr + (int64_t)p.rel * n;
                    const int64_t c0 = chunk_off[(size_t)p.rel], c1 = chunk_off[(size_t)p.rel + 1];
                    const size_t e = (size_t)L.wg_off[(size_t)g] + k;
                    const int64_t u_next = u + L.wave_cnt[e * kRelWaves + w];
                    for (int64_t j = p.u0 + w; j < p.u1; j += kRelWaves, ++u)
                        for (int lg = 0; lg < 4; ++lg) {
                            const int64_t ch = c0 + 4 * j + lg;
                            if (ch >= c1) continue;
                            const int32_t s = chunk_src[(size_t)ch], e0 = chunk_first[(size_t)ch], e1 = std::min(rp[s + 1], e0 + kRelChunk);
                            L.src[(size_t)u * 4 + lg] = (uint16_t)s;
                            for (int32_t t = e0; t < e1; ++t) L.ids[((size_t)u * 4 + lg) * kRelChunk + (t - e0)] = (uint16_t)col[t];
                        }
                    u = u_next;
                }
            }
    });
    L.units = units;
    L.groups = groups; L.scratch_slots = slots;
    L.ok = true;
    return L;
}

// ---- LDS bank balance of the decoder gradient's segment reductions (distmult_bwd.hip, k_seg_lds) ----------------------
// A wave works on 64 records at a time; in step S the quad q of the wave reads the two 64-byte table rows of record 4 q + S.
// A ds_read_b128 is served in four groups of sixteen lanes - the quads {0,3,5,6}, {1,2,4,7}, {8,11,13,14}, {9,10,12,15}
// (MI355X_MICROARCH.md, LDS) - one cycle per group when its sixteen 16-byte slots fall on 64 different banks.  A 64-byte row
// covers one QUARTER of the 64 banks - which quarter is the row index mod 4 - so the four rows of a group cost one cycle when
// their indices differ mod 4 and up to four otherwise: with rows in random order 40 % of the LDS cycles of the reductions are
// bank conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE).  The order of the records inside a batch is free (a fixed order
// is a fixed summation order): for a STATIC list it is chosen once so that the four records of every (step, lane group) have
// four different residues in BOTH tables wherever the batch allows it.
// cls[i] = (first row of record i) mod 4 | ((second row) mod 4) << 2;  order[p] = the record that goes to position p.
constexpr int kLdsGroupQuads[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};

// LDS cycles of the row reads of a batch in the given order (1 per conflict-free group and table, up to 4): what the order buys
inline int batch64_access_cycles(const uint8_t* cls, const int* order) {
    int cycles = 0;
    for (int g = 0; g < 16; ++g)
        for (int table = 0; table < 2; ++table) {
            int hits[4] = {0, 0, 0, 0};
            for (int a = 0; a < 4; ++a) ++hits[(cls[order[4 * kLdsGroupQuads[g >> 2][a] + (g & 3)]] >> (2 * table)) & 3];
            cycles += std::max(std::max(hits[0], hits[1]), std::max(hits[2], hits[3]));
        }
    return cycles;
}

inline void balance_batch64_greedy(const uint8_t* cls, int* order) {
    std::vector<int> of[16];
    for (int i = 63; i >= 0; --i) of[cls[i] & 15].push_back(i);        // (taken from the back: in input order)
    static const int perms[24][4] = {{0,1,2,3},{0,1,3,2},{0,2,1,3},{0,2,3,1},{0,3,1,2},{0,3,2,1},{1,0,2,3},{1,0,3,2},{1,2,0,3},{1,2,3,0},
                                     {1,3,0,2},{1,3,2,0},{2,0,1,3},{2,0,3,1},{2,1,0,3},{2,1,3,0},{2,3,0,1},{2,3,1,0},{3,0,1,2},{3,0,2,1},
                                     {3,1,0,2},{3,1,2,0},{3,2,0,1},{3,2,1,0}};
    int slot[16][4];                                                    // the records of the sixteen groups, -1: still to fill
    for (int g = 0; g < 16; ++g) {
        // the transversal (first residue a -> second residue perm[a]) whose scarcest class is the fullest: keeps the classes level
        int best = -1, best_min = -1, best_sum = -1;
        for (int k = 0; k < 24; ++k) {
            int mn = 1 << 30, sum = 0, have = 0;
            for (int a = 0; a < 4; ++a) {
                const int c = (int)of[a | (perms[k][a] << 2)].size();
                have += c > 0;
                if (c > 0) mn = std::min(mn, c);
                sum += c;
            }
            const int key = have * 1000 + (have ? mn : 0);
            if (key > best_min || (key == best_min && sum > best_sum)) { best = k; best_min = key; best_sum = sum; }
        }
        for (int a = 0; a < 4; ++a) {
            std::vector<int>& l = of[a | (perms[best][a] << 2)];
            slot[g][a] = l.empty() ? -1 : l.back();
            if (!l.empty()) l.pop_back();
        }
    }
    std::vector<int> rest;
    for (int c = 0; c < 16; ++c)
        for (size_t i = of[c].size(); i-- > 0;) rest.push_back(of[c][i]);
    size_t r = 0;
    for (int g = 0; g < 16; ++g) {                                      // group g = (step S, lane group k) = (g % 4, g / 4)
        const int S = g & 3, k = g >> 2;
        for (int a = 0; a < 4; ++a) {
            if (slot[g][a] < 0) slot[g][a] = rest[r++];
            order[4 * kLdsGroupQuads[k][a] + S] = slot[g][a];
        }
    }
}

// (the greedy order, or the input order where that is no worse: rows that share few residues)
inline void balance_batch64(const uint8_t* cls, int* order) {
    int ident[64];
    for (int i = 0; i < 64; ++i) ident[i] = i;
    balance_batch64_greedy(cls, order);
    if (batch64_access_cycles(cls, order) >= batch64_access_cycles(cls, ident))
        for (int i = 0; i < 64; ++i) order[i] = i;
}

}  // namespace gn_layout
