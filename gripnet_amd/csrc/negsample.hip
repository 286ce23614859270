// Device-side typed negative sampling.
//
// The reference draws the negative edges of every epoch on the host (gripnet/utils.py:98-119): per
// relation block, E_r linear pair ids uniformly from [0, n^2) with numpy, np.isin against that block's
// positive pairs, redraw the hits until none is left, copy back to the device; one Python iteration per
// relation (964 on PoSE) and a device -> host -> device round trip per epoch.  Here the positive pairs
// are sorted once per graph into (relation, u*n+v) keys (a "sampler", like a plan), and one kernel
// draws every negative: thread e finds its relation block, draws from a counter-based generator
// keyed by (seed, e, attempt), rejects a draw that is a positive of the same block by binary search,
// and writes (u, v) as int64.  Same distribution as the reference (uniform over the non-positive
// pairs of the block, with replacement); not the same random stream (numpy's global RNG cannot be
// replayed on the device, and the reference's own stream changes with every call).
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <vector>

struct gn_negative_sampler {
    int64_t num_edges = 0, num_nodes = 0, num_relations = 0;
    gn::DevBuf<uint64_t> keys;      // [E] sorted (relation << 40) | (u * n + v)
    gn::DevBuf<int64_t> starts;     // [R + 1] block starts
    // the narrow encoding (n < 2^16, R < 2^16, E < 2^31: every GripNet graph): the pair ids of the sorted keys as 32-bit
    // words and the relation of every edge position, so that a draw costs one 2-byte load, two block bounds and a binary
    // search over 4-byte words of a block that sits in L2 (the edges of a wave share their relation)
    gn::DevBuf<uint32_t> keys32;    // [E] u * n + v in the order of `keys`
    gn::DevBuf<uint16_t> rel16;     // [E] relation of edge position e
    int narrow = 0;
    // and, while R * n^2 bits stay below kBitmapBytes = 128 MB (PoSE: 964 x 645^2 bits = 50 MB), one bit per (relation, pair): a
    // draw is tested with ONE 4-byte load - a binary search costs ~11 dependent loads whose last steps touch a
    // different cache line in every lane
    gn::DevBuf<uint32_t> bitmap;    // [R][words]
    int64_t words = 0;              // 32-bit words per relation (a multiple of four), 0: no bitmap
    // and with the bitmap, the draw by TASKS: a workgroup takes a slice of ONE relation's edge positions, four positions per
    // thread and trip (four tests in flight); a small relation's sorted pair ids are staged in LDS first and the test is a
    // binary search there, a large relation is tested against its row of the bitmap
    gn::DevBuf<int32_t> tasks;      // 2 x int4: (relation, first edge position, end, 0 = ids | 1 = bitmap), (first id, ids, -, -)
    int64_t num_tasks = 0;
    size_t task_lds = 0;
};

namespace {

constexpr int kMaxAttempts = 4096;

__device__ __forceinline__ uint64_t mix64(uint64_t x) {     // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ int relation_of(const int64_t* __restrict__ starts, int R, int64_t e) {
    int a = 0, b = R;                                       // last r with starts[r] <= e
    while (b - a > 1) {
        int mid = (a + b) >> 1;
        if (starts[mid] <= e) a = mid; else b = mid;
    }
    return a;
}

__global__ void k_pair_keys(const int64_t* __restrict__ u, const int64_t* __restrict__ v,
                            const int64_t* __restrict__ starts, int R, int64_t E, int64_t n,
                            uint64_t* __restrict__ keys, int32_t* __restrict__ err) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t uu = u[e], vv = v[e];
        if ((uint64_t)uu >= (uint64_t)n || (uint64_t)vv >= (uint64_t)n) { atomicOr(err, 1); keys[e] = ~0ull; continue; }
        keys[e] = ((uint64_t)relation_of(starts, R, e) << 40) | (uint64_t)(uu * n + vv);
    }
}

__global__ void k_sample_negatives(const uint64_t* __restrict__ keys, const int64_t* __restrict__ starts, int R,
                                   int64_t E, int64_t n, uint64_t seed, const uint64_t* __restrict__ seed_step, int64_t* __restrict__ out_u,
                                   int64_t* __restrict__ out_v, uint32_t* __restrict__ packed, int32_t* __restrict__ err) {
    const uint64_t n2 = (uint64_t)n * (uint64_t)n;
    if (seed_step) seed += *seed_step;                       // draw number `step` of a captured loop = the draw of seed + step
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = relation_of(starts, R, e);
        const int64_t lo0 = starts[r], hi0 = starts[r + 1];
        uint64_t lin = 0;
        bool found = false;
        for (int k = 0; k < kMaxAttempts && !found; ++k) {
            lin = mix64(mix64(seed ^ (uint64_t)e * 0xD6E8FEB86659FD93ull) + (uint64_t)k) % n2;
            const uint64_t key = ((uint64_t)r << 40) | lin;
            int64_t lo = lo0, hi = hi0;                     // is `key` one of this block's positives?
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (keys[mid] < key) lo = mid + 1; else hi = mid;
            }
            found = !(lo < hi0 && keys[lo] == key);
        }
        if (!found && err) atomicOr(err, 2);                // the block's positives (nearly) cover all n^2 pairs
        const uint64_t uu = lin / (uint64_t)n, vv = lin % (uint64_t)n;
        out_u[e] = (int64_t)uu;
        out_v[e] = (int64_t)vv;
        if (packed) packed[e] = (uint32_t)uu | ((uint32_t)vv << 16);
    }
}

__global__ void k_narrow_keys(const uint64_t* __restrict__ keys, const int64_t* __restrict__ starts, int R, int64_t E,
                              uint32_t* __restrict__ keys32, uint16_t* __restrict__ rel16) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        keys32[e] = (uint32_t)(keys[e] & 0xFFFFFFFFull);
        rel16[e] = (uint16_t)relation_of(starts, R, e);
    }
}

// The narrow sampler: no 64-bit division (u and v are two multiply-high draws of one 64-bit hash: uniform over the n^2
// pairs), no search for the relation, 32-bit compares.  Same distribution as k_sample_negatives, another stream.
__global__ __launch_bounds__(256) void k_sample_negatives_narrow(const uint32_t* __restrict__ keys32, const uint16_t* __restrict__ rel16,
                                                                 const int64_t* __restrict__ starts, int64_t E, uint32_t n, uint64_t seed, const uint64_t* __restrict__ seed_step,
                                                                 int64_t* __restrict__ out_u, int64_t* __restrict__ out_v,
                                                                 uint32_t* __restrict__ packed, int32_t* __restrict__ err) {
    if (seed_step) seed += *seed_step;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int r = rel16[e];
        const int lo0 = (int)starts[r], hi0 = (int)starts[r + 1];
        const uint64_t base = mix64(seed ^ (uint64_t)e * 0xD6E8FEB86659FD93ull);
        uint32_t uu = 0, vv = 0;
        bool found = false;
        for (int k = 0; k < kMaxAttempts && !found; ++k) {
            const uint64_t h = mix64(base + (uint64_t)k);
            uu = __umulhi((uint32_t)h, n);
            vv = __umulhi((uint32_t)(h >> 32), n);
            const uint32_t key = uu * n + vv;
            int lo = lo0, hi = hi0;                           // is `key` one of this block's positives?
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (keys32[mid] < key) lo = mid + 1; else hi = mid;
            }
            found = !(lo < hi0 && keys32[lo] == key);
        }
        if (!found && err) atomicOr(err, 2);                  // the block's positives (nearly) cover all n^2 pairs
        out_u[e] = (int64_t)uu;
        out_v[e] = (int64_t)vv;
        if (packed) packed[e] = uu | (vv << 16);
    }
}

constexpr int64_t kBitmapBytes = 128ll << 20;

__global__ void k_fill_bitmap(const uint32_t* __restrict__ keys32, const uint16_t* __restrict__ rel16, int64_t E, int64_t words,
                              uint32_t* __restrict__ bitmap) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t key = keys32[e];
        if ((int64_t)(key >> 5) >= words) continue;           // an edge with a node id out of range (the create call fails after this launch)
        atomicOr(bitmap + (int64_t)rel16[e] * words + (key >> 5), 1u << (key & 31));
    }
}

__global__ __launch_bounds__(256) void k_sample_negatives_bitmap(const uint32_t* __restrict__ bitmap, int64_t words,
                                                                 const uint16_t* __restrict__ rel16, int64_t E, uint32_t n, uint64_t seed, const uint64_t* __restrict__ seed_step,
                                                                 int64_t* __restrict__ out_u, int64_t* __restrict__ out_v,
                                                                 uint32_t* __restrict__ packed, int32_t* __restrict__ err) {
    if (seed_step) seed += *seed_step;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t* __restrict__ bits = bitmap + (int64_t)rel16[e] * words;
        const uint64_t base = mix64(seed ^ (uint64_t)e * 0xD6E8FEB86659FD93ull);   // the stream of the narrow sampler
        uint32_t uu = 0, vv = 0;
        bool found = false;
        for (int k = 0; k < kMaxAttempts && !found; ++k) {
            const uint64_t h = mix64(base + (uint64_t)k);
            uu = __umulhi((uint32_t)h, n);
            vv = __umulhi((uint32_t)(h >> 32), n);
            const uint32_t key = uu * n + vv;
            found = !((bits[key >> 5] >> (key & 31)) & 1u);
        }
        if (!found && err) atomicOr(err, 2);                  // the block's positives (nearly) cover all n^2 pairs
        out_u[e] = (int64_t)uu;
        out_v[e] = (int64_t)vv;
        if (packed) packed[e] = uu | (vv << 16);
    }
}

// GN_SAMPLER_TASKS=0 keeps the one-load-per-draw bitmap kernel (the tests compare the two: same draws)
bool sampler_tasks_disabled() {
    const char* e = getenv("GN_SAMPLER_TASKS");
    return e && e[0] == '0';
}

// Measured on pose0-syn (964 relations of 534 to 130,332 positions, 2 M draws; tools/probes/sampler_probe.py): 28.9 us with one
// draw per thread and trip against the bitmap, 27.4 us with four in flight, 22.2 us with the ids of the relations of up to 1,024
// positions staged (half of the relations, a fifth of the positions); staging longer id lists costs more than it returns
// (2,048: 24.2 us, 8,192: 32.0 us), and so does staging a relation's 52 KB of bits per slice (35 us).  9.5 us without any test.
constexpr int kTaskSliceIds = 2048, kTaskSliceBits = 2048, kTaskMaxIds = 1024;

// Same draws as k_sample_negatives_bitmap / _narrow (same hashes, same order of attempts), tested against a staged table.
__global__ __launch_bounds__(256) void k_sample_negatives_tasks(const int4* __restrict__ tasks, const uint32_t* __restrict__ keys32,
                                                                const int64_t* __restrict__ starts, const uint32_t* __restrict__ bitmap,
                                                                int64_t words, uint32_t n, uint64_t seed, uint64_t* seed_step,
                                                                unsigned int* __restrict__ arrived,
                                                                int64_t* __restrict__ out_u, int64_t* __restrict__ out_v,
                                                                uint32_t* __restrict__ packed, int32_t* __restrict__ err) {
    extern __shared__ uint32_t staged[];
    if (seed_step) seed += *seed_step;
    const int4 task = tasks[2 * blockIdx.x], block = tasks[2 * blockIdx.x + 1];   // (relation, first position, end, kind), (first id, ids, -, -)
    const int tid = threadIdx.x;
    const bool bits = task.w != 0;
    const int len = bits ? 0 : block.y, shift = bits ? 0 : (block.x & 3);
    if (!bits) {
        // the relation's sorted ids from the 16-byte boundary below their first word, 16 bytes per lane, eight loads in flight
        const uint4* src = reinterpret_cast<const uint4*>(keys32 + (block.x - shift));     // (keys32 is 256-byte aligned)
        const int quads = (len + shift + 3) >> 2;                                          // may read up to 3 words past the block: inside the array's padding
        for (int i0 = tid; i0 < quads; i0 += 256 * 8) {
            uint4 q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = src[min(i0 + 256 * j, quads - 1)];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (i0 + 256 * j < quads) reinterpret_cast<uint4*>(staged)[i0 + 256 * j] = q[j];
        }
        __syncthreads();
    }
    const uint32_t* __restrict__ table = staged + shift;
    const uint32_t* __restrict__ row = bitmap + (int64_t)task.x * words;                  // a large relation: its bits where they are
    int top = 1;                                              // smallest power of two > len (uniform)
    while (top <= len) top <<= 1;
    auto is_positive = [&](uint32_t key) -> bool {
        if (bits) return ((row[key >> 5] >> (key & 31)) & 1u) != 0;
        int lo = 0;                                           // number of ids below `key`
        for (int step = top >> 1; step > 0; step >>= 1) {
            const int idx = lo + step;
            if (idx <= len && table[idx - 1] < key) lo = idx;
        }
        return lo < len && table[lo] == key;
    };
    // four positions per thread and trip: their first attempts are tested together (four independent chains of LDS reads);
    // a position whose first attempt hit a positive (0.6 % on PoSE) goes on alone
    constexpr int U = 4;
    for (int e0 = task.y + tid; e0 < task.z; e0 += 256 * U) {
        uint64_t base[U];
        uint32_t uu[U], vv[U];
        bool hit[U];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = e0 + 256 * j;
            base[j] = mix64(seed ^ (uint64_t)e * 0xD6E8FEB86659FD93ull);
            const uint64_t h = mix64(base[j]);
            uu[j] = __umulhi((uint32_t)h, n);
            vv[j] = __umulhi((uint32_t)(h >> 32), n);
        }
        if (bits) {
#pragma unroll
            for (int j = 0; j < U; ++j) { const uint32_t key = uu[j] * n + vv[j]; hit[j] = ((row[key >> 5] >> (key & 31)) & 1u) != 0; }
        } else {
            int lo[U];
            uint32_t key[U];
#pragma unroll
            for (int j = 0; j < U; ++j) { lo[j] = 0; key[j] = uu[j] * n + vv[j]; }
            for (int step = top >> 1; step > 0; step >>= 1) {
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int idx = lo[j] + step;
                    if (idx <= len && table[idx - 1] < key[j]) lo[j] = idx;
                }
            }
#pragma unroll
            for (int j = 0; j < U; ++j) hit[j] = lo[j] < len && table[lo[j]] == key[j];
        }

#pragma unroll
        for (int j = 0; j < U; ++j) {
            const int e = e0 + 256 * j;
            if (e >= task.z) continue;
            bool found = !hit[j];
            for (int k = 1; k < kMaxAttempts && !found; ++k) {
                const uint64_t h = mix64(base[j] + (uint64_t)k);
                uu[j] = __umulhi((uint32_t)h, n);
                vv[j] = __umulhi((uint32_t)(h >> 32), n);
                found = !is_positive(uu[j] * n + vv[j]);
            }
            if (!found && err) atomicOr(err, 2);
            out_u[e] = (int64_t)uu[j];
            out_v[e] = (int64_t)vv[j];
            if (packed) packed[e] = uu[j] | (vv[j] << 16);
        }
    }
    // the step counter moves when every workgroup has read it (each reads it first thing): the last one to arrive writes it
    if (seed_step) {
        __syncthreads();
        if (tid == 0) {
            const unsigned int seen = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (seen == gridDim.x - 1) {
                *seed_step += 1;
                __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next draw (stream-ordered)
            }
        }
    }
}

__global__ void k_advance_step(uint64_t* step) { *step += 1; }

gn_status launch_sample(const gn_negative_sampler* s, uint64_t seed, int64_t* out_u, int64_t* out_v, uint32_t* packed,
                        int32_t* error_flag, hipStream_t st, uint64_t* seed_step = nullptr) {
    if (s->num_tasks > 0)
        k_sample_negatives_tasks<<<(unsigned)s->num_tasks, 256, s->task_lds, st>>>(
            reinterpret_cast<const int4*>(s->tasks.p), s->keys32.p, s->starts.p, s->bitmap.p, s->words, (uint32_t)s->num_nodes, seed, seed_step,
            reinterpret_cast<unsigned int*>(s->tasks.p + 8 * s->num_tasks), out_u, out_v, packed, error_flag);
    else if (s->words > 0)
        k_sample_negatives_bitmap<<<gn::stream_grid(s->num_edges, 256), 256, 0, st>>>(
            s->bitmap.p, s->words, s->rel16.p, s->num_edges, (uint32_t)s->num_nodes, seed, seed_step, out_u, out_v, packed, error_flag);
    else if (s->narrow)
        k_sample_negatives_narrow<<<gn::stream_grid(s->num_edges, 256), 256, 0, st>>>(
            s->keys32.p, s->rel16.p, s->starts.p, s->num_edges, (uint32_t)s->num_nodes, seed, seed_step, out_u, out_v, packed, error_flag);
    else
        k_sample_negatives<<<gn::stream_grid(s->num_edges, 256), 256, 0, st>>>(
            s->keys.p, s->starts.p, (int)s->num_relations, s->num_edges, s->num_nodes, seed, seed_step, out_u, out_v, packed, error_flag);
    GN_LAUNCH_CHECK();
    if (seed_step && s->num_tasks == 0) {                     // (stream order: every workgroup of the draw has read it; the task kernel moves it itself)
        k_advance_step<<<1, 1, 0, st>>>(seed_step);
        GN_LAUNCH_CHECK();
    }
    return GN_OK;
}

}  // namespace

extern "C" {

gn_status gn_negative_sampler_create(const int64_t* u, const int64_t* v, const int64_t* range_list_host,
                                     int64_t R, int64_t E, int64_t N, void* stream, gn_negative_sampler** out) {
    GN_REQUIRE(out != nullptr, "sampler output pointer is null");
    *out = nullptr;
    GN_REQUIRE(R >= 1 && E >= 0 && N >= 1, "bad size (R=%lld, E=%lld, N=%lld)", (long long)R, (long long)E, (long long)N);
    GN_REQUIRE(E == 0 || (u && v), "edge pointers are null");
    GN_REQUIRE(range_list_host != nullptr, "range_list is null");
    if (N >= (1ll << 20) || R >= (1ll << 23))
        return gn::fail(GN_ERR_UNSUPPORTED, "sampler keys hold 2^20 nodes and 2^23 relations at most");
    std::vector<int64_t> starts(R + 1);
    int64_t cursor = 0;
    for (int64_t r = 0; r < R; ++r) {
        if (range_list_host[2 * r] != cursor || range_list_host[2 * r + 1] < cursor)
            return gn::fail(GN_ERR_INVALID_ARG, "range_list must tile [0,E) in relation order (row %lld)", (long long)r);
        starts[r] = cursor;
        cursor = range_list_host[2 * r + 1];
    }
    if (cursor != E) return gn::fail(GN_ERR_INVALID_ARG, "range_list covers %lld edges but edge_index has %lld",
                                     (long long)cursor, (long long)E);
    starts[R] = E;
    hipStream_t st = gn::as_stream(stream);
    gn_negative_sampler* s = new gn_negative_sampler();
    s->num_edges = E; s->num_nodes = N; s->num_relations = R;
    auto bail = [&](gn_status code) { gn_negative_sampler_destroy(s); return code; };
#define GN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(gn::fail(GN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
    GN_TRY(s->keys.alloc(E));
    GN_TRY(s->starts.alloc(R + 1));
    GN_TRY(hipMemcpyAsync(s->starts.p, starts.data(), (R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    uint64_t* raw = nullptr;
    int32_t* err = nullptr;
    void* scratch = nullptr;
    int32_t bad = 0;
    if (E > 0) {
        GN_TRY(hipMalloc(reinterpret_cast<void**>(&raw), E * sizeof(uint64_t)));
        hipError_t e2 = hipMalloc(reinterpret_cast<void**>(&err), sizeof(int32_t));
        if (e2 != hipSuccess) { (void)hipFree(raw); return bail(gn::fail(GN_ERR_HIP, "hipMalloc failed")); }
        (void)hipMemsetAsync(err, 0, sizeof(int32_t), st);
        k_pair_keys<<<gn::stream_grid(E, 256), 256, 0, st>>>(u, v, s->starts.p, (int)R, E, N, raw, err);
        size_t bytes = 0;
        hipError_t e3 = rocprim::radix_sort_keys(nullptr, bytes, raw, s->keys.p, (size_t)E, 0, 64, st);
        if (e3 == hipSuccess) e3 = hipMalloc(&scratch, bytes ? bytes : 1);
        if (e3 == hipSuccess) e3 = rocprim::radix_sort_keys(scratch, bytes, raw, s->keys.p, (size_t)E, 0, 64, st);
        if (e3 == hipSuccess && N < (1ll << 16) && R < (1ll << 16) && E < (1ll << 31) && !gn::fast_paths_disabled()) {
            e3 = s->keys32.alloc(E + 8);                     // (the task kernel stages whole 16-byte words)
            if (e3 == hipSuccess) e3 = s->rel16.alloc(E);
            if (e3 == hipSuccess) {
                k_narrow_keys<<<gn::stream_grid(E, 256), 256, 0, st>>>(s->keys.p, s->starts.p, (int)R, E, s->keys32.p, s->rel16.p);
                e3 = hipGetLastError();
                s->narrow = 1;
            }
            const int64_t words = ((N * N + 31) / 32 + 3) & ~(int64_t)3;
            if (e3 == hipSuccess && R * words * 4 <= kBitmapBytes) {
                e3 = s->bitmap.alloc(R * words);
                if (e3 == hipSuccess) e3 = hipMemsetAsync(s->bitmap.p, 0, (size_t)(R * words) * sizeof(uint32_t), st);
                if (e3 == hipSuccess) {
                    k_fill_bitmap<<<gn::stream_grid(E, 256), 256, 0, st>>>(s->keys32.p, s->rel16.p, E, words, s->bitmap.p);
                    e3 = hipGetLastError();
                    s->words = words;
                }
                if (e3 == hipSuccess && !sampler_tasks_disabled()) {
                    std::vector<int32_t> tasks;
                    int64_t most = 0;
                    for (int pass = 1; pass >= 0; --pass)               // the large relations first
                        for (int64_t r = 0; r < R; ++r) {
                            const int64_t len = starts[r + 1] - starts[r];
                            const int kind = len > kTaskMaxIds ? 1 : 0;
                            if (kind != pass) continue;
                            if (!kind) most = std::max(most, len);
                            const int64_t slice = kind ? kTaskSliceBits : kTaskSliceIds;
                            for (int64_t a = starts[r]; a < starts[r + 1]; a += slice) {
                                const int32_t d[8] = {(int32_t)r, (int32_t)a, (int32_t)std::min(starts[r + 1], a + slice), kind,
                                                      (int32_t)starts[r], (int32_t)len, 0, 0};
                                tasks.insert(tasks.end(), d, d + 8);
                            }
                        }
                    if (!tasks.empty()) {
                        const size_t described = tasks.size();
                        tasks.push_back(0);                                   // the arrival counter of the stepped draw
                        e3 = s->tasks.alloc(tasks.size());
                        if (e3 == hipSuccess) e3 = hipMemcpyAsync(s->tasks.p, tasks.data(), tasks.size() * sizeof(int32_t), hipMemcpyHostToDevice, st);
                        if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);   // `tasks` leaves scope
                        s->num_tasks = (int64_t)(described / 8);
                        s->task_lds = (size_t)(most + 8) * 4;                  // (up to three words of alignment in front)
                    }
                }
            }
        }
        if (e3 == hipSuccess) e3 = hipMemcpyAsync(&bad, err, sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e3 == hipSuccess) e3 = hipStreamSynchronize(st);
        (void)hipFree(raw); (void)hipFree(err); if (scratch) (void)hipFree(scratch);
        if (e3 != hipSuccess) return bail(gn::fail(GN_ERR_HIP, "sampler construction failed: %s", hipGetErrorString(e3)));
    } else {
        GN_TRY(hipStreamSynchronize(st));
    }
#undef GN_TRY
    if (bad) return bail(gn::fail(GN_ERR_INDEX_RANGE, "edge_index holds a node id outside [0,%lld)", (long long)N));
    *out = s;
    return GN_OK;
}

void gn_negative_sampler_destroy(gn_negative_sampler* s) {
    if (!s) return;
    s->keys.release();
    s->starts.release();
    s->keys32.release();
    s->rel16.release();
    s->bitmap.release();
    s->tasks.release();
    delete s;
}

gn_status gn_negative_sampler_sample(const gn_negative_sampler* s, uint64_t seed, int64_t* out_u, int64_t* out_v,
                                     int32_t* error_flag, void* stream) {
    GN_REQUIRE(s != nullptr, "sampler is null");
    if (s->num_edges == 0) return GN_OK;
    GN_REQUIRE(out_u && out_v, "output pointers are null");
    return launch_sample(s, seed, out_u, out_v, nullptr, error_flag, gn::as_stream(stream));
}

gn_status gn_negative_sampler_sample_packed(const gn_negative_sampler* s, uint64_t seed, int64_t* out_u, int64_t* out_v,
                                            uint32_t* packed_uv, int32_t* error_flag, void* stream) {
    GN_REQUIRE(s != nullptr, "sampler is null");
    if (s->num_edges == 0) return GN_OK;
    GN_REQUIRE(out_u && out_v && packed_uv, "output pointers are null");
    if (s->num_nodes > 65535) return gn::fail(GN_ERR_UNSUPPORTED, "packed pairs hold node ids of 16 bits");
    return launch_sample(s, seed, out_u, out_v, packed_uv, error_flag, gn::as_stream(stream));
}

gn_status gn_negative_sampler_sample_stepped(const gn_negative_sampler* s, uint64_t seed, uint64_t* step, int64_t* out_u, int64_t* out_v,
                                             uint32_t* packed_uv, int32_t* error_flag, void* stream) {
    GN_REQUIRE(s != nullptr, "sampler is null");
    GN_REQUIRE(step != nullptr && (reinterpret_cast<uintptr_t>(step) & 7) == 0, "the step counter is null or not 8-byte aligned");
    if (s->num_edges == 0) return GN_OK;
    GN_REQUIRE(out_u && out_v, "output pointers are null");
    if (packed_uv && s->num_nodes > 65535) return gn::fail(GN_ERR_UNSUPPORTED, "packed pairs hold node ids of 16 bits");
    return launch_sample(s, seed, out_u, out_v, packed_uv, error_flag, gn::as_stream(stream), step);
}

}  // extern "C"
