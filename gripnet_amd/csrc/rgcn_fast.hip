// LDS-resident multi-relational layer for small supervertices (the drug supervertex of PoSE:
// n_d = 645 nodes, 48 -> 32 features, ~10^3 relations, millions of edges).
//
//   out[i] = (sum_{e: dst=i} x[src_e] W_{r(e)}) / max(1, indeg_i) + x[i] root (+ bias)
//
// With a few hundred nodes the whole [N, out] accumulator (82.5 KB) fits in a CU's LDS, and so
// does a source tile of H_r = X W_r.  The type-sorted edge list is cut at plan time into work
// items (relation r, source tile t, <= kChunk edges); one persistent workgroup per CU walks its
// items:  (1) H tile = X[tile] @ W_r on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32),
// written to LDS;  (2) every edge of the item adds one 128-byte LDS row H[src] into the LDS
// accumulator row of its destination.  Inside an item the edges are bucketed by
// (dst mod #slots): each 8-lane slot of each wave owns a disjoint set of destinations, so the
// accumulation needs no atomics and runs in a fixed order (bitwise reproducible).
// The per-workgroup accumulators leave as slabs and a small second kernel sums them in a fixed
// order and applies mean / root / bias / activation (or emits the raw partial sum for the
// multi-GPU all-reduce).
//
// HBM traffic per edge is the 4-byte packed (dst, src) pair; W (R x 6 KB) and X (124 KB) are
// L2-resident.  The matrix-core time of (1) is ~2 N_tile x in x out flops per item and is the
// larger cost below ~10^4 edges per relation.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 1024;
constexpr int kWaves = kThreads / 64;
constexpr int kGroups = 256;               // persistent workgroups = CUs of an MI355X
constexpr int kChunk = 6144;               // edges per work item (bounds the load imbalance)
constexpr int kItemOverhead = 1536;        // H-tile cost in edge equivalents (LPT balancing)
constexpr size_t kLdsBudget = 156 * 1024;

struct FastGeom {
    int tiles = 0;      // source tiles
    int ts = 0;         // rows per source tile
    int ts_pad = 0;     // rounded up to 16 (MFMA row tile)
    size_t lds_bytes = 0;
};

FastGeom geometry(int64_t n, int64_t fout) {
    FastGeom g;
    for (int t = 1; t <= 8; ++t) {
        const int64_t ts = gn::ceil_div(n, t), ts_pad = gn::ceil_div(ts, 16) * 16;
        const size_t bytes = (size_t)(n + ts_pad) * fout * sizeof(float);
        if (bytes <= kLdsBudget) {
            g.tiles = t; g.ts = (int)ts; g.ts_pad = (int)ts_pad; g.lds_bytes = bytes;
            return g;
        }
    }
    return g;
}

struct FastArgs {
    const float* x; int64_t ld_x; int n;
    const float* w;                 // [R, FIN*FOUT]
    const uint32_t* packed;         // (dst << 16) | (src - tile * ts), bucket-sorted inside each item
    const int32_t* bucket_off;      // [n_items * NB + 1]
    const int32_t* item_rel; const int32_t* item_tile;
    const int32_t* wg_begin;        // [groups + 1] ranges into wg_items
    const int32_t* wg_items;
    int ts, ts_pad;
    float* slabs;                   // [groups, n, FOUT]
};

template <int FIN, int FOUT>
__global__ __launch_bounds__(kThreads) void k_rgcn_lds(FastArgs a) {
    constexpr int LPR = FOUT / 4;            // lanes per feature row (float4 each)
    constexpr int SLOTS = 64 / LPR;          // edges in flight per wave
    constexpr int NB = kWaves * SLOTS;       // destination buckets per item
    constexpr int KC = FIN / 16;             // 16-deep K chunks
    constexpr int CT = FOUT / 16;            // 16-column tiles
    extern __shared__ f32x4 lds4[];
    f32x4* acc4 = lds4;                                  // [n][LPR]
    f32x4* h4 = lds4 + (size_t)a.n * LPR;                // [ts_pad][LPR]
    float* hf = reinterpret_cast<float*>(h4);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int slot = lane / LPR, j = lane % LPR;

    for (int i = tid; i < a.n * LPR; i += kThreads) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int wi = a.wg_begin[blockIdx.x]; wi < a.wg_begin[blockIdx.x + 1]; ++wi) {
        const int item = a.wg_items[wi];
        const int rel = a.item_rel[item], tile = a.item_tile[item];
        const int row_base = tile * a.ts;
        const int rows = min(a.ts, a.n - row_base);
        // ---- (1) H tile = X[tile rows] @ W_rel ----------------------------------------------
        // B fragments: lane (c16, q) holds W[k = 16 kc + 4 q + jj][16 ct + c16]  (k permuted like A)
        const float* __restrict__ wr = a.w + (size_t)rel * (FIN * FOUT);
        float b[KC][4][CT];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[kc][jj][ct] = wr[(16 * kc + 4 * q + jj) * FOUT + 16 * ct + c16];
        __syncthreads();          // previous item's gather is done with the H tile (and acc is zeroed)
        for (int rt = wave; rt * 16 < rows; rt += kWaves) {
            const int lrow = rt * 16 + c16;
            const bool valid = lrow < rows;
            const float* __restrict__ xr = a.x + (int64_t)(row_base + (valid ? lrow : 0)) * a.ld_x + 4 * q;
            f32x4 d[CT];
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) d[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                f32x4 av = *reinterpret_cast<const f32x4*>(xr + 16 * kc);
                if (!valid) av = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        d[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[jj], b[kc][jj][ct], d[ct], 0, 0, 0);
            }
            // D fragment: lane (c16, q) holds H[rt*16 + 4q + i][16 ct + c16], i = 0..3
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int i = 0; i < 4; ++i) hf[(rt * 16 + 4 * q + i) * FOUT + 16 * ct + c16] = d[ct][i];
        }
        __syncthreads();
        // ---- (2) gather-accumulate: this slot owns the destinations of bucket (wave, slot) -------
        const int bidx = item * NB + wave * SLOTS + slot;
        int p = a.bucket_off[bidx];
        const int pend = a.bucket_off[bidx + 1];
        int cur = -1;
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
        uint32_t pk = p < pend ? a.packed[p] : 0u;
        while (__any(p < pend)) {
            const bool live = p < pend;
            const uint32_t now = pk;
            ++p;
            pk = p < pend ? a.packed[p] : 0u;                 // next edge in flight
            if (live) {
                const int dst = (int)(now >> 16), src = (int)(now & 0xffffu);
                const f32x4 v = h4[src * LPR + j];
                if (dst != cur) {
                    if (cur >= 0) acc4[cur * LPR + j] += sum;  // exclusive owner: plain read-modify-write
                    cur = dst;
                    sum = v;
                } else {
                    sum += v;
                }
            }
        }
        if (cur >= 0) acc4[cur * LPR + j] += sum;
    }
    __syncthreads();
    f32x4* slab = reinterpret_cast<f32x4*>(a.slabs) + (size_t)blockIdx.x * a.n * LPR;
    for (int i = tid; i < a.n * LPR; i += kThreads) slab[i] = acc4[i];
}

// out[i, c] = act( (sum_g slab[g][i][c]) / max(1, indeg) + x[i] . root[:, c] + bias[c] )   (partial: raw sum)
struct FinArgs {
    const float* slabs; int groups; int n; int fout;
    const float* indeg; const float* x; int64_t ld_x; int fin; const float* root; const float* bias;
    int relu; int partial; float* out; int64_t ld_out;
};

__global__ __launch_bounds__(256) void k_rgcn_slab_finalize(FinArgs a) {
    // 256 threads = 32 consecutive output elements x 8 slab groups
    __shared__ float part[8][33];
    const int e_local = threadIdx.x & 31, gg = threadIdx.x >> 5;
    const int64_t total = (int64_t)a.n * a.fout;
    const int64_t elem = (int64_t)blockIdx.x * 32 + e_local;
    float s = 0.f;
    if (elem < total) {
        for (int g = gg; g < a.groups; g += 8) s += a.slabs[(size_t)g * total + elem];   // fixed order
    }
    part[gg][e_local] = s;
    __syncthreads();
    if (gg == 0 && elem < total) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += part[k][e_local];
        const int i = (int)(elem / a.fout), c = (int)(elem - (int64_t)i * a.fout);
        if (!a.partial) {
            v = v / fmaxf(a.indeg[i], 1.0f);
            float xr = 0.f;
            const float* xi = a.x + (int64_t)i * a.ld_x;
            for (int k = 0; k < a.fin; ++k) xr += xi[k] * a.root[k * a.fout + c];
            v += xr;
            if (a.bias) v += a.bias[c];
            if (a.relu) v = fmaxf(v, 0.f);
        }
        a.out[(int64_t)i * a.ld_out + c] = v;
    }
}

// ---- plan construction --------------------------------------------------------------------------
__global__ void k_seg_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                           const int64_t* __restrict__ range_start, int R, int64_t lo, int64_t hi, int ts, int tiles,
                           uint32_t* __restrict__ key, uint32_t* __restrict__ packed) {
    for (int64_t e = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < hi; e += (int64_t)gridDim.x * blockDim.x) {
        int a = 0, b = R;
        while (b - a > 1) {
            int mid = (a + b) >> 1;
            if (range_start[mid] <= e) a = mid; else b = mid;
        }
        const int s = (int)src[e], d = (int)dst[e];          // validated by the general plan builder
        const int tile = s / ts;
        key[e - lo] = (uint32_t)(a * tiles + tile);
        packed[e - lo] = ((uint32_t)d << 16) | (uint32_t)(s - tile * ts);
    }
}

__global__ void k_lower_bounds_u32(const uint32_t* __restrict__ sorted, int n, int count, int32_t* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > count) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sorted[mid] < (uint32_t)i) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
}

// key2 = (item << 32) | (bucket << 16) | dst, where item = the work item that holds position p
__global__ void k_item_keys(const uint32_t* __restrict__ packed, const int32_t* __restrict__ item_begin, int n_items,
                            int n, int nb, uint64_t* __restrict__ key2) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int a = 0, b = n_items;
    while (b - a > 1) {
        int mid = (a + b) >> 1;
        if (item_begin[mid] <= p) a = mid; else b = mid;
    }
    const uint32_t dst = packed[p] >> 16;
    key2[p] = ((uint64_t)a << 32) | ((uint64_t)(dst % (uint32_t)nb) << 16) | dst;
}

__global__ void k_bucket_offsets(const uint64_t* __restrict__ sorted, int n, int n_items, int nb,
                                 int32_t* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_items * nb) return;
    const uint64_t target = ((uint64_t)(i / nb) << 32) | ((uint64_t)(i % nb) << 16);
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sorted[mid] < target) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
}

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

struct Scratch {
    std::vector<void*> ptrs;
    ~Scratch() { for (void* p : ptrs) (void)hipFree(p); }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
        if (e == hipSuccess) ptrs.push_back(p);
        *out = static_cast<T*>(p);
        return e;
    }
};

constexpr int kPlanFout = 32;   // the bucket layout is built for 8-lane rows (out_features = 32)

template <int FIN, int FOUT>
gn_status launch_main(const FastArgs& a, int groups, size_t lds_bytes, hipStream_t st) {
    static thread_local bool configured = false;
    if (!configured) {
        GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_rgcn_lds<FIN, FOUT>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    k_rgcn_lds<FIN, FOUT><<<groups, kThreads, lds_bytes, st>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

// Builds the relation-major work items of the shard.  Leaves plan->fast_ok = 0 when the graph
// does not qualify (too many nodes for the LDS accumulator, too many relations for the key).
gn_status gn_rgcn_build_fast_segments(gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst,
                                      const std::vector<int64_t>& ranges, hipStream_t st) {
    plan->fast_ok = 0;
    const int64_t N = plan->num_nodes, R = plan->num_relations, E = plan->shard_edges;
    if (gn::fast_paths_disabled() || N < 1 || N > 65535 || R < 1 || E < 1) return GN_OK;
    const FastGeom g = geometry(N, kPlanFout);
    if (g.tiles == 0 || R * g.tiles >= (1 << 20)) return GN_OK;
    constexpr int NB = kWaves * (64 / (kPlanFout / 4));

    Scratch tmp;
    int64_t* starts_dev;
    uint32_t *key, *key_sorted, *packed, *packed_sorted;
    int32_t* seg_off;
    GN_HIP(tmp.get(&starts_dev, R + 1));
    GN_HIP(tmp.get(&key, E));
    GN_HIP(tmp.get(&key_sorted, E));
    GN_HIP(tmp.get(&packed, E));
    GN_HIP(tmp.get(&packed_sorted, E));
    const int n_seg = (int)(R * g.tiles);
    GN_HIP(tmp.get(&seg_off, n_seg + 1));
    std::vector<int64_t> starts(R + 1, plan->input_edges);
    for (int64_t r = 0; r < R; ++r) starts[r] = ranges[2 * r];
    GN_HIP(hipMemcpyAsync(starts_dev, starts.data(), (R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    k_seg_keys<<<gn::stream_grid(E, 256), 256, 0, st>>>(src, dst, starts_dev, (int)R, plan->edge_lo, plan->edge_hi,
                                                       g.ts, g.tiles, key, packed);
    GN_LAUNCH_CHECK();
    {
        size_t bytes = 0;
        GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key, key_sorted, packed, packed_sorted, (size_t)E, 0,
                                         bits_for(n_seg), st));
        char* scratch = nullptr;
        GN_HIP(tmp.get(&scratch, bytes));
        GN_HIP(rocprim::radix_sort_pairs(scratch, bytes, key, key_sorted, packed, packed_sorted, (size_t)E, 0,
                                         bits_for(n_seg), st));
    }
    k_lower_bounds_u32<<<(int)gn::ceil_div(n_seg + 1, 256), 256, 0, st>>>(key_sorted, (int)E, n_seg, seg_off);
    GN_LAUNCH_CHECK();
    std::vector<int32_t> seg(n_seg + 1);
    GN_HIP(hipMemcpyAsync(seg.data(), seg_off, (n_seg + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));

    // work items: every non-empty (relation, tile) segment, cut into chunks of <= kChunk edges
    std::vector<int32_t> item_rel, item_tile, item_begin;
    for (int s = 0; s < n_seg; ++s) {
        for (int32_t b = seg[s]; b < seg[s + 1]; b += kChunk) {
            item_rel.push_back(s / g.tiles);
            item_tile.push_back(s % g.tiles);
            item_begin.push_back(b);
        }
    }
    const int n_items = (int)item_rel.size();
    item_begin.push_back((int32_t)E);
    if ((int64_t)n_items * NB >= (1ll << 31)) return GN_OK;
    // longest-processing-time assignment of items to the persistent workgroups
    std::vector<int> order(n_items);
    std::iota(order.begin(), order.end(), 0);
    auto cost = [&](int i) { return (int64_t)(item_begin[i + 1] - item_begin[i]) + kItemOverhead; };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return cost(x) > cost(y); });
    const int groups = std::min(kGroups, std::max(n_items, 1));
    std::vector<int64_t> load(groups, 0);
    std::vector<std::vector<int32_t>> bins(groups);
    {
        // min-heap over (load, group)
        std::vector<std::pair<int64_t, int>> heap;
        for (int gidx = 0; gidx < groups; ++gidx) heap.emplace_back(0, gidx);
        auto cmp = [](const std::pair<int64_t, int>& x, const std::pair<int64_t, int>& y) { return x > y; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (int i : order) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto& top = heap.back();
            bins[top.second].push_back(i);
            top.first += cost(i);
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    std::vector<int32_t> wg_begin(groups + 1, 0), wg_items;
    for (int gidx = 0; gidx < groups; ++gidx) {
        std::sort(bins[gidx].begin(), bins[gidx].end());          // relation order: W_r / X tile reuse in L2
        wg_items.insert(wg_items.end(), bins[gidx].begin(), bins[gidx].end());
        wg_begin[gidx + 1] = (int32_t)wg_items.size();
    }

    int32_t* item_begin_dev;
    uint64_t *key2, *key2_sorted;
    GN_HIP(tmp.get(&item_begin_dev, n_items + 1));
    GN_HIP(tmp.get(&key2, E));
    GN_HIP(tmp.get(&key2_sorted, E));
    GN_HIP(hipMemcpyAsync(item_begin_dev, item_begin.data(), (n_items + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    k_item_keys<<<(int)gn::ceil_div(E, 256), 256, 0, st>>>(packed_sorted, item_begin_dev, n_items, (int)E, NB, key2);
    GN_LAUNCH_CHECK();
    GN_HIP(plan->packed.alloc(E));
    {
        size_t bytes = 0;
        GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key2, key2_sorted, packed_sorted, plan->packed.p, (size_t)E, 0,
                                         32 + bits_for(n_items), st));
        char* scratch = nullptr;
        GN_HIP(tmp.get(&scratch, bytes));
        GN_HIP(rocprim::radix_sort_pairs(scratch, bytes, key2, key2_sorted, packed_sorted, plan->packed.p, (size_t)E, 0,
                                         32 + bits_for(n_items), st));
    }
    GN_HIP(plan->seg_begin.alloc((size_t)n_items * NB + 1));
    k_bucket_offsets<<<(int)gn::ceil_div((int64_t)n_items * NB + 1, 256), 256, 0, st>>>(key2_sorted, (int)E, n_items, NB,
                                                                                      plan->seg_begin.p);
    GN_LAUNCH_CHECK();
    GN_HIP(plan->seg_rel.alloc(n_items));
    GN_HIP(plan->item_tile.alloc(n_items));
    GN_HIP(plan->wg_begin.alloc(groups + 1));
    GN_HIP(plan->wg_items.alloc(wg_items.size()));
    GN_HIP(hipMemcpyAsync(plan->seg_rel.p, item_rel.data(), n_items * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->item_tile.p, item_tile.data(), n_items * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->wg_begin.p, wg_begin.data(), (groups + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->wg_items.p, wg_items.data(), wg_items.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));       // host vectors go out of scope after this
    plan->n_seg = n_items;
    plan->fast_groups = groups;
    plan->fast_ts = g.ts;
    plan->fast_ts_pad = g.ts_pad;
    plan->fast_lds_bytes = g.lds_bytes;
    plan->fast_ok = 1;
    return GN_OK;
}

bool gn_rgcn_fast_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    if (!plan->fast_ok || gn::fast_paths_disabled()) return false;
    return fout == kPlanFout && (fin == 16 || fin == 32 || fin == 48 || fin == 64);
}

size_t gn_rgcn_fast_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    const size_t w = ((size_t)plan->num_relations * fin * fout * sizeof(float) + 255) & ~size_t(255);
    return w + (size_t)plan->fast_groups * plan->num_nodes * fout * sizeof(float);
}

gn_status gn_rgcn_fast_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin,
                               const float* basis, const float* att, int64_t bases, const float* root,
                               const float* bias, int64_t fout, int relu, int partial, float* out, int64_t ld_out,
                               void* ws, size_t ws_bytes, hipStream_t st) {
    GN_REQUIRE(ld_x % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned with ld_x %% 4 == 0");
    const int64_t N = plan->num_nodes, R = plan->num_relations;
    float* W = static_cast<float*>(ws);
    const size_t w_bytes = ((size_t)R * fin * fout * sizeof(float) + 255) & ~size_t(255);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + w_bytes);
    // K7: W[R, fin*fout] = att[R,B] @ basis[B, fin*fout]   (layers.py:172-173)
    gn_status s = gn_gemm_f32(att, bases, 0, nullptr, 0, basis, fin * fout, 0, W, fin * fout, 0, R, fin * fout, bases, 1,
                              nullptr, 0, st);
    if (s != GN_OK) return s;
    FastArgs a;
    a.x = x; a.ld_x = ld_x; a.n = (int)N; a.w = W; a.packed = plan->packed.p; a.bucket_off = plan->seg_begin.p;
    a.item_rel = plan->seg_rel.p; a.item_tile = plan->item_tile.p; a.wg_begin = plan->wg_begin.p;
    a.wg_items = plan->wg_items.p; a.ts = plan->fast_ts; a.ts_pad = plan->fast_ts_pad; a.slabs = slabs;
    const int groups = plan->fast_groups;
    switch (fin) {
        case 16: s = launch_main<16, 32>(a, groups, plan->fast_lds_bytes, st); break;
        case 32: s = launch_main<32, 32>(a, groups, plan->fast_lds_bytes, st); break;
        case 48: s = launch_main<48, 32>(a, groups, plan->fast_lds_bytes, st); break;
        default: s = launch_main<64, 32>(a, groups, plan->fast_lds_bytes, st); break;
    }
    if (s != GN_OK) return s;
    FinArgs f;
    f.slabs = slabs; f.groups = groups; f.n = (int)N; f.fout = (int)fout; f.indeg = plan->indeg.p; f.x = x; f.ld_x = ld_x;
    f.fin = (int)fin; f.root = root; f.bias = bias; f.relu = relu; f.partial = partial; f.out = out; f.ld_out = ld_out;
    k_rgcn_slab_finalize<<<(int)gn::ceil_div(N * fout, 32), 256, 0, st>>>(f);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
