// Per-relation link-prediction metrics on the device.
//
// After every epoch the reference computes, for each of the R relations, AUPRC / AUROC / AP of that relation's positive
// and negative scores with scikit-learn on the host: R device -> host copies and R sklearn calls per epoch
// (GripNet-pose.py:148-160,188-199; gripnet/utils.py:28-35).
//
// Round 6: own kernels that use the layout of the score vectors instead of a global 64-bit radix sort (rocPRIM: ~40
// launches and 715 us for the 2 x 2 M scores of pose0-syn's training list - more than the training step itself).  The scores of
// a relation are CONTIGUOUS in both vectors (type-sorted edge list, range_list), so every (relation, class) pair is a
// SEGMENT that is sorted on its own:
//   1. k_metric_sort_chunks: a segment is cut into chunks of <= 4,096 scores; a workgroup turns a chunk's scores into
//      32-bit keys (ascending key = descending score), sorts them in LDS (bitonic) and writes them back in place;
//   2. k_metric_merge (only when a segment has more than one chunk; log2(chunks) rounds, ping-pong): pairs of adjacent
//      sorted runs of one segment are merged by merge path, a workgroup per 1,024 outputs;
//   3. k_metric_terms: with both classes of a relation sorted, the four counts every curve point needs - positives and
//      negatives above / at-or-above a threshold - are an element's own index and two binary searches in the other class.
//      The LAST element of every group of tied scores of a class stands for the group: a positive one adds the terms of the
//      precision-recall trapezoid and of the average-precision step, a negative one the ROC trapezoid (the definitions
//      scikit-learn uses: roc_auc_score, average_precision_score, auc(precision_recall_curve) with the extra point
//      (recall 0, precision 1); thresholds where only the other class changes contribute zero to a curve's sum).  Terms are
//      summed in double precision, in a fixed order, per tile of 1,024 elements;
//   4. k_metric_fold: a wave per relation adds its tiles' sums in tile order.  Bitwise reproducible, no atomics.
// What depends on the range list only (segments, chunk / tile maps) is a PLAN (gn_link_metrics_plan_*): built once per
// list, a planned call is asynchronous and makes 4 + rounds launches.
#include "common.h"

#include <algorithm>
#include <vector>

namespace {

constexpr int kChunk = 4096;               // keys a workgroup sorts in LDS (16 KB)
constexpr int kSortThreads = 256;
constexpr int kTile = 1024;                // outputs of a merge workgroup, elements of a terms workgroup
constexpr int kTileThreads = 256;

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

__device__ __forceinline__ uint32_t descending_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone in x
    return ~asc;
}

// first index in [0, n) with a[idx] >= v (n if none)
__device__ __forceinline__ int lower_bound_u32(const uint32_t* __restrict__ a, int n, uint32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// first index in [0, n) with a[idx] > v (n if none)
__device__ __forceinline__ int upper_bound_u32(const uint32_t* __restrict__ a, int n, uint32_t v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
// last s with prefix[s] <= v, prefix ascending with prefix[0] = 0
__device__ __forceinline__ int owner_of(const int32_t* __restrict__ prefix, int n, int v) {
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (prefix[mid] <= v) lo = mid; else hi = mid;
    }
    return lo;
}

// Segment s = (relation s >> 1, class s & 1): class 0 = positives, 1 = negatives.  Keys of class c live at key[c * E + position].
struct Segs {
    const int64_t* start;      // [R + 1] positions of the relations' ranges (the range list's starts, then E)
    int64_t E;
    int S;                     // 2 R
    __device__ __forceinline__ int64_t base(int s) const { return (int64_t)(s & 1) * E + start[s >> 1]; }
    __device__ __forceinline__ int len(int s) const { return (int)(start[(s >> 1) + 1] - start[s >> 1]); }
};

// Bitonic sort of 256 x EPT keys held EPT per thread (thread t: elements EPT t .. EPT t + EPT - 1).  A compare-exchange at
// distance j < EPT stays inside a thread's registers; up to 32 threads apart it is a shuffle inside the wave; only the last
// stages of the largest size (64 and 128 threads apart) go through LDS: 3 exchanges with barriers instead of the 78 of a sort
// that keeps the keys in LDS (measured: 91 us per launch for that one, the launch ending with its slowest workgroup).
template <int EPT>
__device__ __forceinline__ void sort_in_registers(uint32_t* __restrict__ buf, int tid) {
    constexpr int LOGN = EPT == 16 ? 12 : (EPT == 4 ? 10 : 8);
    uint32_t v[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) v[e] = buf[EPT * tid + e];
#pragma unroll
    for (int lk = 1; lk <= LOGN; ++lk) {
        const int k = 1 << lk;
#pragma unroll
        for (int lj = lk - 1; lj >= 0; --lj) {
            const int j = 1 << lj;
            if (j < EPT) {
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    if (e & j) continue;
                    const bool up = ((EPT * tid + e) & k) == 0;
                    const uint32_t a = v[e], b = v[e | j];
                    const uint32_t lo = min(a, b), hi = max(a, b);
                    v[e] = up ? lo : hi;
                    v[e | j] = up ? hi : lo;
                }
            } else {
                const int td = j / EPT;                        // the partner is `td` threads away
                const bool lower = (tid & td) == 0;
                uint32_t other[EPT];
                if (td < 64) {
#pragma unroll
                    for (int e = 0; e < EPT; ++e) other[e] = (uint32_t)__shfl_xor((int)v[e], td);
                } else {
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < EPT; ++e) buf[EPT * tid + e] = v[e];
                    __syncthreads();
#pragma unroll
                    for (int e = 0; e < EPT; ++e) other[e] = buf[EPT * (tid ^ td) + e];
                }
#pragma unroll
                for (int e = 0; e < EPT; ++e) {
                    const bool up = ((EPT * tid + e) & k) == 0;
                    v[e] = (lower == up) ? min(v[e], other[e]) : max(v[e], other[e]);
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < EPT; ++e) buf[EPT * tid + e] = v[e];
    __syncthreads();
}

__global__ __launch_bounds__(kSortThreads) void k_metric_sort_chunks(const float* __restrict__ pos, const float* __restrict__ neg, Segs sg,
                                                                      const int32_t* __restrict__ chunk_prefix, const int32_t* __restrict__ chunk_seg,
                                                                      uint32_t* __restrict__ key) {
    __shared__ uint32_t buf[kChunk];
    const int s = chunk_seg[blockIdx.x];                      // (a table: the binary search over the prefix was eleven dependent loads in front of everything)
    const int c = (int)blockIdx.x - chunk_prefix[s];
    const int n_seg = sg.len(s);
    const int off = c * kChunk, n = min(kChunk, n_seg - off);
    const float* __restrict__ src = ((s & 1) ? neg : pos) + sg.start[s >> 1] + off;
    const int N = n <= kSortThreads ? kSortThreads : (n <= 4 * kSortThreads ? 4 * kSortThreads : 16 * kSortThreads);   // uniform
    for (int i = threadIdx.x; i < N; i += kSortThreads) buf[i] = i < n ? descending_bits(src[i]) : 0xffffffffu;
    __syncthreads();
    if (N == kSortThreads) sort_in_registers<1>(buf, threadIdx.x);
    else if (N == 4 * kSortThreads) sort_in_registers<4>(buf, threadIdx.x);
    else sort_in_registers<16>(buf, threadIdx.x);
    uint32_t* __restrict__ dst = key + sg.base(s) + off;
    for (int i = threadIdx.x; i < n; i += kSortThreads) dst[i] = buf[i];
}

// merge path: how many of the first `d` outputs of merge(A, B) come from A (ties: A first)
__device__ __forceinline__ int merge_path(const uint32_t* __restrict__ A, int na, const uint32_t* __restrict__ B, int nb, int d) {
    int lo = max(0, d - nb), hi = min(d, na);
    while (lo < hi) {
        const int i = (lo + hi) >> 1;          // take i from A, d - i from B: valid if A[i] > B[d - i - 1] is false ...
        if (A[i] <= B[d - 1 - i]) lo = i + 1; else hi = i;
    }
    return lo;
}

// One round: inside every segment with more than one chunk, runs of length L (the last may be shorter) are merged in pairs.
// Workgroup -> (big segment, tile of kTile outputs); 2 L is a multiple of kTile, so a tile never straddles two pairs.
__global__ __launch_bounds__(kTileThreads) void k_metric_merge(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, Segs sg,
                                                                const int32_t* __restrict__ big_seg, const int32_t* __restrict__ big_tile_prefix,
                                                                const int32_t* __restrict__ big_tile_owner, int L) {
    __shared__ uint32_t la[kTile + 1], lb[kTile + 1];
    __shared__ int split[2], cnt[4], open[2];
    const int bs = big_tile_owner[blockIdx.x];
    const int s = big_seg[bs];
    const int t = (int)blockIdx.x - big_tile_prefix[bs];
    const int n = sg.len(s);
    const int64_t base = sg.base(s);
    const int out0 = t * kTile;                                // first output of the tile inside the segment
    const int pair0 = (out0 / (2 * L)) * (2 * L);              // start of the pair of runs inside the segment
    const int na = min(L, n - pair0), nb = max(0, min(L, n - pair0 - L));
    const uint32_t* __restrict__ A = src + base + pair0;
    const uint32_t* __restrict__ B = A + na;
    const int d0 = out0 - pair0, d1 = min(d0 + kTile, na + nb);
    {
        // both splits at once, 128 threads each: every round samples 128 candidates of the remaining range (the predicate is
        // monotone: the number of samples that pass names the sub-range) - three rounds for runs of 65,536 where a binary search
        // by one thread made 17 dependent round trips (most of the launch's 10 us)
        const int grp = threadIdx.x >> 7, g = threadIdx.x & 127, d = grp ? d1 : d0;
        int lo = max(0, d - nb), hi = min(d, na);
        while (true) {                                         // (uniform across the workgroup: both groups loop until both are done)
            const int span = hi - lo, step = max(1, (span + 127) >> 7);
            const int i = lo + g * step;
            const bool ok = span > 0 && i < hi && A[i] <= B[d - 1 - i];
            const unsigned long long m = __ballot(ok);
            if ((threadIdx.x & 63) == 0) cnt[threadIdx.x >> 6] = __popcll(m);
            __syncthreads();
            const int c = cnt[2 * grp] + cnt[2 * grp + 1];
            const int nlo = c > 0 ? lo + (c - 1) * step + 1 : lo;
            const int nhi = (lo + c * step < hi) ? lo + c * step : hi;
            if (span > 0) { lo = nlo; hi = nhi; }
            if (g == 0) open[grp] = hi > lo ? 1 : 0;
            __syncthreads();
            if (!(open[0] | open[1])) break;
        }
        if (g == 0) split[grp] = lo;
        __syncthreads();
    }
    const int ia0 = split[0], ia1 = split[1], ib0 = d0 - ia0, ib1 = d1 - ia1;
    const int ca = ia1 - ia0, cb = ib1 - ib0;
    for (int i = threadIdx.x; i < ca; i += kTileThreads) la[i] = A[ia0 + i];
    for (int i = threadIdx.x; i < cb; i += kTileThreads) lb[i] = B[ib0 + i];
    __syncthreads();
    constexpr int PER = kTile / kTileThreads;
    const int q0 = min((int)threadIdx.x * PER, ca + cb);
    int i = merge_path(la, ca, lb, cb, q0), j = q0 - i;
    uint32_t* __restrict__ o = dst + base + out0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        if (q0 + q >= ca + cb) break;
        const bool from_a = j >= cb || (i < ca && la[i] <= lb[j]);
        o[q0 + q] = from_a ? la[i] : lb[j];
        if (from_a) ++i; else ++j;
    }
}

// Terms of the three curves, per tile of kTile elements of a segment; sums[tile] = (a, b): positives (PR trapezoid, AP step),
// negatives (ROC trapezoid, 0).  The tile's own keys (and one neighbour on each side) and the WINDOW of the other class that its
// first and last key bracket sit in LDS: the two searches per element run there (a search per element in global memory was ~40
// dependent L2 round trips per thread: 42 us per launch).  A window beyond kWindow keys is searched in global memory, inside its bounds.
constexpr int kWindow = 4096;

__global__ __launch_bounds__(kTileThreads) void k_metric_terms(const uint32_t* __restrict__ key_small, const uint32_t* __restrict__ key_big, Segs sg,
                                                                const int32_t* __restrict__ tile_prefix, const int32_t* __restrict__ tile_seg,
                                                                const uint8_t* __restrict__ is_big, double* __restrict__ sums) {
    __shared__ double red[2][kTileThreads];
    __shared__ uint32_t mine[kTile + 2], win[kWindow];
    __shared__ int bounds[2], scnt[4], sopen[2];
    const int s = tile_seg[blockIdx.x];
    const int t = (int)blockIdx.x - tile_prefix[s];
    const int n = sg.len(s);
    const uint32_t* __restrict__ own = (is_big[s] ? key_big : key_small) + sg.base(s);
    const uint32_t* __restrict__ oth = (is_big[s ^ 1] ? key_big : key_small) + sg.base(s ^ 1);     // the other class: same length
    const int t0 = t * kTile, t1 = min(t0 + kTile, n), cnt = t1 - t0;
    // mine[1 + q] = own[t0 + q]; mine[0] / mine[cnt + 1] = the neighbours outside the tile (or a key no element has: ~own)
    for (int q = threadIdx.x; q < cnt + 2; q += kTileThreads) {
        const int i = t0 - 1 + q;
        mine[q] = (i >= 0 && i < n) ? own[i] : ~own[min(max(i, 0), n - 1)];
    }
    {
        // the window's two bounds at once, 128 threads each, 128 samples of the remaining range per round (as in k_metric_merge):
        // group 0 the first index with oth >= the tile's first key, group 1 the first index with oth > its last key
        const int grp = threadIdx.x >> 7, g = threadIdx.x & 127;
        const uint32_t kk = grp ? own[t1 - 1] : own[t0];
        int lo = 0, hi = n;
        while (true) {
            const int span = hi - lo, step = max(1, (span + 127) >> 7);
            const int i = lo + g * step;
            const bool ok = span > 0 && i < hi && (grp ? oth[i] <= kk : oth[i] < kk);
            const unsigned long long m = __ballot(ok);
            if ((threadIdx.x & 63) == 0) scnt[threadIdx.x >> 6] = __popcll(m);
            __syncthreads();
            const int c = scnt[2 * grp] + scnt[2 * grp + 1];
            if (span > 0) {
                const int nlo = c > 0 ? lo + (c - 1) * step + 1 : lo;
                hi = (lo + c * step < hi) ? lo + c * step : hi;
                lo = nlo;
            }
            if (g == 0) sopen[grp] = hi > lo ? 1 : 0;
            __syncthreads();
            if (!(sopen[0] | sopen[1])) break;
        }
        if (g == 0) bounds[grp] = lo;
        __syncthreads();
    }
    const int wlo = bounds[0], whi = bounds[1], wn = whi - wlo;
    const bool in_lds = wn <= kWindow;
    if (in_lds)
        for (int q = threadIdx.x; q < wn; q += kTileThreads) win[q] = oth[wlo + q];
    __syncthreads();
    const uint32_t* __restrict__ wp = in_lds ? win : oth + wlo;
    const double P = (double)n;
    double a = 0.0, b = 0.0;
    constexpr int PER = kTile / kTileThreads;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int li = q * kTileThreads + (int)threadIdx.x, i = t0 + li;
        if (li >= cnt) continue;
        const uint32_t k = mine[1 + li];
        if (mine[2 + li] == k && i + 1 < n) continue;          // not the last of its group of ties
        const int own_le = i + 1;
        int own_lt = i;
        if (li > 0 ? mine[li] == k : (i > 0 && mine[0] == k)) {
            // the group began earlier: inside the tile (LDS), or - the tile's first key - in front of it (global, rare)
            own_lt = (mine[1] == k && t0 > 0 && mine[0] == k) ? lower_bound_u32(own, t0, k) : t0 + lower_bound_u32(mine + 1, li, k);
        }
        const int oth_lt = wlo + lower_bound_u32(wp, wn, k);
        const int oth_le = wlo + upper_bound_u32(wp, wn, k);
        if ((s & 1) == 0) {                                    // positives: tp0 = own_lt, tp = own_le, fp0 = oth_lt, fp = oth_le
            const double tp0 = own_lt, tp = own_le, fp0 = oth_lt, fp = oth_le;
            const double prec = tp / (tp + fp), prec0 = (tp0 + fp0) > 0.0 ? tp0 / (tp0 + fp0) : 1.0;
            a += (tp - tp0) / P * (prec + prec0) * 0.5;        // trapezoid of the PR curve
            b += (tp - tp0) / P * prec;                        // average precision step
        } else {                                               // negatives: fp0 = own_lt, fp = own_le, tp0 = oth_lt, tp = oth_le
            const double fp0 = own_lt, fp = own_le, tp0 = oth_lt, tp = oth_le;
            a += (fp - fp0) * (tp + tp0) * 0.5 / (P * P);      // trapezoid of the ROC curve (as many negatives as positives)
        }
    }
    red[0][threadIdx.x] = a; red[1][threadIdx.x] = b;
    __syncthreads();
    for (int w = kTileThreads / 2; w > 0; w >>= 1) {           // fixed tree: the same bits every run
        if ((int)threadIdx.x < w) { red[0][threadIdx.x] += red[0][threadIdx.x + w]; red[1][threadIdx.x] += red[1][threadIdx.x + w]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { sums[2 * (size_t)blockIdx.x] = red[0][0]; sums[2 * (size_t)blockIdx.x + 1] = red[1][0]; }
}

// out [3][R]: a wave per relation adds its tiles' sums in tile order (lanes stride over the tiles, then a fixed butterfly).
__global__ __launch_bounds__(256) void k_metric_fold(const double* __restrict__ sums, const int32_t* __restrict__ tile_prefix, Segs sg, int R,
                                                     double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int r = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    if (r >= R) return;
    double acc[3] = {0.0, 0.0, 0.0};                           // auprc, auroc, ap
    for (int c = 0; c < 2; ++c) {
        const int s = 2 * r + c;
        for (int t = tile_prefix[s] + lane; t < tile_prefix[s + 1]; t += 64) {
            if (c == 0) { acc[0] += sums[2 * (size_t)t]; acc[2] += sums[2 * (size_t)t + 1]; }
            else acc[1] += sums[2 * (size_t)t];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int k = 0; k < 3; ++k) acc[k] += __shfl_xor(acc[k], off);
    if (lane == 0) {
        const bool empty = sg.len(2 * r) == 0;
        const double nan = __builtin_nan("");
        out[r] = empty ? nan : acc[0];
        out[R + r] = empty ? nan : acc[1];
        out[2 * (size_t)R + r] = empty ? nan : acc[2];
    }
}

}  // namespace

struct gn_link_metrics_plan {
    int64_t R = 0, E = 0;
    int chunks = 0, tiles = 0, big_tiles = 0, n_big = 0, rounds = 0;
    gn::DevBuf<int64_t> start;            // [R + 1]
    gn::DevBuf<int32_t> chunk_prefix;     // [2 R + 1]
    gn::DevBuf<int32_t> tile_prefix;      // [2 R + 1]
    gn::DevBuf<int32_t> big_seg;          // [n_big] segments with more than one chunk
    gn::DevBuf<int32_t> big_tile_prefix;  // [n_big + 1]
    gn::DevBuf<uint8_t> is_big;           // [2 R]: the segment's sorted keys end in the buffer the last merge round wrote
    gn::DevBuf<int32_t> chunk_seg, tile_seg, big_tile_owner;   // owner of every chunk / tile / merge tile (a load instead of a search per workgroup)
    ~gn_link_metrics_plan() {
        start.release(); chunk_prefix.release(); tile_prefix.release(); big_seg.release(); big_tile_prefix.release(); is_big.release();
        chunk_seg.release(); tile_seg.release(); big_tile_owner.release();
    }
};

namespace {

struct Layout { size_t key_a, key_b, sums, total; };

Layout layout(int64_t E, int64_t tiles) {
    Layout l;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return at; };
    l.key_a = take(2 * (size_t)E * 4); l.key_b = take(2 * (size_t)E * 4); l.sums = take(2 * (size_t)std::max<int64_t>(tiles, 1) * 8);
    l.total = o;
    return l;
}

}  // namespace

extern "C" {

gn_status gn_link_metrics_plan_create(const int64_t* range_list_host, int64_t R, int64_t E, void* stream, gn_link_metrics_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(R >= 1 && E >= 0 && 2 * E < (1ll << 31) && 2 * R < (1ll << 30), "bad size (R=%lld, E=%lld)", (long long)R, (long long)E);
    GN_REQUIRE(range_list_host != nullptr, "range_list is null");
    std::vector<int64_t> start(R + 1);
    int64_t cursor = 0;
    for (int64_t r = 0; r < R; ++r) {
        GN_REQUIRE(range_list_host[2 * r] == cursor && range_list_host[2 * r + 1] >= cursor,
                   "range_list must tile [0,E) in relation order (row %lld)", (long long)r);
        start[r] = cursor;
        cursor = range_list_host[2 * r + 1];
    }
    GN_REQUIRE(cursor == E, "range_list covers %lld edges, %lld scores given", (long long)cursor, (long long)E);
    start[R] = E;
    const int S = (int)(2 * R);
    std::vector<int32_t> chunk_prefix(S + 1, 0), tile_prefix(S + 1, 0), big_seg, big_tile_prefix(1, 0);
    std::vector<uint8_t> is_big(S, 0);
    int64_t max_chunks = 1;
    for (int s = 0; s < S; ++s) {
        const int64_t n = start[(s >> 1) + 1] - start[s >> 1];
        const int64_t ch = (n + kChunk - 1) / kChunk, ti = (n + kTile - 1) / kTile;
        chunk_prefix[s + 1] = chunk_prefix[s] + (int32_t)ch;
        tile_prefix[s + 1] = tile_prefix[s] + (int32_t)ti;
        if (ch > 1) {
            big_seg.push_back(s);
            big_tile_prefix.push_back(big_tile_prefix.back() + (int32_t)ti);
            max_chunks = std::max(max_chunks, ch);
        }
    }
    gn_link_metrics_plan* p = new gn_link_metrics_plan();
    p->R = R; p->E = E; p->chunks = chunk_prefix[S]; p->tiles = tile_prefix[S];
    p->n_big = (int)big_seg.size(); p->big_tiles = big_tile_prefix.back();
    while ((1ll << p->rounds) < max_chunks) ++p->rounds;
    if (p->rounds & 1) for (int32_t s : big_seg) is_big[s] = 1;   // an odd number of ping-pong rounds ends in the second buffer
    hipStream_t st = gn::as_stream(stream);
    auto bail = [&](hipError_t e, const char* what) { delete p; return gn::fail(GN_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e)); };
#define GN_UP(buf, vec)                                                                                                  \
    do {                                                                                                                 \
        hipError_t e_ = p->buf.alloc(std::max<size_t>((vec).size(), 1));                                                 \
        if (e_ != hipSuccess) return bail(e_, "hipMalloc");                                                              \
        if (!(vec).empty()) {                                                                                            \
            e_ = hipMemcpyAsync(p->buf.p, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice, st);      \
            if (e_ != hipSuccess) return bail(e_, "hipMemcpyAsync");                                                     \
        }                                                                                                                \
    } while (0)
    GN_UP(start, start); GN_UP(chunk_prefix, chunk_prefix); GN_UP(tile_prefix, tile_prefix);
    GN_UP(big_seg, big_seg); GN_UP(big_tile_prefix, big_tile_prefix); GN_UP(is_big, is_big);
    std::vector<int32_t> chunk_seg, tile_seg, big_tile_owner;
    for (int s2 = 0; s2 < S; ++s2) {
        chunk_seg.insert(chunk_seg.end(), chunk_prefix[s2 + 1] - chunk_prefix[s2], s2);
        tile_seg.insert(tile_seg.end(), tile_prefix[s2 + 1] - tile_prefix[s2], s2);
    }
    for (size_t b2 = 0; b2 + 1 < big_tile_prefix.size(); ++b2)
        big_tile_owner.insert(big_tile_owner.end(), big_tile_prefix[b2 + 1] - big_tile_prefix[b2], (int32_t)b2);
    GN_UP(chunk_seg, chunk_seg); GN_UP(tile_seg, tile_seg); GN_UP(big_tile_owner, big_tile_owner);
#undef GN_UP
    hipError_t e = hipStreamSynchronize(st);                  // the vectors above are the sources of asynchronous copies
    if (e != hipSuccess) return bail(e, "hipStreamSynchronize");
    *out = p;
    return GN_OK;
}

void gn_link_metrics_plan_destroy(gn_link_metrics_plan* plan) { delete plan; }

size_t gn_link_metrics_plan_workspace_bytes(const gn_link_metrics_plan* plan) {
    if (!plan || plan->E == 0) return 0;
    return layout(plan->E, plan->tiles).total;
}

// out: [3, R] float64 (device): AUPRC, AUROC, AP of every relation (NaN for a relation without edges).  Asynchronous.
gn_status gn_link_metrics_planned_f32(const gn_link_metrics_plan* plan, const float* pos_score, const float* neg_score, double* out,
                                      void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(plan != nullptr && out != nullptr, "null pointer");
    hipStream_t st = gn::as_stream(stream);
    const int R = (int)plan->R;
    Segs sg{plan->start.p, plan->E, 2 * R};
    if (plan->E == 0) {
        k_metric_fold<<<(int)gn::ceil_div((int64_t)R * 64, 256), 256, 0, st>>>(nullptr, plan->tile_prefix.p, sg, R, out);
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    GN_REQUIRE(pos_score && neg_score, "score pointers are null");
    const Layout l = layout(plan->E, plan->tiles);
    GN_REQUIRE(workspace && workspace_bytes >= l.total, "workspace too small: need %zu bytes", l.total);
    char* ws = static_cast<char*>(workspace);
    uint32_t* ka = reinterpret_cast<uint32_t*>(ws + l.key_a);
    uint32_t* kb = reinterpret_cast<uint32_t*>(ws + l.key_b);
    double* sums = reinterpret_cast<double*>(ws + l.sums);
    k_metric_sort_chunks<<<plan->chunks, kSortThreads, 0, st>>>(pos_score, neg_score, sg, plan->chunk_prefix.p, plan->chunk_seg.p, ka);
    GN_LAUNCH_CHECK();
    uint32_t *src = ka, *dst = kb;
    for (int round = 0; round < plan->rounds; ++round) {
        k_metric_merge<<<plan->big_tiles, kTileThreads, 0, st>>>(src, dst, sg, plan->big_seg.p, plan->big_tile_prefix.p, plan->big_tile_owner.p,
                                                                  kChunk << round);
        GN_LAUNCH_CHECK();
        std::swap(src, dst);
    }
    // (single-chunk segments never left `ka`; the others are where the last round put them: `kb` after an odd number of rounds)
    k_metric_terms<<<plan->tiles, kTileThreads, 0, st>>>(ka, (plan->rounds & 1) ? kb : ka, sg, plan->tile_prefix.p, plan->tile_seg.p, plan->is_big.p, sums);
    GN_LAUNCH_CHECK();
    k_metric_fold<<<(int)gn::ceil_div((int64_t)R * 64, 256), 256, 0, st>>>(sums, plan->tile_prefix.p, sg, R, out);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

size_t gn_link_metrics_workspace_bytes(int64_t R, int64_t E) {
    if (R <= 0 || E <= 0) return 0;
    return layout(E, (E + kTile - 1) / kTile * 2 + 2 * R).total;      // (an upper bound of the tiles: every segment rounds up once)
}

// The same without a kept plan (one is built, used and dropped: synchronises `stream`).
gn_status gn_link_metrics_f32(const float* pos_score, const float* neg_score, const int64_t* range_list_host,
                              int64_t R, int64_t E, double* out, void* workspace, size_t workspace_bytes, void* stream) {
    gn_link_metrics_plan* plan = nullptr;
    gn_status s = gn_link_metrics_plan_create(range_list_host, R, E, stream, &plan);
    if (s != GN_OK) return s;
    s = gn_link_metrics_planned_f32(plan, pos_score, neg_score, out, workspace, workspace_bytes, stream);
    hipError_t e = hipStreamSynchronize(gn::as_stream(stream));
    gn_link_metrics_plan_destroy(plan);
    if (s == GN_OK && e != hipSuccess) return gn::fail(GN_ERR_HIP, "hipStreamSynchronize failed: %s", hipGetErrorString(e));
    return s;
}

}  // extern "C"
