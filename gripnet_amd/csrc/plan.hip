// One-time graph preprocessing ("plans") for the supergraph propagation kernels.
//
// A plan is what the reference caches on its first forward (myGCN.cached_result,
// gripnet/layers.py:83-90) plus the destination-major re-encoding the gfx950 kernels stream:
// int32 columns + fp32 coefficients in a CSR whose rows keep the reference's edge order
// (stable sort), so per-destination sums run in the same order as the reference's sequential
// scatter_add.  Plan construction is not on the steady-state path; it may synchronise.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <mutex>
#include <set>
#include <utility>
#include <vector>

namespace gn {
gn_status allow_large_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;     // hipFuncSetAttribute applies to the current device only
    int dev = 0;
    GN_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return GN_OK;
    GN_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    done.insert({kernel, dev});
    return GN_OK;
}

static thread_local LaunchEvents pending_launch_events;
LaunchEvents take_launch_events() {
    const LaunchEvents e = pending_launch_events;
    pending_launch_events = LaunchEvents{};
    return e;
}
}  // namespace gn

extern "C" {
// The next launch of an entry point that supports it (gn_rgcn_forward_f32 on the destination-major kernel,
// gn_distmult_plan_forward_f32 on the row-class kernel) carries these two HIP events as its dispatch's own start / stop stamps.
gn_status gn_time_next_launch(void* start_event, void* stop_event) {
    gn::pending_launch_events.start = static_cast<hipEvent_t>(start_event);
    gn::pending_launch_events.stop = static_cast<hipEvent_t>(stop_event);
    return GN_OK;
}
// 1 when the events of gn_time_next_launch are still waiting (the call in between launched a kernel that does not take them);
// clears them.
int gn_time_launch_pending(void) {
    const gn::LaunchEvents e = gn::take_launch_events();
    return (e.start || e.stop) ? 1 : 0;
}

// The builders' kept host block (host_layout.hpp: HostArena) back to the system.  No array outlives the build that took it from
// the block (ArenaHold), so a block nobody holds has no users.
size_t gn_host_scratch_release(void) {
    gn::HostArena& a = gn::host_arena();
    if (!a.lock.try_lock()) return 0;
    const size_t bytes = a.bytes.load();
    char* old = a.base.load();
    a.bytes.store(0);
    a.base.store(nullptr);
    std::free(old);
    a.want = 0;
    a.lock.unlock();
    return bytes;
}

// `later` waits for what `earlier` has been given so far.  A wait takes the event's state at the time of the call, so the events
// are reused round-robin (a ring per device and thread; 64 orderings can be in flight before an event is recorded again - and
// re-recording an event a stream still waits on is harmless: the wait was bound to the earlier record).
gn_status gn_stream_order(void* earlier_stream, void* later_stream) {
    if (earlier_stream == later_stream) return GN_OK;
    constexpr int kRing = 64, kDevices = 16;
    static thread_local hipEvent_t ring[kDevices][kRing] = {};
    static thread_local unsigned next[kDevices] = {};
    int dev = 0;
    GN_HIP(hipGetDevice(&dev));
    GN_REQUIRE(dev >= 0 && dev < kDevices, "device index %d out of range", dev);
    hipEvent_t& ev = ring[dev][next[dev]++ % kRing];
    if (!ev) GN_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    GN_HIP(hipEventRecord(ev, gn::as_stream(earlier_stream)));
    GN_HIP(hipStreamWaitEvent(gn::as_stream(later_stream), ev, 0));
    return GN_OK;
}
}

namespace {

using gn::DevBuf;

// ---- small device helpers ------------------------------------------------------------------
__device__ __forceinline__ int lower_bound_i32(const int32_t* a, int n, int v) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// keep[e] = 1 for valid non-loop edges; last_loop[i] = largest e with src=dst=i (or -1).
__global__ void k_mark_edges(const int64_t* __restrict__ src, const int64_t* __restrict__ dst, int64_t E,
                             int64_t n_src, int64_t n_dst, int drop_loops, int32_t* __restrict__ keep,
                             int32_t* __restrict__ last_loop, int32_t* __restrict__ err) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e <= E; e += (int64_t)gridDim.x * blockDim.x) {
        if (e == E) { keep[e] = 0; break; }   // sentinel so that the exclusive scan yields the total
        int64_t s = src[e], d = dst[e];
        bool ok = (uint64_t)s < (uint64_t)n_src && (uint64_t)d < (uint64_t)n_dst;
        if (!ok) { atomicOr(err, 1); keep[e] = 0; continue; }
        if (drop_loops && s == d) {
            atomicMax(&last_loop[s], (int32_t)e);
            keep[e] = 0;
        } else {
            keep[e] = 1;
        }
    }
}

// Non-loop edges keep their input order; then (GCN) one loop per node in node order.
__global__ void k_compact(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                          const float* __restrict__ w, int64_t E, int64_t n_loops, float fill,
                          const int32_t* __restrict__ keep, const int32_t* __restrict__ pos,
                          const int32_t* __restrict__ last_loop, int32_t* __restrict__ src2,
                          int32_t* __restrict__ dst2, float* __restrict__ w2, int32_t* __restrict__ iota) {
    const int32_t kept = pos[E];
    const int64_t total = E + n_loops;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        if (t < E) {
            if (keep[t]) {
                int32_t p = pos[t];
                src2[p] = (int32_t)src[t];
                dst2[p] = (int32_t)dst[t];
                w2[p] = w ? w[t] : 1.0f;
                iota[p] = p;
            }
        } else {
            int32_t i = (int32_t)(t - E);
            int32_t p = kept + i;
            int32_t ll = last_loop[i];
            src2[p] = i;
            dst2[p] = i;
            w2[p] = (ll >= 0) ? (w ? w[ll] : 1.0f) : fill;
            iota[p] = p;
        }
    }
}

__global__ void k_rowptr(const int32_t* __restrict__ sorted_dst, int nnz, int rows, int32_t* __restrict__ rowptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= rows) rowptr[i] = lower_bound_i32(sorted_dst, nnz, i);
}

// deg[i] = sum of weights into i in stored order (+ extra), dis = deg^-1/2 with inf -> 0.
__global__ void k_degree(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ perm,
                         const float* __restrict__ w2, int rows, float extra, float* __restrict__ dis) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows) return;
    float deg = 0.0f;
    for (int p = rowptr[i]; p < rowptr[i + 1]; ++p) deg += w2[perm[p]];
    deg += extra;
    float r = 1.0f / sqrtf(deg);
    if (r == INFINITY) r = 0.0f;
    dis[i] = r;
}

// flag[0] = 0 as soon as one stored weight differs from 1
__global__ void k_all_ones(const float* __restrict__ w, int64_t n, int32_t* __restrict__ flag) {
    bool ok = true;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) ok = ok && w[i] == 1.0f;
    if (!ok) atomicAnd(flag, 0);
}

// GCN: norm in the reference's order, then the CSR copies.
__global__ void k_gcn_norm(const int32_t* __restrict__ src2, const int32_t* __restrict__ dst2,
                           const float* __restrict__ w2, const float* __restrict__ dis, int nnz,
                           int64_t* __restrict__ ref_ei, float* __restrict__ ref_norm) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int s = src2[p], d = dst2[p];
    ref_ei[p] = s;
    ref_ei[(int64_t)nnz + p] = d;
    ref_norm[p] = dis[s] * w2[p] * dis[d];
}

__global__ void k_fill_csr(const int32_t* __restrict__ perm, const int32_t* __restrict__ src2,
                           const float* __restrict__ coef_in, int nnz, int32_t* __restrict__ col,
                           float* __restrict__ coef) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int o = perm[p];
    col[p] = src2[o];
    coef[p] = coef_in[o];
}

// Bipartite: coefficient = (1 * w) * dis[target]  (deg of a source is its self loop only).
__global__ void k_bip_coef(const int32_t* __restrict__ dst2, const float* __restrict__ w2,
                           const float* __restrict__ dis, int nnz, float* __restrict__ coef) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    coef[p] = (1.0f * w2[p]) * dis[dst2[p]];
}

// row_of[p] = destination row that holds CSR position p; iota[p] = p
__global__ void k_expand_rows(const int32_t* __restrict__ rowptr, int rows, int nnz, int32_t* __restrict__ row_of,
                              int32_t* __restrict__ iota) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nnz) return;
    int lo = 0, hi = rows;                 // last row with rowptr[row] <= p
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (rowptr[mid] <= p) lo = mid; else hi = mid;
    }
    row_of[p] = lo;
    iota[p] = p;
}

__global__ void k_fill_transpose(const int32_t* __restrict__ perm, const int32_t* __restrict__ row_of,
                                 const float* __restrict__ coef, int nnz, int32_t* __restrict__ t_col,
                                 float* __restrict__ t_coef) {
    int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nnz) return;
    const int p = perm[q];
    t_col[q] = row_of[p];
    t_coef[q] = coef[p];
}

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

using Temp = gn::Scratch;   // scoped device scratch for plan construction (common.h)

gn_status sort_by_dst(Temp& tmp, const int32_t* keys_in, int32_t* keys_out, const int32_t* vals_in,
                      int32_t* vals_out, int64_t n, int64_t key_range, hipStream_t st) {
    if (n == 0) return GN_OK;
    size_t bytes = 0;
    GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                     bits_for(key_range), st));
    char* scratch = nullptr;
    GN_HIP(tmp.get(&scratch, bytes));
    GN_HIP(rocprim::radix_sort_pairs(scratch, bytes, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0,
                                     bits_for(key_range), st));
    return GN_OK;
}

gn_status build_graph_plan(const int64_t* src, const int64_t* dst, const float* w, int64_t E, int64_t n_src,
                           int64_t n_dst, bool gcn, int improved, hipStream_t st, gn_graph_plan* plan,
                           bool raw = false) {
    const int64_t n_loops = gcn ? n_dst : 0;
    GN_LAP(nullptr);
    Temp tmp;
    GN_HIP(tmp.reserve((size_t)64 * (size_t)(E + n_dst) + ((size_t)1 << 20)));
    int32_t *keep, *pos, *last_loop, *err;
    GN_HIP(tmp.get(&keep, E + 1));
    GN_HIP(tmp.get(&pos, E + 1));
    GN_HIP(tmp.get(&last_loop, n_dst + 1));
    GN_HIP(tmp.get(&err, 1));
    GN_HIP(hipMemsetAsync(last_loop, 0xFF, (n_dst + 1) * sizeof(int32_t), st));
    GN_HIP(hipMemsetAsync(err, 0, sizeof(int32_t), st));
    k_mark_edges<<<gn::stream_grid(E + 1, 256), 256, 0, st>>>(src, dst, E, n_src, n_dst, gcn ? 1 : 0, keep,
                                                             last_loop, err);
    GN_LAUNCH_CHECK();
    {
        size_t bytes = 0;
        GN_HIP(rocprim::exclusive_scan(nullptr, bytes, keep, pos, 0, (size_t)(E + 1), rocprim::plus<int32_t>(), st));
        char* scratch = nullptr;
        GN_HIP(tmp.get(&scratch, bytes));
        GN_HIP(rocprim::exclusive_scan(scratch, bytes, keep, pos, 0, (size_t)(E + 1), rocprim::plus<int32_t>(), st));
    }
    int32_t kept = 0, bad = 0;
    GN_HIP(hipMemcpyAsync(&kept, pos + E, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipMemcpyAsync(&bad, err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("graph: mark + scan (sync)");
    if (bad) return gn::fail(GN_ERR_INDEX_RANGE, "edge_index holds a node id outside [0,%lld) x [0,%lld)",
                             (long long)n_src, (long long)n_dst);
    const int64_t nnz = (int64_t)kept + n_loops;
    if (nnz >= ((int64_t)1 << 31)) return gn::fail(GN_ERR_UNSUPPORTED, "more than 2^31 stored edges");

    int32_t *src2, *dst2, *iota, *sorted_dst, *perm;
    float *w2, *dis, *coef_ref;
    GN_HIP(tmp.get(&src2, nnz));
    GN_HIP(tmp.get(&dst2, nnz));
    GN_HIP(tmp.get(&iota, nnz));
    GN_HIP(tmp.get(&sorted_dst, nnz));
    GN_HIP(tmp.get(&perm, nnz));
    GN_HIP(tmp.get(&w2, nnz));
    GN_HIP(tmp.get(&dis, n_dst));
    GN_HIP(tmp.get(&coef_ref, nnz));
    if (E + n_loops > 0) {
        k_compact<<<gn::stream_grid(E + n_loops, 256), 256, 0, st>>>(src, dst, w, E, n_loops, improved ? 2.0f : 1.0f,
                                                                   keep, pos, last_loop, src2, dst2, w2, iota);
        GN_LAUNCH_CHECK();
    }
    gn_status s = sort_by_dst(tmp, dst2, sorted_dst, iota, perm, nnz, n_dst, st);
    if (s != GN_OK) return s;

    plan->input_edges = E;
    plan->nnz = nnz;
    plan->rows = n_dst;
    plan->table_rows = n_src;
    plan->is_gcn = gcn ? 1 : 0;
    GN_HIP(plan->rowptr.alloc(n_dst + 1));
    GN_HIP(plan->col.alloc(nnz + 8));                   // (k_aggregate_lds_table reads a row's ids eight at a time)
    GN_HIP(plan->coef.alloc(nnz));
    k_rowptr<<<(int)gn::ceil_div(n_dst + 1, 256), 256, 0, st>>>(sorted_dst, (int)nnz, (int)n_dst, plan->rowptr.p);
    GN_LAUNCH_CHECK();
    if (n_dst > 0) {
        k_degree<<<(int)gn::ceil_div(n_dst, 256), 256, 0, st>>>(plan->rowptr.p, perm, w2, (int)n_dst,
                                                               gcn ? 0.0f : 1.0f, dis);
        GN_LAUNCH_CHECK();
    }
    if (nnz > 0) {
        const int g = (int)gn::ceil_div(nnz, 256);
        if (gcn) {
            GN_HIP(plan->ref_edge_index.alloc(2 * nnz));
            GN_HIP(plan->ref_norm.alloc(nnz));
            k_gcn_norm<<<g, 256, 0, st>>>(src2, dst2, w2, dis, (int)nnz, plan->ref_edge_index.p, plan->ref_norm.p);
            GN_LAUNCH_CHECK();
            k_fill_csr<<<g, 256, 0, st>>>(perm, src2, plan->ref_norm.p, (int)nnz, plan->col.p, plan->coef.p);
        } else if (raw) {                                  // plain weighted sum: coefficient = edge weight
            k_fill_csr<<<g, 256, 0, st>>>(perm, src2, w2, (int)nnz, plan->col.p, plan->coef.p);
        } else {
            k_bip_coef<<<g, 256, 0, st>>>(dst2, w2, dis, (int)nnz, coef_ref);
            GN_LAUNCH_CHECK();
            k_fill_csr<<<g, 256, 0, st>>>(perm, src2, coef_ref, (int)nnz, plan->col.p, plan->coef.p);
        }
        GN_LAUNCH_CHECK();
    }
    int32_t ones = 0;
    if (gcn) {                                             // what the source-blocked path needs (gcn_blocked.hip)
        GN_HIP(plan->dis.alloc(n_dst));
        GN_HIP(hipMemcpyAsync(plan->dis.p, dis, n_dst * sizeof(float), hipMemcpyDeviceToDevice, st));
        GN_HIP(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(err), 1, 1, st));
        if (nnz > 0) {
            k_all_ones<<<gn::stream_grid(nnz, 256), 256, 0, st>>>(w2, nnz, err);
            GN_LAUNCH_CHECK();
        }
        GN_HIP(hipMemcpyAsync(&ones, err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    }
    std::vector<int32_t> rp(n_dst + 1);
    GN_HIP(hipMemcpyAsync(rp.data(), plan->rowptr.p, (n_dst + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("graph: compact, sort, CSR (sync)");
    plan->unit_weights = ones;
    plan->plain_ones = (raw && w == nullptr) ? 1 : 0;
    int64_t mx = 0;
    for (int64_t i = 0; i < n_dst; ++i) mx = std::max<int64_t>(mx, rp[i + 1] - rp[i]);
    plan->max_row_nnz = mx;
    return GN_OK;
}

__global__ void k_build_ell(const int32_t* __restrict__ rowptr, const int32_t* __restrict__ col, const float* __restrict__ coef,
                            int rows, uint32_t* __restrict__ ell_col, float* __restrict__ ell_coef) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= (int64_t)rows * 64) return;
    const int r = (int)(t >> 6), k = (int)(t & 63);
    const int at = rowptr[r] + k;
    const bool live = at < rowptr[r + 1];
    ell_col[t] = live ? (uint32_t)col[at] : 0xffffffffu;
    ell_coef[t] = live ? (coef ? coef[at] : 1.0f) : 0.f;
}

void free_graph_plan(gn_graph_plan* p) {
    p->ell_col.release();
    p->ell_coef.release();
    p->rowptr.release();
    p->col.release();
    p->coef.release();
    p->ref_edge_index.release();
    p->ref_norm.release();
    p->t_rowptr.release();
    p->t_col.release();
    p->t_coef.release();
    p->dis.release();
    p->blk_dis.release();
    p->blk_tile_off.release();
    p->blk_ids.release();
    p->blk_cell.release();
    p->blk_tile_rows.release();
    p->blk_tile_dis.release();
    p->blk_table.release();
}

// ---- RGCN ------------------------------------------------------------------------------------
// In-degree of every destination.  A supervertex of a few hundred nodes under millions of edges (PoSE: 645 under 2 M)
// puts thousands of atomic adds on every counter: each workgroup counts into an LDS histogram first when the counters
// fit there (8,192 nodes), and adds its non-zero bins once (520 -> 43 us on pose0-syn).
__global__ __launch_bounds__(256) void k_indegree(const int64_t* __restrict__ dst, int64_t E, int64_t N, int32_t* __restrict__ cnt,
                                                   int32_t* __restrict__ err) {
    __shared__ int32_t bins[8192];
    const bool local = N <= 8192;
    if (local) {
        for (int i = threadIdx.x; i < (int)N; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        int64_t d = dst[e];
        if ((uint64_t)d < (uint64_t)N) atomicAdd(local ? &bins[d] : &cnt[d], 1); else atomicOr(err, 1);
    }
    if (local) {
        __syncthreads();
        for (int i = threadIdx.x; i < (int)N; i += blockDim.x)
            if (bins[i] != 0) atomicAdd(&cnt[i], bins[i]);
    }
}

__global__ void k_i32_to_f32(const int32_t* __restrict__ a, int n, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)a[i];
}

// key = relation(e) * N + src(e); relation found from the range starts (edges are type-sorted).
__global__ void k_rel_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                           const int64_t* __restrict__ range_start, int R, int64_t lo, int64_t hi, int64_t N,
                           int32_t* __restrict__ dst32, uint32_t* __restrict__ key, uint32_t* __restrict__ skey, int32_t* __restrict__ err) {
    for (int64_t e = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < hi; e += (int64_t)gridDim.x * blockDim.x) {
        int a = 0, b = R;  // last r with range_start[r] <= e
        while (b - a > 1) {
            int mid = (a + b) >> 1;
            if (range_start[mid] <= e) a = mid; else b = mid;
        }
        int64_t s = src[e], d = dst[e];
        bool ok = (uint64_t)s < (uint64_t)N && (uint64_t)d < (uint64_t)N;
        if (!ok) { atomicOr(err, 1); s = 0; d = 0; }
        dst32[e - lo] = (int32_t)d;
        key[e - lo] = (uint32_t)((int64_t)a * N + s);
        if (skey) skey[e - lo] = (uint32_t)(s * (int64_t)R + a);       // source-major: sorted, a (destination, source) pair's edges are neighbours
    }
}

}  // namespace

// Implemented in rgcn_fast.hip: relation-major segments for the LDS-resident path.
gn_status gn_rgcn_build_fast_segments(gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst,
                                      const std::vector<int64_t>& ranges_host, hipStream_t st);

// Implemented in rgcn_pair.hip: per-workgroup destination rows and per-wave streams of the destination-major path.
gn_status gn_rgcn_build_pair_plan(gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst,
                                  const std::vector<int64_t>& ranges, hipStream_t st);

extern "C" {

int gn_version(void) { return GN_VERSION; }
const char* gn_last_error(void) { return gn::error_buffer(); }

gn_status gn_gcn_plan_create(const int64_t* src, const int64_t* dst, const float* w, int64_t E, int64_t N,
                             int improved, void* stream, gn_graph_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(E >= 0 && N >= 0, "negative size (E=%lld, N=%lld)", (long long)E, (long long)N);
    GN_REQUIRE(E == 0 || (src && dst), "edge pointers are null");
    if (E + N >= ((int64_t)1 << 31) || N >= ((int64_t)1 << 31))
        return gn::fail(GN_ERR_UNSUPPORTED, "graph too large for the int32 plan encoding");
    gn_graph_plan* p = new gn_graph_plan();
    gn_status s = build_graph_plan(src, dst, w, E, N, N, true, improved, gn::as_stream(stream), p);
    if (s != GN_OK) { free_graph_plan(p); delete p; return s; }
    *out = p;
    return GN_OK;
}

gn_status gn_bipartite_plan_create(const int64_t* src, const int64_t* dst, const float* w, int64_t E,
                                   int64_t n_src, int64_t n_tgt, void* stream, gn_graph_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(E >= 0 && n_src >= 0 && n_tgt >= 0, "negative size");
    GN_REQUIRE(E == 0 || (src && dst), "edge pointers are null");
    if (E >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || n_tgt >= ((int64_t)1 << 31))
        return gn::fail(GN_ERR_UNSUPPORTED, "graph too large for the int32 plan encoding");
    gn_graph_plan* p = new gn_graph_plan();
    gn_status s = build_graph_plan(src, dst, w, E, n_src, n_tgt, false, 0, gn::as_stream(stream), p);
    if (s != GN_OK) { free_graph_plan(p); delete p; return s; }
    // short rows (the external layer: 645 targets of ~29 edges): the padded layout as well
    if (p->rows > 0 && p->rows <= (1 << 20) && p->max_row_nnz <= 64 && p->nnz > 0 && !gn::fast_paths_disabled()) {
        hipError_t he = p->ell_col.alloc((size_t)p->rows * 64);
        if (he == hipSuccess) he = p->ell_coef.alloc((size_t)p->rows * 64);
        if (he != hipSuccess) { free_graph_plan(p); delete p; return gn::fail(GN_ERR_HIP, "padded rows: %s", hipGetErrorString(he)); }
        k_build_ell<<<(int)gn::ceil_div(p->rows * 64, 256), 256, 0, gn::as_stream(stream)>>>(p->rowptr.p, p->col.p, p->coef.p, (int)p->rows,
                                                                                            p->ell_col.p, p->ell_coef.p);
        if (hipGetLastError() != hipSuccess || hipStreamSynchronize(gn::as_stream(stream)) != hipSuccess) {
            free_graph_plan(p); delete p;
            return gn::fail(GN_ERR_HIP, "padded rows: the layout kernel failed");
        }
        p->ell_ok = 1;
    }
    *out = p;
    return GN_OK;
}

gn_status gn_sum_plan_create(const int64_t* src, const int64_t* dst, const float* w, int64_t E, int64_t n_src,
                             int64_t n_dst, void* stream, gn_graph_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(E >= 0 && n_src >= 0 && n_dst >= 0, "negative size");
    GN_REQUIRE(E == 0 || (src && dst), "edge pointers are null");
    if (E >= ((int64_t)1 << 31) || n_src >= ((int64_t)1 << 31) || n_dst >= ((int64_t)1 << 31))
        return gn::fail(GN_ERR_UNSUPPORTED, "graph too large for the int32 plan encoding");
    gn_graph_plan* p = new gn_graph_plan();
    gn_status s = build_graph_plan(src, dst, w, E, n_src, n_dst, false, 0, gn::as_stream(stream), p, true);
    if (s != GN_OK) { free_graph_plan(p); delete p; return s; }
    *out = p;
    return GN_OK;
}

void gn_graph_plan_destroy(gn_graph_plan* plan) {
    if (!plan) return;
    free_graph_plan(plan);
    delete plan;
}

int64_t gn_graph_plan_input_edges(const gn_graph_plan* plan) { return plan ? plan->input_edges : -1; }
int64_t gn_graph_plan_nnz(const gn_graph_plan* plan) { return plan ? plan->nnz : -1; }

gn_status gn_graph_plan_export(const gn_graph_plan* plan, int64_t* ei_out, float* norm_out, void* stream) {
    GN_REQUIRE(plan && plan->is_gcn, "export needs a GCN plan");
    GN_REQUIRE(plan->nnz == 0 || (ei_out && norm_out), "output pointers are null");
    if (plan->nnz == 0) return GN_OK;
    GN_HIP(hipMemcpyAsync(ei_out, plan->ref_edge_index.p, 2 * plan->nnz * sizeof(int64_t), hipMemcpyDeviceToDevice,
                          gn::as_stream(stream)));
    GN_HIP(hipMemcpyAsync(norm_out, plan->ref_norm.p, plan->nnz * sizeof(float), hipMemcpyDeviceToDevice,
                          gn::as_stream(stream)));
    return GN_OK;
}

gn_status gn_graph_plan_build_transpose(gn_graph_plan* plan, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    if (plan->has_transpose) return GN_OK;
    hipStream_t st = gn::as_stream(stream);
    const int64_t nnz = plan->nnz, rows = plan->rows, srcs = plan->table_rows;
    GN_HIP(plan->t_rowptr.alloc(srcs + 1));
    GN_HIP(plan->t_col.alloc(nnz));
    GN_HIP(plan->t_coef.alloc(nnz));
    Temp tmp;
    int32_t *row_of, *iota, *sorted_src, *perm;
    GN_HIP(tmp.get(&row_of, nnz));
    GN_HIP(tmp.get(&iota, nnz));
    GN_HIP(tmp.get(&sorted_src, nnz));
    GN_HIP(tmp.get(&perm, nnz));
    if (nnz > 0) {
        const int g = (int)gn::ceil_div(nnz, 256);
        k_expand_rows<<<g, 256, 0, st>>>(plan->rowptr.p, (int)rows, (int)nnz, row_of, iota);
        GN_LAUNCH_CHECK();
        // stable sort by source: inside a source row the destinations keep the CSR (= reference) order
        gn_status s = sort_by_dst(tmp, plan->col.p, sorted_src, iota, perm, nnz, srcs, st);
        if (s != GN_OK) return s;
        k_fill_transpose<<<g, 256, 0, st>>>(perm, row_of, plan->coef.p, (int)nnz, plan->t_col.p, plan->t_coef.p);
        GN_LAUNCH_CHECK();
    }
    k_rowptr<<<(int)gn::ceil_div(srcs + 1, 256), 256, 0, st>>>(sorted_src, (int)nnz, (int)srcs, plan->t_rowptr.p);
    GN_LAUNCH_CHECK();
    GN_HIP(hipStreamSynchronize(st));       // scratch is freed on return
    plan->has_transpose = 1;
    return GN_OK;
}

gn_status gn_rgcn_plan_create(const int64_t* src, const int64_t* dst, const int64_t* range_list, int on_host,
                              int64_t R, int64_t E, int64_t N, int64_t lo, int64_t hi, void* stream,
                              gn_rgcn_plan** out) {
    return gn_rgcn_plan_create_ex(src, dst, range_list, on_host, R, E, N, lo, hi, 0, stream, out);
}

gn_status gn_rgcn_plan_create_ex(const int64_t* src, const int64_t* dst, const int64_t* range_list, int on_host,
                                 int64_t R, int64_t E, int64_t N, int64_t lo, int64_t hi, int flags, void* stream,
                                 gn_rgcn_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(R >= 0 && E >= 0 && N >= 0, "negative size");
    GN_REQUIRE(E == 0 || (src && dst), "edge pointers are null");
    GN_REQUIRE(R == 0 || range_list, "range_list is null");
    GN_REQUIRE(0 <= lo && lo <= hi && hi <= E, "edge sub-range [%lld,%lld) outside [0,%lld)", (long long)lo,
               (long long)hi, (long long)E);
    if (E >= ((int64_t)1 << 31) || N >= ((int64_t)1 << 31) || R * N >= ((int64_t)1 << 32))
        return gn::fail(GN_ERR_UNSUPPORTED, "graph too large for the 32-bit plan encoding (R*N must be < 2^32)");
    hipStream_t st = gn::as_stream(stream);

    std::vector<int64_t> ranges(2 * R);
    if (R > 0) {
        if (on_host) {
            memcpy(ranges.data(), range_list, 2 * R * sizeof(int64_t));
        } else {
            GN_HIP(hipMemcpyAsync(ranges.data(), range_list, 2 * R * sizeof(int64_t), hipMemcpyDeviceToHost, st));
            GN_HIP(hipStreamSynchronize(st));
        }
    }
    int64_t cursor = 0;
    for (int64_t r = 0; r < R; ++r) {
        if (ranges[2 * r] != cursor || ranges[2 * r + 1] < ranges[2 * r])
            return gn::fail(GN_ERR_INVALID_ARG, "range_list must tile [0,E) in relation order (row %lld is [%lld,%lld), expected start %lld)",
                            (long long)r, (long long)ranges[2 * r], (long long)ranges[2 * r + 1], (long long)cursor);
        cursor = ranges[2 * r + 1];
    }
    if (cursor != E)
        return gn::fail(GN_ERR_INVALID_ARG, "range_list covers %lld edges but edge_index has %lld", (long long)cursor,
                        (long long)E);

    GN_LAP(nullptr);
    gn_rgcn_plan* p = new gn_rgcn_plan();
    auto bail = [&](gn_status s) { gn_rgcn_plan_destroy(p); return s; };
#define GN_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) return bail(gn::fail(GN_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e))); } while (0)
    p->input_edges = E; p->edge_lo = lo; p->edge_hi = hi; p->shard_edges = hi - lo;
    p->num_nodes = N; p->num_relations = R;
    Temp tmp;
    GN_TRY(tmp.reserve((size_t)24 * (size_t)(hi - lo) + (size_t)8 * (size_t)(N + R) + ((size_t)1 << 20)));
    int32_t *cnt, *err, *dst32, *sorted_dst;
    int64_t* starts_dev;
    uint32_t* key;
    GN_TRY(tmp.get(&cnt, N));
    GN_TRY(tmp.get(&err, 1));
    GN_TRY(tmp.get(&dst32, p->shard_edges));
    GN_TRY(tmp.get(&sorted_dst, p->shard_edges));
    GN_TRY(tmp.get(&key, p->shard_edges));
    GN_TRY(tmp.get(&starts_dev, R + 1));
    GN_TRY(hipMemsetAsync(cnt, 0, (N ? N : 1) * sizeof(int32_t), st));
    GN_TRY(hipMemsetAsync(err, 0, sizeof(int32_t), st));
    std::vector<int64_t> starts(R + 1, E);
    for (int64_t r = 0; r < R; ++r) starts[r] = ranges[2 * r];
    GN_TRY(hipMemcpyAsync(starts_dev, starts.data(), (R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    GN_TRY(p->indeg.alloc(N));
    GN_TRY(p->rowptr.alloc(N + 1));
    GN_TRY(p->key.alloc(p->shard_edges));
    if (E > 0) {
        k_indegree<<<gn::stream_grid(E, 256), 256, 0, st>>>(dst, E, N, cnt, err);
        GN_TRY(hipGetLastError());
    }
    if (N > 0) {
        k_i32_to_f32<<<(int)gn::ceil_div(N, 256), 256, 0, st>>>(cnt, (int)N, p->indeg.p);
        GN_TRY(hipGetLastError());
    }
    if (p->shard_edges > 0) {
        // (the source-ordered list of rgcn_basis.hip's measurement hook GN_RGCN_BASIS_ORDER: only built when it is set)
        const bool want_skey = getenv("GN_RGCN_BASIS_ORDER") != nullptr;
        uint32_t *skey = nullptr, *skey_s = nullptr;
        int32_t *dst_s = nullptr, *dst_s2 = nullptr;
        if (want_skey) {
            GN_TRY(tmp.get(&skey, p->shard_edges));
            GN_TRY(tmp.get(&skey_s, p->shard_edges));
            GN_TRY(tmp.get(&dst_s, p->shard_edges));
            GN_TRY(tmp.get(&dst_s2, p->shard_edges));
            GN_TRY(p->skey.alloc(p->shard_edges));
        }
        k_rel_keys<<<gn::stream_grid(p->shard_edges, 256), 256, 0, st>>>(src, dst, starts_dev, (int)R, lo, hi, N,
                                                                        dst32, key, skey, err);
        GN_TRY(hipGetLastError());
        size_t bytes = 0, bytes2 = 0;
        GN_TRY(rocprim::radix_sort_pairs(nullptr, bytes, dst32, sorted_dst, key, p->key.p, (size_t)p->shard_edges, 0,
                                         bits_for(N), st));
        if (want_skey)
            GN_TRY(rocprim::radix_sort_pairs(nullptr, bytes2, skey, skey_s, dst32, dst_s, (size_t)p->shard_edges, 0,
                                             bits_for(std::max<int64_t>(R * N, 2)), st));
        bytes = std::max(bytes, bytes2);
        char* scratch = nullptr;
        GN_TRY(tmp.get(&scratch, bytes));
        GN_TRY(rocprim::radix_sort_pairs(scratch, bytes, dst32, sorted_dst, key, p->key.p, (size_t)p->shard_edges, 0,
                                         bits_for(N), st));
        // the same edges by (destination, source, relation) (round 6: rgcn_basis.hip sums the att rows of a (destination, source)
        // pair's edges first and fetches x[source] once per pair): by the source-major key, then - stably - by destination
        if (want_skey) {
            GN_TRY(rocprim::radix_sort_pairs(scratch, bytes, skey, skey_s, dst32, dst_s, (size_t)p->shard_edges, 0,
                                             bits_for(std::max<int64_t>(R * N, 2)), st));
            GN_TRY(rocprim::radix_sort_pairs(scratch, bytes, dst_s, dst_s2, skey_s, p->skey.p, (size_t)p->shard_edges, 0,
                                             bits_for(N), st));
        }
    }
    k_rowptr<<<(int)gn::ceil_div(N + 1, 256), 256, 0, st>>>(sorted_dst, (int)p->shard_edges, (int)N, p->rowptr.p);
    GN_TRY(hipGetLastError());
    int32_t bad = 0;
    std::vector<int32_t> rp(N + 1);
    GN_TRY(hipMemcpyAsync(&bad, err, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_TRY(hipMemcpyAsync(rp.data(), p->rowptr.p, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_TRY(hipStreamSynchronize(st));
    GN_LAP("rgcn: keys + sort + rowptr (sync)");
    if (bad) return bail(gn::fail(GN_ERR_INDEX_RANGE, "edge_index holds a node id outside [0,%lld)", (long long)N));
    for (int64_t i = 0; i < N; ++i) p->max_row_nnz = std::max<int64_t>(p->max_row_nnz, rp[i + 1] - rp[i]);
    if (N > 0) {
        // rows by in-degree, largest first (host_layout.hpp: the row pointers are on the host anyway)
        std::vector<int32_t> order;
        gn_layout::degree_order(rp, order, p->heavy_rows);
        GN_TRY(p->row_order.alloc(N));
        GN_TRY(hipMemcpyAsync(p->row_order.p, order.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, st));
        GN_TRY(hipStreamSynchronize(st));
    }
    {
        // the general weight gradient's work items (host_layout.hpp)
        const gn_layout::RelDwItems dwl = gn_layout::build_rel_dw_items(ranges, lo, hi);
        p->n_dw_items = (int64_t)dwl.items.size() / 4;
        p->n_dw_parts = dwl.parts;
        p->n_dw_multi = (int64_t)dwl.multi.size() / 4;
        if (p->n_dw_items > 0) {
            GN_TRY(p->dw_items.alloc(dwl.items.size()));
            GN_TRY(hipMemcpyAsync(p->dw_items.p, dwl.items.data(), dwl.items.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            if (p->n_dw_multi > 0) {
                GN_TRY(p->dw_multi.alloc(dwl.multi.size()));
                GN_TRY(hipMemcpyAsync(p->dw_multi.p, dwl.multi.data(), dwl.multi.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
            }
            GN_TRY(hipStreamSynchronize(st));
        }
    }
    GN_LAP("rgcn: row order + dw items");
    if (!(flags & GN_RGCN_PLAN_LIGHT)) {
        // the encodings of the LDS-resident kernels (host-side schedules: most of the build time); a light plan serves every
        // forward on the general O(E) path, from the device-sorted key list above alone
        gn_status fs = gn_rgcn_build_fast_segments(p, src, dst, ranges, st);
        if (fs != GN_OK) return bail(fs);
        GN_LAP("rgcn: LDS-accumulator segments (total)");
        fs = gn_rgcn_build_pair_plan(p, src, dst, ranges, st);
        if (fs != GN_OK) return bail(fs);
        GN_LAP("rgcn: destination-major units (total)");
    }
#undef GN_TRY
    *out = p;
    return GN_OK;
}

void gn_rgcn_plan_destroy(gn_rgcn_plan* p) {
    if (!p) return;
    p->indeg.release();
    p->rowptr.release();
    p->key.release();
    p->skey.release();
    p->row_order.release();
    p->dw_items.release();
    p->dw_multi.release();
    p->seg_rel.release();
    p->item_tile.release();
    p->seg_begin.release();
    p->packed.release();
    p->wg_begin.release();
    p->wg_items.release();
    p->pair_stream.release();
    p->pair_wave_first.release();
    p->pair_desc.release();
    p->pair_wave_units.release();
    p->pair_wave_desc.release();
    p->pair_wg_dst.release();
    delete p;
}

int64_t gn_rgcn_plan_input_edges(const gn_rgcn_plan* plan) { return plan ? plan->input_edges : -1; }

}  // extern "C"
