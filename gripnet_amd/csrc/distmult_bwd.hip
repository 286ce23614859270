// Backward of the DistMult decoder (autograd of multiRelaInnerProductDecoder.forward,
// gripnet/decoder.py:19-23, as used by the loss of GripNet-pose.py:140-146):
//
//   s_e = sum_k z[u_e,k] z[v_e,k] D[r_e,k]        g_e = d loss / d s_e   (the caller folds the sigmoid in)
//   dz[i,:] = sum_{e: u_e=i} g_e z[v_e,:] * D[r_e,:]  +  sum_{e: v_e=i} g_e z[u_e,:] * D[r_e,:]
//   dD[r,:] = sum_{e: r_e=r} g_e z[u_e,:] * z[v_e,:]
//
// All three are the same segmented gather-reduce  out[key_e,:] += g_e * A[a_e,:] * B[b_e,:]  with
// (key, a, b) = (u, v, r), (v, u, r) and (r, u, v).  Scatter with float atomics is the wrong tool here:
// LDS float atomics retire at about one lane per three cycles per CU (measured: 1.7 ms for 2 M edges),
// and the targets are hit thousands of times each.  Instead every pass sorts 12-byte records
// (a, b, g) by key (rocPRIM radix sort, stable), and one workgroup per output row streams its records
// with coalesced loads, gathers the two factor rows from L2, sums in registers and folds the 16 edge
// lanes in a fixed order: no atomics, bitwise reproducible gradients.  dz is the sum of two passes.
//
// That general path gathers both factor rows of every record from L2 (2.5 GB per call at 2 M edges x 80
// features) behind three radix sorts.  When the tables are small - the drug supervertex: a few hundred nodes,
// ~10^3 relations - the LDS path below is used instead:
//   * dz: ONE sort of 2 E half-edge records (key = the node that receives, payload = the other endpoint, the
//     relation and g) - dz[i] = sum over the records of i of g z[other] * D[r]; a counting sort on wave-private LDS
//     histograms for up to 4,096 nodes (count pass, scan, scatter pass staged through LDS), rocPRIM's radix sort beyond;
//   * dD: when the caller says edge_type is sorted (GN_DM_TYPES_SORTED; it is in the reference's layout,
//     utils.py:168-198) the records stay in edge order and the row offsets come from a binary search;
//   * k_seg_lds: a workgroup keeps a 16-column block of both tables in LDS, a wave owns a task (<= 512 records of
//     one key), four lanes per record as in the forward kernel, per-lane sums over the task, one fold across the
//     16 quads, partial per task; k_seg_lds_combine adds the tasks of a key in order.  Still no float atomics.
//   * a static edge list keeps what depends on the triples only in a gn_distmult_bwd_plan (end of this file): the
//     pairing of an edge's two directions, the sort's offsets, the task lists.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <unordered_map>
#include <vector>

namespace {

struct Rec { uint32_t a, b; float g; };       // 12 bytes: factor rows and the edge's upstream gradient

// d loss / d logit of an edge: the caller's gradient, times sigma'(s) = p (1 - p) when the forward returned
// probabilities p (decoder.py:23) and the caller hands them over instead of folding the factor in itself
struct GradSrc {
    const float* g;            // null: the "gradient" of triple e is the bit pattern of e (a plan sorts the triples' positions once)
    const float* p;            // nullable
    // loss != 0 (round 6): the scores feed the link loss of GripNet-pose.py:140-142 directly - p are the probabilities, g points
    // at the loss's ONE upstream gradient (or is null: 1), and d loss / d p is computed here, exactly as gn_link_loss_backward_f32
    // computes it (-g / count / (p + eps) for the positives, +g / count / (1 - p + eps) for the negatives): the same bits without
    // the loss's backward launch and the gradient vector's round trip through memory
    int loss = 0;              // 0: none, 1: positives, 2: negatives
    float count = 1.f, eps = 0.f;
    __device__ __forceinline__ float at(int64_t e) const {
        // (no contraction of these products into a caller's add: the loss-fed form and the two-step form - the same expression
        // behind different branches - must round the same way, whatever the compiler makes of the code around them)
#pragma clang fp contract(off)
        if (loss) {
            const float up = g ? *g : 1.0f;
            const float q = p[e];
            const float v = loss == 1 ? (-up / count) / (q + eps) : (up / count) / (1.0f - q + eps);
            return v * q * (1.0f - q);
        }
        if (!g) return __int_as_float((int)e);
        const float v = g[e];
        if (!p) return v;
        const float q = p[e];
        return v * q * (1.0f - q);
    }
};

GradSrc make_grad_src(const float* grad_logit, const float* sigmoid_scores, const gn_link_loss_grad* loss, int64_t count) {
    GradSrc gs = {grad_logit, sigmoid_scores};
    if (loss) {
        gs.g = loss->upstream; gs.loss = loss->negative ? 2 : 1; gs.count = (float)count; gs.eps = loss->eps;
    }
    return gs;
}

// Where a call's triples come from: the int64 arrays of the reference's tensors, or - the negative samples as the sampler
// leaves them next to its int64 output - one 32-bit word per pair (u | v << 16) and a 16-bit relation id per position:
// 6 instead of 24 bytes per edge in each of the counting sort's two passes.
struct EdgeSrc {
    const int64_t* u; const int64_t* v; const int64_t* et;
    const uint32_t* packed; const uint16_t* rel16;
    __device__ __forceinline__ void load(int64_t e, int64_t& uu, int64_t& vv, int64_t& rr) const {
        if (packed) {
            const uint32_t w = packed[e];
            uu = w & 0xffffu; vv = w >> 16; rr = rel16[e];
        } else {
            uu = u[e]; vv = v[e]; rr = et[e];
        }
    }
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

// mode 0: key = u, a = v, b = r;   mode 1: key = v, a = u, b = r;   mode 2: key = r, a = u, b = v.
// Edges with an id outside its table get key = num_keys (sorted past the last row, never read).
__global__ void k_make_recs(int mode, const int64_t* __restrict__ u, const int64_t* __restrict__ v,
                            const int64_t* __restrict__ et, GradSrc gs, int64_t E, int64_t n,
                            int64_t R, uint32_t num_keys, uint32_t* __restrict__ keys, Rec* __restrict__ recs) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t uu = u[e], vv = v[e], rr = et[e];
        const bool ok = (uint64_t)uu < (uint64_t)n && (uint64_t)vv < (uint64_t)n && (uint64_t)rr < (uint64_t)R;
        Rec rec;
        uint32_t key;
        if (mode == 0) { key = (uint32_t)uu; rec.a = (uint32_t)vv; rec.b = (uint32_t)rr; }
        else if (mode == 1) { key = (uint32_t)vv; rec.a = (uint32_t)uu; rec.b = (uint32_t)rr; }
        else { key = (uint32_t)rr; rec.a = (uint32_t)uu; rec.b = (uint32_t)vv; }
        rec.g = ok ? gs.at(e) : 0.f;
        if (!ok) { key = num_keys; rec.a = 0; rec.b = 0; }
        keys[e] = key;
        recs[e] = rec;
    }
}

__global__ void k_key_offsets(const uint32_t* __restrict__ sorted, int64_t n, int rows, int32_t* __restrict__ rowptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted[mid] < (uint32_t)i) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = (int32_t)lo;
}

// Chunked segmented reduction.  The sorted record array is cut into chunks of kChunkRecs records, one
// workgroup each (hub rows of 10^5 records and rows of a few records cost the same per record).  A
// chunk holds a run of rows; for every row the workgroup sums  g * A[a, c..] * B[b, c..]  over the row's
// records inside the chunk: 256 threads = (256 / LPE) records in flight x LPE lanes of 16 bytes, folded
// in a fixed order.  A row that lies wholly inside the chunk is final and goes to `out`; the chunk's
// first / last row may continue in a neighbour chunk and goes to partial slot 2b / 2b+1 (columns
// [c0, c0+64)), which k_seg_combine adds up in chunk order: bitwise reproducible.
constexpr int kChunkRecs = 2048;

template <int LPE>
__global__ __launch_bounds__(256) void k_seg_reduce(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ keys,
                                                    const Rec* __restrict__ recs, int64_t n_recs, int rows,
                                                    const float* __restrict__ A, int64_t ld_a,
                                                    const float* __restrict__ B, int64_t ld_b, float* __restrict__ out,
                                                    int64_t ld_out, float* __restrict__ partial, int c0, int width,
                                                    int accumulate) {
    constexpr int S = 256 / LPE;                       // records per step
    __shared__ f32x4 part[S][LPE];
    const int j = threadIdx.x % LPE, slot = threadIdx.x / LPE;
    const int lo = blockIdx.x * kChunkRecs;
    const int hi = (int)min((int64_t)lo + kChunkRecs, n_recs);
    const bool col_ok = 4 * j < width;                 // width is a multiple of 4
    const int col = c0 + 4 * j;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    if (threadIdx.x < 2 * 16) {                        // both partial slots start at zero (64 columns each)
        reinterpret_cast<f32x4*>(partial)[(size_t)blockIdx.x * 32 + threadIdx.x] = zero4;
    }
    const int r_first = (int)min(keys[lo], (uint32_t)rows), r_last = (int)min(keys[hi - 1], (uint32_t)rows);
    for (int row = r_first; row <= r_last && row < rows; ++row) {
        const int rb = rowptr[row], re = rowptr[row + 1];
        const int begin = max(lo, rb), end = min(hi, re);
        if (begin >= end) continue;                    // workgroup-uniform
        f32x4 acc = zero4;
        for (int base = begin; base < end; base += 2 * S) {     // two records per thread and trip: four gathers in flight
            const int i0 = base + slot, i1 = base + S + slot;
            Rec r0 = {0, 0, 0.f}, r1 = {0, 0, 0.f};
            if (i0 < end) r0 = recs[i0];
            if (i1 < end) r1 = recs[i1];
            if (col_ok) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(A + (int64_t)r0.a * ld_a + col);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(B + (int64_t)r0.b * ld_b + col);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(A + (int64_t)r1.a * ld_a + col);
                const f32x4 b1 = *reinterpret_cast<const f32x4*>(B + (int64_t)r1.b * ld_b + col);
                acc += i0 < end ? r0.g * (a0 * b0) : zero4;   // loads past the end read row 0 and are dropped
                acc += i1 < end ? r1.g * (a1 * b1) : zero4;
            }
        }
        __syncthreads();                               // previous row's fold is done with `part`
        part[slot][j] = acc;
        __syncthreads();
        if (slot == 0 && col_ok) {
            f32x4 s = part[0][j];
#pragma unroll
            for (int k = 1; k < S; ++k) s += part[k][j];   // fixed order
            if (rb >= lo && re <= hi) {                // the whole row is here: final
                float* o = out + (int64_t)row * ld_out + col;
                if (accumulate) s += *reinterpret_cast<const f32x4*>(o);
                *reinterpret_cast<f32x4*>(o) = s;
            } else {                                   // shared with a neighbour chunk
                const int which = (row == r_first) ? 0 : 1;
                reinterpret_cast<f32x4*>(partial)[((size_t)blockIdx.x * 2 + which) * 16 + j] = s;
            }
        }
    }
}

// Rows that straddle chunks: out[row, c0 + c] (+)= sum of the chunks' partials, in chunk order.
__global__ void k_seg_combine(const int32_t* __restrict__ rowptr, const uint32_t* __restrict__ keys, int64_t n_recs,
                              int rows, const float* __restrict__ partial, float* __restrict__ out, int64_t ld_out,
                              int c0, int width, int accumulate) {
    const int row = blockIdx.x, c = threadIdx.x;
    if (c >= width) return;
    const int rb = rowptr[row], re = rowptr[row + 1];
    float* o = out + (int64_t)row * ld_out + c0 + c;
    if (rb >= re) {                                    // no record at all
        if (!accumulate) *o = 0.f;
        return;
    }
    const int cb = rb / kChunkRecs, ce = (re - 1) / kChunkRecs;
    if (cb == ce && rb >= cb * kChunkRecs && re <= min((int64_t)(cb + 1) * kChunkRecs, n_recs)) {
        // candidate for "whole row inside one chunk": final value already written by k_seg_reduce
        return;
    }
    float s = 0.f;
    for (int b = cb; b <= ce; ++b) {
        const int first = (int)min(keys[(size_t)b * kChunkRecs], (uint32_t)rows);
        const int which = (row == first) ? 0 : 1;
        s += partial[((size_t)b * 2 + which) * 64 + c];
    }
    *o = accumulate ? *o + s : s;
}

// scalar-column variant for shapes the float4 kernel cannot take (features % 4 != 0 or unaligned rows)
__global__ __launch_bounds__(256) void k_seg_reduce_scalar(const int32_t* __restrict__ rowptr, const Rec* __restrict__ recs,
                                                           const float* __restrict__ A, int64_t ld_a,
                                                           const float* __restrict__ B, int64_t ld_b,
                                                           float* __restrict__ out, int64_t ld_out, int features,
                                                           int accumulate) {
    const int row = blockIdx.x;
    const int begin = rowptr[row], end = rowptr[row + 1];
    for (int c = threadIdx.x; c < features; c += 256) {
        float s = 0.f;
        for (int i = begin; i < end; ++i) {
            const Rec r = recs[i];
            s += r.g * (A[(int64_t)r.a * ld_a + c] * B[(int64_t)r.b * ld_b + c]);
        }
        float* o = out + (int64_t)row * ld_out + c;
        *o = accumulate ? *o + s : s;
    }
}


// ---- LDS path --------------------------------------------------------------------------------------
constexpr int kTaskRecs = 512;                 // records per wave task
constexpr int kLdsThreads = 1024;
constexpr size_t kLdsTableBudget = 150 * 1024;

__device__ __forceinline__ uint64_t pack_rec(uint32_t a, uint32_t b, float g) {
    return (uint64_t)(a | (b << 16)) | ((uint64_t)__float_as_uint(g) << 32);
}

// two half-edge records per edge, in edge order (the sort is stable, so the records of a node stay in edge order)
__global__ void k_half_recs(const int64_t* __restrict__ u, const int64_t* __restrict__ v, const int64_t* __restrict__ et,
                            GradSrc gs, int64_t E, int64_t n, int64_t R, uint32_t* __restrict__ keys,
                            uint64_t* __restrict__ recs) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t uu = u[e], vv = v[e], rr = et[e];
        const bool ok = (uint64_t)uu < (uint64_t)n && (uint64_t)vv < (uint64_t)n && (uint64_t)rr < (uint64_t)R;
        const float g = ok ? gs.at(e) : 0.f;
        reinterpret_cast<uint2*>(keys)[e] = ok ? make_uint2((uint32_t)uu, (uint32_t)vv) : make_uint2((uint32_t)n, (uint32_t)n);
        recs[2 * e] = ok ? pack_rec((uint32_t)vv, (uint32_t)rr, g) : 0ull;
        recs[2 * e + 1] = ok ? pack_rec((uint32_t)uu, (uint32_t)rr, g) : 0ull;
    }
}

// one record per edge for dD: (u, v, g); keys only when a sort follows
__global__ void k_pair_recs(const int64_t* __restrict__ u, const int64_t* __restrict__ v, const int64_t* __restrict__ et,
                            GradSrc gs, int64_t E, int64_t n, int64_t R, uint32_t* __restrict__ keys,
                            uint64_t* __restrict__ recs) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t uu = u[e], vv = v[e], rr = et[e];
        const bool ok = (uint64_t)uu < (uint64_t)n && (uint64_t)vv < (uint64_t)n && (uint64_t)rr < (uint64_t)R;
        recs[e] = ok ? pack_rec((uint32_t)uu, (uint32_t)vv, gs.at(e)) : 0ull;
        if (keys) keys[e] = ok ? (uint32_t)rr : (uint32_t)R;
    }
}

// rowptr[i] = first position whose (sorted, int64) key is >= i
__global__ void k_key_offsets64(const int64_t* __restrict__ sorted, int64_t n, int rows, int32_t* __restrict__ rowptr) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > rows) return;
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if (sorted[mid] < (int64_t)i) lo = mid + 1; else hi = mid;
    }
    rowptr[i] = (int32_t)lo;
}

// taskptr[k] = number of tasks of the keys before k (a task = up to kTaskRecs records of one key), and the tasks
// themselves as (key, first record, end) descriptors; one workgroup.  The row offsets are read `stride` apart (the
// scanned (node, wave) counts of the counting sort are the offsets of wave 0).
__global__ __launch_bounds__(1024) void k_task_ptr(const int32_t* __restrict__ rowptr, int64_t stride, int keys,
                                                   int32_t* __restrict__ taskptr, int4* __restrict__ tasks) {
    __shared__ int32_t sums[1024];
    const int tid = threadIdx.x;
    const int strip = (keys + 1023) / 1024;
    const int k0 = min(tid * strip, keys), k1 = min(k0 + strip, keys);
    int32_t mine = 0;
    for (int k = k0; k < k1; ++k) mine += (rowptr[(k + 1) * stride] - rowptr[k * stride] + kTaskRecs - 1) / kTaskRecs;
    sums[tid] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                 // inclusive scan
        const int32_t add = tid >= d ? sums[tid - d] : 0;
        __syncthreads();
        sums[tid] += add;
        __syncthreads();
    }
    int32_t run = sums[tid] - mine;
    constexpr int kInLds = 4096;                          // keys whose offsets fit the LDS: descriptors written task-parallel
    __shared__ int32_t tp_l[kInLds + 1], rp_l[kInLds + 1];
    const bool par = keys <= kInLds;
    for (int k = k0; k < k1; ++k) {
        taskptr[k] = run;
        const int32_t b = rowptr[k * stride], e = rowptr[(k + 1) * stride];
        if (par) {
            tp_l[k] = run; rp_l[k] = b;
            if (k == keys - 1) rp_l[keys] = e;
            run += (e - b + kTaskRecs - 1) / kTaskRecs;
        } else {
            for (int32_t p = b; p < e; p += kTaskRecs) tasks[run++] = make_int4(k, p, min(e, p + kTaskRecs), 0);
        }
    }
    const int32_t total = sums[1023];
    if (tid == 1023) taskptr[keys] = total;
    if (!par) return;
    if (tid == 0) tp_l[keys] = total;
    __syncthreads();
    for (int t = tid; t < total; t += 1024) {             // a hub key has hundreds of tasks: one thread per task
        int lo = 0, hi = keys;                            // last k with tp_l[k] <= t (keys without tasks repeat a value)
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (tp_l[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int32_t b = rp_l[lo] + (t - tp_l[lo]) * kTaskRecs;
        tasks[t] = make_int4(lo, b, min(rp_l[lo + 1], b + kTaskRecs), 0);
    }
}


// ---- half-edge records straight into node order: a counting sort for a small key range -----------------------------
// kSortWaves waves own one contiguous slice of the edge list each.  Pass 1 counts a slice's records per node in a
// wave-private LDS histogram; an exclusive scan over (node, wave) turns the counts into write offsets; pass 2 walks
// the same slice again and places every record with a returning LDS add on the wave's private offsets.  Nothing is
// shared between waves, so the place of a record does not depend on timing: reproducible like the radix sort, at a
// third of its cost (one pass over the edges instead of two over 12-byte pairs, no separate record / offset kernels).
constexpr int kSortWavesPerWg = 4;
constexpr int kSortWaves = 2048;
static_assert(kSortWavesPerWg == 4, "k_he_scatter_staged reads the four wave offsets of a node as one int4");
constexpr int64_t kSortMaxKeys = 4096;        // 4 waves x 16 KB of histogram

template <bool SCATTER>
__global__ __launch_bounds__(kSortWavesPerWg * 64) void k_he_sort(EdgeSrc src, GradSrc gs,
                                                                   int64_t E, int n, int64_t R, int32_t* __restrict__ counts,
                                                                   uint64_t* __restrict__ recs, uint64_t* __restrict__ pair_recs) {
    extern __shared__ int32_t he_hist[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int w = blockIdx.x * kSortWavesPerWg + wave;
    int32_t* mine = he_hist + (size_t)wave * n;
    for (int b = lane; b < n; b += 64) mine[b] = SCATTER ? counts[(size_t)b * kSortWaves + w] : 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t per = (E + kSortWaves - 1) / kSortWaves;
    const int64_t e0 = w * per, e1 = min(E, e0 + per);
    // the triples of the next step are requested before this step's are placed (a step is one round trip otherwise)
    // (unconditional loads at a clamped index: hipcc waits for a conditional load on its own)
    if (SCATTER && e0 >= e1) return;
    const size_t dump = (size_t)2 * E;                   // 64 spare records behind the 2 E real ones
    int64_t at = max<int64_t>(0, min(e0 + lane, e1 - 1));
    int64_t nu, nv, nr;
    src.load(at, nu, nv, nr);
    float ng = SCATTER ? gs.at(at) : 0.f;
    for (int64_t base = e0; base < e1; base += 64) {
        const int64_t uu = nu, vv = nv, rr = nr;
        const float g = ng;
        const bool have = base + lane < e1;
        at = min(base + 64 + lane, e1 - 1);
        src.load(at, nu, nv, nr);
        if (SCATTER) ng = gs.at(at);
        const bool ok = have && (uint64_t)uu < (uint64_t)n && (uint64_t)vv < (uint64_t)n && (uint64_t)rr < (uint64_t)R;
        if (SCATTER) {
            // The stores are unconditional (edges that are dropped write to a spare slot past the last record): vector
            // memory retires in order, and behind a conditional store hipcc has to drain the queue - stores included -
            // before it may touch the triples requested above.
            size_t pu = dump + lane, pv = dump + lane;
            if (ok) {
                pu = (size_t)atomicAdd(&mine[uu], 1);
                pv = (size_t)atomicAdd(&mine[vv], 1);
            }
            recs[pu] = pack_rec((uint32_t)vv, (uint32_t)rr, g);
            recs[pv] = pack_rec((uint32_t)uu, (uint32_t)rr, g);
            // the (u, v, g) record of the dD pass, in edge order, while the triple is in registers
            if (pair_recs) pair_recs[have ? base + lane : dump + lane] = ok ? pack_rec((uint32_t)uu, (uint32_t)vv, g) : 0ull;
        } else if (ok) {
            atomicAdd(&mine[uu], 1);
            atomicAdd(&mine[vv], 1);
        }
    }
    if (!SCATTER) {
        // the four waves' counts of a node are adjacent words of the (node, wave) array: one 16-byte store
        __syncthreads();
        const int w0 = blockIdx.x * kSortWavesPerWg;
        for (int b = threadIdx.x; b < n; b += kSortWavesPerWg * 64)
            *reinterpret_cast<int4*>(counts + (size_t)b * kSortWaves + w0) =
                make_int4(he_hist[b], he_hist[n + b], he_hist[2 * n + b], he_hist[3 * n + b]);
        if (w == 0 && lane == 0) counts[(size_t)n * kSortWaves] = 0;       // the cell whose scan is the record count
    }
}


// From the (node, wave) counts of the counting sort to its write offsets and to the task list of the node-major reduction,
// in two small launches (a device-wide exclusive scan over the n x 2048 cells plus the one-workgroup k_task_ptr were 27 us of
// every backward call on a fresh edge list): k_node_totals adds up a node's 2048 cells; k_he_offsets, a workgroup per node,
// derives the node's first record and first task from the totals of the nodes before it (every workgroup adds them up itself:
// n words), scans the node's cells in place and writes the node's task descriptors.  Same values as the scan + k_task_ptr.
constexpr int kOffsetThreads = kSortWaves / 4;
static_assert(kOffsetThreads == 512, "a thread per four wave cells");

__global__ __launch_bounds__(kOffsetThreads) void k_node_totals(const int32_t* __restrict__ counts, int32_t* __restrict__ totals) {
    __shared__ int32_t part[kOffsetThreads / 64];
    const int tid = threadIdx.x;
    const int4 v = *reinterpret_cast<const int4*>(counts + (size_t)blockIdx.x * kSortWaves + 4 * tid);
    int32_t s = v.x + v.y + v.z + v.w;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d);
    if ((tid & 63) == 0) part[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        int32_t t = 0;
#pragma unroll
        for (int k = 0; k < kOffsetThreads / 64; ++k) t += part[k];
        totals[blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(kOffsetThreads) void k_he_offsets(int32_t* __restrict__ counts, const int32_t* __restrict__ totals, int n,
                                                               int32_t* __restrict__ taskptr, int4* __restrict__ tasks) {
    __shared__ int32_t part[2][kOffsetThreads / 64];
    __shared__ int32_t wave_base[kOffsetThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.x;
    // records and tasks of the nodes before this one
    int32_t recs = 0, tks = 0;
    for (int i = tid; i < b; i += kOffsetThreads) {
        const int32_t t = totals[i];
        recs += t;
        tks += (t + kTaskRecs - 1) / kTaskRecs;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) { recs += __shfl_xor(recs, d); tks += __shfl_xor(tks, d); }
    if (lane == 0) { part[0][wave] = recs; part[1][wave] = tks; }
    const int4 v = *reinterpret_cast<const int4*>(counts + (size_t)b * kSortWaves + 4 * tid);
    const int32_t mine = v.x + v.y + v.z + v.w;
    int32_t incl = mine;                                       // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int32_t up = __shfl_up(incl, d);
        if (lane >= d) incl += up;
    }
    if (lane == 63) wave_base[wave] = incl;
    __syncthreads();
    int32_t start = 0, first_task = 0, before = 0;
#pragma unroll
    for (int k = 0; k < kOffsetThreads / 64; ++k) {
        start += part[0][k];
        first_task += part[1][k];
        if (k < wave) before += wave_base[k];
    }
    const int32_t x = start + before + incl - mine;
    *reinterpret_cast<int4*>(counts + (size_t)b * kSortWaves + 4 * tid) = make_int4(x, x + v.x, x + v.x + v.y, x + v.x + v.y + v.z);
    const int32_t total = totals[b], end = start + total, my_tasks = (total + kTaskRecs - 1) / kTaskRecs;
    if (tid == 0) {
        taskptr[b] = first_task;
        if (b == n - 1) {
            taskptr[n] = first_task + my_tasks;
            counts[(size_t)n * kSortWaves] = end;              // the cell behind the last node: the record count
        }
    }
    for (int j = tid; j < my_tasks; j += kOffsetThreads)
        tasks[first_task + j] = make_int4(b, start + j * kTaskRecs, min(end, start + (j + 1) * kTaskRecs), 0);
}

// The scatter pass with the workgroup's records staged in LDS.  Placed directly (k_he_sort<true>), the 64 lanes of a
// store hit 64 different cache lines with 8 bytes each: the pass was bound by those partial-line writes (63 us, 18 us
// without them).  Here the waves of a workgroup own adjacent slices, so for every node the workgroup's records form
// one contiguous run of the output; they are placed in an LDS copy of that layout first (same wave-private offsets,
// shifted to the workgroup's base) and then copied out in order: consecutive lanes write consecutive records.
__global__ __launch_bounds__(kSortWavesPerWg * 64) void k_he_scatter_staged(EdgeSrc src, GradSrc gs,
                                                                             int64_t E, int n, int64_t R, const int32_t* __restrict__ offsets,
                                                                             uint64_t* __restrict__ recs, uint64_t* __restrict__ pair_recs,
                                                                             int stage_cap) {
    extern __shared__ int32_t he_hist[];
    constexpr int T = kSortWavesPerWg * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w0 = blockIdx.x * kSortWavesPerWg, w = w0 + wave;
    int32_t* loff = he_hist;                                   // [waves][n]
    int32_t* lbase = he_hist + (size_t)kSortWavesPerWg * n;    // [n + 1]: where a node's run starts in the stage
    int32_t* gbase = lbase + n + 1;                            // [n]: where it starts in the output
    uint64_t* stage = reinterpret_cast<uint64_t*>(gbase + n + 1);   // 6 n + 2 words in: 8-byte aligned
    __shared__ int32_t strip_sum[T];
    // Offsets of this workgroup's four waves and of the wave after them, per node: five adjacent words.  Run lengths of
    // the workgroup per node, their exclusive scan = where a node's run starts in the stage.
    const int strip = (n + T - 1) / T;
    const int b0 = min(tid * strip, n), b1 = min(b0 + strip, n);
    int32_t mine_sum = 0;
    for (int b = b0; b < b1; ++b) {
        const int32_t* o = offsets + (size_t)b * kSortWaves + w0;
        const int4 q = *reinterpret_cast<const int4*>(o);     // w0 is a multiple of four, the array 256-byte aligned
        const int32_t g1 = o[kSortWavesPerWg];
        gbase[b] = q.x;
        loff[b] = 0; loff[n + b] = q.y - q.x; loff[2 * n + b] = q.z - q.x; loff[3 * n + b] = q.w - q.x;
        lbase[b] = g1 - q.x;                                   // run length, replaced by the scan below
        mine_sum += g1 - q.x;
    }
    strip_sum[tid] = mine_sum;
    __syncthreads();
    for (int d = 1; d < T; d <<= 1) {
        const int32_t add = tid >= d ? strip_sum[tid - d] : 0;
        __syncthreads();
        strip_sum[tid] += add;
        __syncthreads();
    }
    int32_t run = strip_sum[tid] - mine_sum;
    for (int b = b0; b < b1; ++b) {
        const int32_t len = lbase[b];
        lbase[b] = run;
#pragma unroll
        for (int j = 0; j < kSortWavesPerWg; ++j) loff[j * n + b] += run;
        run += len;
    }
    if (tid == T - 1) lbase[n] = strip_sum[T - 1];
    __syncthreads();
    const int total = lbase[n];
    if (total <= stage_cap) {                                  // (always: the host sized the stage for a full slice)
        int32_t* mine = loff + (size_t)wave * n;
        const int64_t per = (E + kSortWaves - 1) / kSortWaves;
        const int64_t e0 = w * per, e1 = min(E, e0 + per);
        const size_t dump = (size_t)2 * E;
        if (e0 < e1) {
            int64_t at = min(e0 + lane, e1 - 1);
            int64_t nu, nv, nr;
            src.load(at, nu, nv, nr);
            float ng = gs.at(at);
            for (int64_t base = e0; base < e1; base += 64) {
                const int64_t uu = nu, vv = nv, rr = nr;
                const float g = ng;
                const bool have = base + lane < e1;
                at = min(base + 64 + lane, e1 - 1);
                src.load(at, nu, nv, nr); ng = gs.at(at);
                const bool ok = have && (uint64_t)uu < (uint64_t)n && (uint64_t)vv < (uint64_t)n && (uint64_t)rr < (uint64_t)R;
                if (ok) {
                    const int pu = atomicAdd(&mine[uu], 1);
                    stage[pu] = pack_rec((uint32_t)vv, (uint32_t)rr, g);
                    const int pv = atomicAdd(&mine[vv], 1);
                    stage[pv] = pack_rec((uint32_t)uu, (uint32_t)rr, g);
                }
                if (pair_recs) pair_recs[have ? base + lane : dump + lane] = ok ? pack_rec((uint32_t)uu, (uint32_t)vv, g) : 0ull;
            }
        }
        __syncthreads();
        // copy-out: sixteen lanes per node run (a run of the workgroup is ~12 records on pose0-syn)
        for (int b = tid >> 4; b < n; b += T / 16) {
            const int s0 = lbase[b], s1 = lbase[b + 1];
            uint64_t* dst = recs + (size_t)gbase[b] - s0;
            for (int i = s0 + (tid & 15); i < s1; i += 16) dst[i] = stage[i];
        }
    }
}

struct SegLdsArgs {
    const uint64_t* recs;        // sorted by key: a | b << 16 | g << 32
    const int4* tasks;           // (key, first record, end) per task
    const int32_t* n_tasks;      // device scalar
    const float* A; int64_t ld_a; int rows_a;
    const float* B; int64_t ld_b; int rows_b;       // B == A: one table serves both factors
    int f, col_blocks, workers;
    float* partial;              // [tasks][f]
};

// Workgroup = (column block of 16, worker); the blocks of both tables sit in LDS, 64 bytes per row.  Wave tasks are
// dealt round-robin.  Lane l of a wave loads record l of a 64-record batch; in step S the quad q works on record
// 4 q + S (quad_perm broadcast) and lane l4 of the quad covers columns 4 l4 .. 4 l4 + 3 of the block.
__global__ __launch_bounds__(kLdsThreads) void k_seg_lds(SegLdsArgs a) {
    extern __shared__ float4 seg_lds4[];
    const int tid = threadIdx.x, lane = tid & 63, l4 = lane & 3;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The column blocks of one worker read the SAME records: they sit on one XCD (workgroup i runs on XCD i % 8), so that one of
    // them pulls a batch from the memory side and the others find it in that XCD's L2 - spread over the XCDs every column block
    // streamed all records again (8 bytes x 5 blocks per record: more time than the LDS reads and the arithmetic together)
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = slot % a.col_blocks, worker = (slot / a.col_blocks) * 8 + xcd;
    const int c0 = cb * 16, w4 = min(4, (a.f - c0) / 4);
    const bool same = a.B == a.A;
    {
        const int total = (a.rows_a + (same ? 0 : a.rows_b)) * 4;
        for (int i = tid; i < total; i += kLdsThreads) {
            const int row = i >> 2, c4 = i & 3;
            const float* src = row < a.rows_a ? a.A + (int64_t)row * a.ld_a : a.B + (int64_t)(row - a.rows_a) * a.ld_b;
            seg_lds4[i] = c4 < w4 ? *reinterpret_cast<const float4*>(src + c0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    const char* lds_base = reinterpret_cast<const char*>(seg_lds4);
    const uint32_t ta32 = (uint32_t)l4 * 16u, tb32 = same ? ta32 : ta32 + (uint32_t)a.rows_a * 64u;   // byte offsets of the lane's chunk in the tables
    const int tasks = __builtin_amdgcn_readfirstlane(*a.n_tasks);
    const int t_step = a.workers * (kLdsThreads / 64);
    // The wave's tasks (t, t + t_step, ...) are ONE stream of 64-record batches, requested FOUR batches ahead of the arithmetic and
    // across task boundaries: with one batch ahead a wave waited a memory round trip (~0.8 us) per batch - 126 clocks per batch
    // and CU where the LDS reads need ~70 - and once more at every task's start.  Four fixed register sets (a rotation of
    // registers with loads in flight makes hipcc wait for all of them); what a batch belongs to travels in scalars.
    // (the descriptors are uniform: read through the scalar cache - as vector loads they would share the records' counter and
    // every use of one would wait for all record loads in flight)
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    typedef const __attribute__((address_space(4))) i32x4* scalar_desc_t;
    const scalar_desc_t descs = (scalar_desc_t)(uintptr_t)a.tasks;
    int ft = worker * (kLdsThreads / 64) + wave;             // fetch cursor: the task, its descriptor, the next one's, the position
    bool fvalid = ft < tasks;
    i32x4 fdesc = descs[fvalid ? ft : 0];
    i32x4 fnext = descs[ft + t_step < tasks ? ft + t_step : 0];
    int fpos = fdesc.y;
    if (!fvalid) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // (unconditional loads at a clamped index: past the wave's last task the cursor stays on that task and the batch is not used)
#define GN_SEG_FETCH(REC, CNT, TASK, LAST, VALID)                                                              \
    {                                                                                                         \
        VALID = fvalid; TASK = ft;                                                                            \
        REC = a.recs[min(fpos + lane, fdesc.z - 1)];          /* (masked where it is used: not here, behind the load) */ \
        CNT = fdesc.z - fpos;                                                                                 \
        LAST = fpos + 64 >= fdesc.z;                                                                          \
        if (!LAST) {                                                                                          \
            fpos += 64;                                                                                       \
        } else if (ft + t_step < tasks) {                                                                     \
            ft += t_step; fdesc = fnext; fpos = fdesc.y;                                                      \
            fnext = descs[ft + t_step < tasks ? ft + t_step : 0];                                             \
        } else {                                                                                              \
            fvalid = false;                                                                                   \
        }                                                                                                     \
    }
// (Measured with variants of this loop, decoder backward of pose0-syn, static list / fresh list: 79 / 157 us as it is, 74 / 136 us
// without the LDS row reads, 77 / 148 us without the arithmetic, 61 / 114 us without both: more than half of the reductions' time is
// neither - the record stream, the quad broadcasts, the bookkeeping of the batches in flight, the per-task folds.)
#define GN_SEG_MATH _Pragma("unroll") for (int s_ = 0; s_ < 4; ++s_) acc += gs[s_] * (va[s_] * vb[s_]);
#define GN_SEG_STEP(S)                                                                                    \
            {                                                                                             \
                const uint32_t pa_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)ra_, (S) * 0x55, 0xf, 0xf, true) + ta32;   \
                const uint32_t pb_ = (uint32_t)__builtin_amdgcn_mov_dpp((int)rb_, (S) * 0x55, 0xf, 0xf, true) + tb32;   \
                gs[S] = __int_as_float(__builtin_amdgcn_mov_dpp(gb, (S) * 0x55, 0xf, 0xf, true));         \
                va[S] = *reinterpret_cast<const f32x4*>(lds_base + pa_);                                  \
                vb[S] = *reinterpret_cast<const f32x4*>(lds_base + pb_);                                  \
            }
#define GN_SEG_CONSUME(REC, CNT, TASK, LAST)                                                              \
    {                                                                                                         \
        const uint64_t rec_ = lane < CNT ? REC : 0ull;                                                        \
        const int gb = (int)(uint32_t)(rec_ >> 32);                                                           \
        /* the byte offsets of this lane's record's two rows, once; the quad broadcast rides on the address add */ \
        const uint32_t ra_ = ((uint32_t)rec_ & 0xffffu) << 6, rb_ = ((uint32_t)rec_ >> 16) << 6;              \
        f32x4 va[4], vb[4];                                                                                   \
        float gs[4];                                                                                          \
        GN_SEG_STEP(0) GN_SEG_STEP(1) GN_SEG_STEP(2) GN_SEG_STEP(3)                                           \
        GN_SEG_MATH                                                                                           \
        if (LAST) {                                                                                \
            /* the 16 quads of the wave, in a fixed order: inside the rows of 16 lanes, then across the four rows */ \
            /* (ds_bpermute: once per task; hipcc folds the four DPP row rotations of a float4 into one, wrongly) */ \
            _Pragma("unroll") for (int c = 0; c < 4; ++c) {                                                   \
                float x = acc[c];                                                                             \
                x += __shfl_xor(x, 4);                                                                        \
                x += __shfl_xor(x, 8);                                                                        \
                x += __shfl_xor(x, 16);                                                                       \
                x += __shfl_xor(x, 32);                                                                       \
                acc[c] = x;                                                                                   \
            }                                                                                                 \
            if (lane < w4) *reinterpret_cast<f32x4*>(a.partial + (size_t)TASK * a.f + c0 + 4 * lane) = acc;  \
            acc = (f32x4){0.f, 0.f, 0.f, 0.f};                                                                \
        }                                                                                                     \
    }
    uint64_t r0, r1, r2, r3;
    int k0, k1, k2, k3, c0_, c1_, c2_, c3_;
    bool l0, l1, l2, l3, v0, v1, v2, v3;
    GN_SEG_FETCH(r0, c0_, k0, l0, v0)
    GN_SEG_FETCH(r1, c1_, k1, l1, v1)
    GN_SEG_FETCH(r2, c2_, k2, l2, v2)
    GN_SEG_FETCH(r3, c3_, k3, l3, v3)
    while (v0) {
        GN_SEG_CONSUME(r0, c0_, k0, l0)
        GN_SEG_FETCH(r0, c0_, k0, l0, v0)
        if (!v1) break;
        GN_SEG_CONSUME(r1, c1_, k1, l1)
        GN_SEG_FETCH(r1, c1_, k1, l1, v1)
        if (!v2) break;
        GN_SEG_CONSUME(r2, c2_, k2, l2)
        GN_SEG_FETCH(r2, c2_, k2, l2, v2)
        if (!v3) break;
        GN_SEG_CONSUME(r3, c3_, k3, l3)
        GN_SEG_FETCH(r3, c3_, k3, l3, v3)
    }
#undef GN_SEG_CONSUME
#undef GN_SEG_STEP
#undef GN_SEG_MATH
#undef GN_SEG_FETCH
}

// out[key, :] = the key's task partials.  One workgroup per key: eight slices take every eighth task each (four
// loads in flight), then the slices are added in slice order - a fixed order, whatever the number of tasks (a hub
// relation has hundreds).
struct CombineSet {
    const int32_t* taskptr; const float* partial; float* out; int64_t ld_out;
    const float* add = nullptr; int64_t ld_add = 0;      // (optional) another gradient of the same tensor, added where the sum is stored
};

__global__ __launch_bounds__(256) void k_seg_lds_combine(CombineSet first, int first_keys, CombineSet second, int f) {
    // (both reductions of a call in one launch: the node keys, then the relation keys)
    const bool in_first = (int)blockIdx.x < first_keys;
    const int32_t* __restrict__ taskptr = in_first ? first.taskptr : second.taskptr;
    const float* __restrict__ partial = in_first ? first.partial : second.partial;
    float* __restrict__ out = in_first ? first.out : second.out;
    const int64_t ld_out = in_first ? first.ld_out : second.ld_out;
    const float* __restrict__ add = in_first ? first.add : second.add;
    const int64_t ld_add = in_first ? first.ld_add : second.ld_add;
    __shared__ f32x4 fold[8][32];
    const int key = in_first ? (int)blockIdx.x : (int)blockIdx.x - first_keys, slice = threadIdx.x >> 5, j = threadIdx.x & 31;
    const int t0 = taskptr[key], t1 = taskptr[key + 1];
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < f; c0 += 128) {                 // 32 float4 lanes per pass
        const bool col_ok = c0 + 4 * j < f;
        f32x4 s = zero4;
        if (col_ok) {
            const float* p = partial + c0 + 4 * j;
            int t = t0 + slice;
            for (; t + 24 < t1; t += 32) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(p + (size_t)t * f);
                const f32x4 b = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 8) * f);
                const f32x4 c = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 16) * f);
                const f32x4 d = *reinterpret_cast<const f32x4*>(p + (size_t)(t + 24) * f);
                s += a; s += b; s += c; s += d;
            }
            for (; t < t1; t += 8) s += *reinterpret_cast<const f32x4*>(p + (size_t)t * f);
        }
        __syncthreads();
        fold[slice][j] = s;
        __syncthreads();
        if (slice == 0 && col_ok) {
            f32x4 r = fold[0][j];
#pragma unroll
            for (int k = 1; k < 8; ++k) r += fold[k][j];
            if (add) r += *reinterpret_cast<const f32x4*>(add + (int64_t)key * ld_add + c0 + 4 * j);
            *reinterpret_cast<f32x4*>(out + (int64_t)key * ld_out + c0 + 4 * j) = r;
        }
    }
}

struct WsLayout { size_t keys, keys_sorted, recs, recs_sorted, rowptr, partial, sort_tmp, total; };

WsLayout ws_layout(int64_t e, int64_t max_rows) {
    WsLayout l;
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const Rec*)nullptr,
                                    (Rec*)nullptr, (size_t)e, 0, 32, (hipStream_t)0);
    l.keys = 0;
    l.keys_sorted = l.keys + align_up(e * sizeof(uint32_t));
    l.recs = l.keys_sorted + align_up(e * sizeof(uint32_t));
    l.recs_sorted = l.recs + align_up(e * sizeof(Rec));
    l.rowptr = l.recs_sorted + align_up(e * sizeof(Rec));
    l.partial = l.rowptr + align_up((max_rows + 2) * sizeof(int32_t));
    l.sort_tmp = l.partial + align_up((size_t)gn::ceil_div(e, kChunkRecs) * 2 * 64 * sizeof(float));
    l.total = l.sort_tmp + align_up(sort_bytes);
    return l;
}

struct LdsLayout { size_t keys, keys_sorted, recs, recs_sorted, rowptr, taskptr, tasks, partial, counts, sort_tmp, sort_tmp_bytes, total; };

int64_t lds_max_tasks(int64_t records, int64_t keys) { return records / kTaskRecs + keys + 1; }

LdsLayout lds_layout(int64_t e, int64_t n, int64_t r, int64_t f) {
    LdsLayout l;
    size_t sort_bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint64_t*)nullptr,
                                    (uint64_t*)nullptr, (size_t)(2 * e), 0, 32, (hipStream_t)0);
    const int64_t rows = std::max(n, r);
    const int64_t tasks = lds_max_tasks(2 * e, n) + lds_max_tasks(e, r);        // both reductions' partial sums side by side
    l.keys = 0;
    l.keys_sorted = l.keys + align_up(2 * e * sizeof(uint32_t));
    l.recs = l.keys_sorted + align_up(2 * e * sizeof(uint32_t));
    l.recs_sorted = l.recs + align_up((2 * e + 64) * sizeof(uint64_t));
    l.rowptr = l.recs_sorted + align_up((2 * e + 64) * sizeof(uint64_t));
    l.taskptr = l.rowptr + align_up((rows + 2) * sizeof(int32_t));
    l.tasks = l.taskptr + align_up((rows + 2) * sizeof(int32_t));
    l.partial = l.tasks + align_up((size_t)tasks * sizeof(int4));
    l.counts = l.partial + align_up((size_t)tasks * f * sizeof(float));
    const size_t cells = n <= kSortMaxKeys ? (size_t)n * kSortWaves + 1 : 1;
    size_t scan_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0, cells, rocprim::plus<int32_t>(),
                                  (hipStream_t)0);
    l.sort_tmp = l.counts + align_up(cells * sizeof(int32_t));
    l.sort_tmp_bytes = align_up(std::max(sort_bytes, scan_bytes));
    l.total = l.sort_tmp + l.sort_tmp_bytes;
    return l;
}

// what the LDS path takes: 16-bit ids in the packed record, float4 columns, the table blocks inside the LDS
bool lds_path_shapes(int64_t n, int64_t f, int64_t r) {
    return !gn::fast_paths_disabled() && n <= 65535 && r <= 65535 && f % 4 == 0 && f >= 4;
}
bool lds_dz_fits(int64_t n, int64_t r) { return (size_t)(n + r) * 64 <= kLdsTableBudget; }
bool lds_dd_fits(int64_t n) { return (size_t)n * 64 <= kLdsTableBudget; }

gn_status launch_seg_lds(const uint64_t* recs, const int32_t* rowptr, int64_t rowptr_stride, int32_t* taskptr, int4* tasks, int64_t keys,
                         const float* A, int64_t ld_a,
                         int64_t rows_a, const float* B, int64_t ld_b, int64_t rows_b, int64_t f, float* partial, float* out,
                         int64_t ld_out, hipStream_t st, bool tasks_ready = false, CombineSet* defer = nullptr) {
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_seg_lds), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    if (!tasks_ready) {
        k_task_ptr<<<1, 1024, 0, st>>>(rowptr, rowptr_stride, (int)keys, taskptr, tasks);
        GN_LAUNCH_CHECK();
    }
    SegLdsArgs a;
    a.recs = recs; a.tasks = tasks; a.n_tasks = taskptr + keys;
    a.A = A; a.ld_a = ld_a; a.rows_a = (int)rows_a; a.B = B; a.ld_b = ld_b; a.rows_b = (int)rows_b;
    a.f = (int)f; a.col_blocks = (int)gn::ceil_div(f, 16);
    const size_t lds_bytes = (size_t)(rows_a + (B == A ? 0 : rows_b)) * 64;
    const int per_cu = lds_bytes * 2 <= 156 * 1024 ? 2 : 1;           // 1024-thread workgroups: two per CU at most
    a.workers = 8 * std::max(1, 32 * per_cu / a.col_blocks);  // per XCD: as many workers as its 32 CUs hold, all column blocks of a worker together
    a.partial = partial;
    k_seg_lds<<<a.col_blocks * a.workers, kLdsThreads, lds_bytes, st>>>(a);
    GN_LAUNCH_CHECK();
    const CombineSet mine = {taskptr, partial, out, ld_out};
    if (defer) {                                               // the caller combines two reductions in one launch (combine_both)
        *defer = mine;
        return GN_OK;
    }
    k_seg_lds_combine<<<(unsigned)keys, 256, 0, st>>>(mine, (int)keys, mine, (int)f);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status combine_both(const CombineSet& a, int64_t keys_a, const CombineSet& b, int64_t keys_b, int64_t f, hipStream_t st) {
    k_seg_lds_combine<<<(unsigned)(keys_a + keys_b), 256, 0, st>>>(a, (int)keys_a, b, (int)f);
    GN_LAUNCH_CHECK();
    return GN_OK;
}


// ---- a static edge list (the positive edges of a training loop: the same tensors every epoch) -----------------------
// What the sort and the reductions derive from the triples alone is kept: the pairing of an edge's two directions, the
// task lists of both reductions and the SORTED RECORDS themselves, with the position of a record's pair where its gradient
// goes.  A call computes the pairs' gradients (k_pair_grad, which also completes the list-order records of the dD pass),
// places them into the node-order records (k_place_g: 2 M random 4-byte reads, 19.5 us - the scatter pass it replaces took
// 26 us plus the list-order records; without the pairing the gather is 4 M reads and loses to the scatter) and runs the two
// segment reductions and their common combine: 5 launches.
}  // namespace

struct gn_distmult_bwd_plan {
    int64_t e = 0, n = 0, r = 0, he_tasks_max = 0, pr_tasks_max = 0;   // e: triples after pairing (below)
    int64_t e_list = 0;                             // triples of the caller's list
    // Two triples with the same unordered node pair and relation (the two directions of an edge) produce the same
    // records but for g: they are reduced as ONE triple with g1 + g2.  eu / ev / er: the triples that are left, in list
    // order; own / mir: where their gradients sit in the caller's list (mir = kNoPair: unpaired).
    gn::DevBuf<int64_t> eu, ev, er;
    gn::DevBuf<uint32_t> own, mir;
    gn::DevBuf<int32_t> offsets;                    // [n * kSortWaves + 1] where every wave's records of every node start
    gn::DevBuf<int32_t> he_taskptr, pr_taskptr;     // [n + 1], [R + 1]
    gn::DevBuf<int32_t> he_tasks, pr_tasks;         // int4 descriptors
    // The half-edge records in node order and the pair records in list order with the POSITION of their triple (in eu / ev /
    // er) where the gradient goes: what a step's gradients are placed into (k_place_g), instead of sorting every step.
    gn::DevBuf<uint64_t> he_static, pr_static;      // [2 e + 64], [e + 64]
};

namespace {

void bwd_plan_free(gn_distmult_bwd_plan* p) {
    if (!p) return;
    p->eu.release(); p->ev.release(); p->er.release(); p->own.release(); p->mir.release();
    p->offsets.release(); p->he_taskptr.release(); p->pr_taskptr.release(); p->he_tasks.release(); p->pr_tasks.release();
    p->he_static.release(); p->pr_static.release();
    delete p;
}

struct PlanWs { size_t g, he, pr, partial, total; };
PlanWs plan_ws(const gn_distmult_bwd_plan* p, int64_t f) {
    PlanWs w;
    w.g = 0;
    w.he = w.g + align_up((size_t)(p->e + 64) * sizeof(float));
    w.pr = w.he + align_up((size_t)(2 * p->e + 64) * sizeof(uint64_t));
    w.partial = w.pr + align_up((size_t)(2 * p->e + 64) * sizeof(uint64_t));
    w.total = w.partial + align_up((size_t)(p->he_tasks_max + p->pr_tasks_max) * f * sizeof(float));   // both reductions' partial sums (one combine launch)
    return w;
}

}  // namespace

namespace {
// a static edge_type's offsets with the task list of the relation-major reduction behind them (GN_DM_TYPE_TASKS)
struct TypeTasks { size_t taskptr, tasks, total; };           // in int32 words
TypeTasks type_tasks_layout(int64_t r, int64_t e) {
    TypeTasks t;
    t.taskptr = (size_t)((r + 2 + 63) & ~(int64_t)63);
    t.tasks = t.taskptr + (size_t)((r + 2 + 63) & ~(int64_t)63);
    t.total = t.tasks + (size_t)lds_max_tasks(e, r) * 4;
    return t;
}
}  // namespace

extern "C" size_t gn_distmult_type_tasks_bytes(int64_t r, int64_t e) {
    if (r <= 0 || e < 0) return 0;
    return type_tasks_layout(r, e).total * sizeof(int32_t);
}

extern "C" gn_status gn_distmult_type_tasks(const int32_t* type_offsets, int64_t r, int64_t e, void* out, size_t out_bytes, void* stream) {
    GN_REQUIRE(type_offsets && out && r > 0 && e >= 0, "null pointer or bad size");
    GN_REQUIRE(r < (1ll << 31) && e < (1ll << 31), "too large");
    GN_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "the buffer is not 16-byte aligned");
    const TypeTasks t = type_tasks_layout(r, e);
    GN_REQUIRE(out_bytes >= t.total * sizeof(int32_t), "buffer too small: need %zu bytes", t.total * sizeof(int32_t));
    hipStream_t st = gn::as_stream(stream);
    int32_t* o = static_cast<int32_t*>(out);
    if (o != type_offsets) GN_HIP(hipMemcpyAsync(o, type_offsets, (size_t)(r + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
    k_task_ptr<<<1, 1024, 0, st>>>(o, 1, (int)r, o + t.taskptr, reinterpret_cast<int4*>(o + t.tasks));
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" size_t gn_distmult_backward_workspace_bytes(int64_t n, int64_t f, int64_t r, int64_t e) {
    if (n <= 0 || f <= 0 || r <= 0 || e <= 0) return 0;
    return std::max(ws_layout(e, std::max(n, r)).total, lds_layout(e, n, r, f).total);
}

namespace {
gn_status backward_impl(const float* z, int64_t ld_z, int64_t n, int64_t f, const EdgeSrc& edges, const float* d, int64_t ld_d,
                        int64_t r, int64_t e, const float* grad_logit, float* dz, int64_t ld_dz,
                        float* dd, int64_t ld_dd, int flags, const float* sigmoid_scores,
                        const int32_t* type_offsets, void* workspace, size_t workspace_bytes,
                        void* stream, const gn_link_loss_grad* loss = nullptr, const float* dz_add = nullptr, int64_t ld_dz_add = 0,
                        const float* dd_add = nullptr, int64_t ld_dd_add = 0) {
    const int64_t* u = edges.u;
    const int64_t* v = edges.v;
    const int64_t* et = edges.et;
    const bool from_words = edges.packed != nullptr;           // no int64 arrays: the counting-sort / LDS path or nothing
    GN_REQUIRE(n >= 0 && f >= 0 && r >= 0 && e >= 0, "negative size");
    GN_REQUIRE(f < (1ll << 31) && n < (1ll << 31) && r < (1ll << 31) && e < (1ll << 31), "table or edge list too large");
    GN_REQUIRE((n == 0 || f == 0 || dz) && (r == 0 || f == 0 || dd), "gradient output pointer is null");
    GN_REQUIRE(ld_dz >= f && ld_dd >= f, "leading dimension smaller than the row length");
    hipStream_t st = gn::as_stream(stream);
    if (e == 0 || f == 0 || n == 0 || r == 0) {
        if (n > 0 && f > 0) GN_HIP(hipMemset2DAsync(dz, ld_dz * sizeof(float), 0, f * sizeof(float), n, st));
        if (r > 0 && f > 0) GN_HIP(hipMemset2DAsync(dd, ld_dd * sizeof(float), 0, f * sizeof(float), r, st));
        GN_REQUIRE(e == 0 || f == 0 || (n > 0 && r > 0), "edges given but the node or relation table is empty");
        return GN_OK;
    }
    GN_REQUIRE(z && d && (loss ? sigmoid_scores != nullptr : grad_logit != nullptr) && (from_words ? edges.rel16 != nullptr : (u && v && et)), "operand pointer is null");
    const GradSrc grad = make_grad_src(grad_logit, sigmoid_scores, loss, e);
    GN_REQUIRE(ld_z >= f && ld_d >= f, "leading dimension smaller than the row length");
    const WsLayout l = ws_layout(e, std::max(n, r));
    const LdsLayout ll = lds_layout(e, n, r, f);
    GN_REQUIRE(workspace && workspace_bytes >= std::max(l.total, ll.total), "workspace too small: need %zu bytes",
               std::max(l.total, ll.total));
    char* ws = static_cast<char*>(workspace);
    uint32_t* keys = reinterpret_cast<uint32_t*>(ws + l.keys);
    uint32_t* keys_sorted = reinterpret_cast<uint32_t*>(ws + l.keys_sorted);
    Rec* recs = reinterpret_cast<Rec*>(ws + l.recs);
    Rec* recs_sorted = reinterpret_cast<Rec*>(ws + l.recs_sorted);
    int32_t* rowptr = reinterpret_cast<int32_t*>(ws + l.rowptr);
    float* partial = reinterpret_cast<float*>(ws + l.partial);
    const unsigned chunks = (unsigned)gn::ceil_div(e, kChunkRecs);
    size_t sort_bytes = l.total - l.sort_tmp;
    const bool vec = (f % 4 == 0) && (ld_z % 4 == 0) && (ld_d % 4 == 0) && (ld_dz % 4 == 0) && (ld_dd % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(dz) |
                       reinterpret_cast<uintptr_t>(dd)) & 15) == 0;

    const bool lds_shapes = lds_path_shapes(n, f, r) && 2 * e + 64 < (1ll << 31) && (ld_z % 4 == 0) && (ld_d % 4 == 0) && (ld_dz % 4 == 0) && (ld_dd % 4 == 0) &&
                            ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(dz) |
                              reinterpret_cast<uintptr_t>(dd)) & 15) == 0;
    const bool lds_dz = lds_shapes && lds_dz_fits(n, r), lds_dd = lds_shapes && lds_dd_fits(n);
    if (from_words && !(lds_dz && lds_dd && n <= kSortMaxKeys && (flags & GN_DM_TYPES_SORTED) && type_offsets))
        return gn::fail(GN_ERR_UNSUPPORTED, "packed pairs: only the counting-sort path (tables in LDS, sorted edge_type with its offsets); use gn_distmult_backward_ex_f32");
    if (lds_dz || lds_dd) {
        uint32_t* k2 = reinterpret_cast<uint32_t*>(ws + ll.keys);
        uint32_t* k2s = reinterpret_cast<uint32_t*>(ws + ll.keys_sorted);
        uint64_t* r2 = reinterpret_cast<uint64_t*>(ws + ll.recs);
        uint64_t* r2s = reinterpret_cast<uint64_t*>(ws + ll.recs_sorted);
        int32_t* rp = reinterpret_cast<int32_t*>(ws + ll.rowptr);
        int32_t* tp = reinterpret_cast<int32_t*>(ws + ll.taskptr);
        int4* tk = reinterpret_cast<int4*>(ws + ll.tasks);
        const bool sorted_types = (flags & GN_DM_TYPES_SORTED) != 0;
        bool pairs_done = false;
        float* part = reinterpret_cast<float*>(ws + ll.partial);
        float* part_dd = part + (size_t)lds_max_tasks(2 * e, n) * f;      // the relation-major reduction's partial sums
        CombineSet cz, cd;                                                 // both reductions are combined in one launch ...
        // ... when the relation-major one brings its task list (the scratch holds one list: the node-major one's must outlive its combine)
        const bool listed = lds_dd && (flags & GN_DM_TYPES_SORTED) && type_offsets && (flags & GN_DM_TYPE_TASKS);
        const bool both = lds_dz && listed;
        if ((dz_add || dd_add) && !both)
            return gn::fail(GN_ERR_UNSUPPORTED, "addends ride on the common combine launch of the two LDS reductions: not taken for these shapes / flags");
        size_t sort2 = ll.sort_tmp_bytes;
        if (lds_dz && n <= kSortMaxKeys) {
            static thread_local bool sort_configured = false;
            if (!sort_configured) {
                GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_he_sort<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_he_sort<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_he_scatter_staged), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
                sort_configured = true;
            }
            int32_t* counts = reinterpret_cast<int32_t*>(ws + ll.counts);
            const size_t hist_bytes = (size_t)kSortWavesPerWg * n * sizeof(int32_t);
            k_he_sort<false><<<kSortWaves / kSortWavesPerWg, kSortWavesPerWg * 64, hist_bytes, st>>>(edges, grad, e, (int)n, r,
                                                                                                   counts, nullptr, nullptr);
            GN_LAUNCH_CHECK();
            // offsets of every (node, wave) cell + the reduction's task list (the totals sit in the row-offset scratch, unused on this path)
            k_node_totals<<<(unsigned)n, kOffsetThreads, 0, st>>>(counts, rp);
            GN_LAUNCH_CHECK();
            k_he_offsets<<<(unsigned)n, kOffsetThreads, 0, st>>>(counts, rp, (int)n, tp, tk);
            GN_LAUNCH_CHECK();
            pairs_done = lds_dd && sorted_types;               // the dD records come out of the same pass
            const int64_t per_wave = gn::ceil_div(e, kSortWaves);
            const size_t stage_cap = (size_t)2 * per_wave * kSortWavesPerWg;
            const size_t staged_bytes = hist_bytes + (2 * (size_t)n + 2 + 1) * sizeof(int32_t) + stage_cap * sizeof(uint64_t);
            if (staged_bytes <= 127 * 1024) {                  // (78 KB on pose0-syn: two workgroups per CU)
                k_he_scatter_staged<<<kSortWaves / kSortWavesPerWg, kSortWavesPerWg * 64, staged_bytes, st>>>(
                    edges, grad, e, (int)n, r, counts, r2s, pairs_done ? r2 : nullptr, (int)stage_cap);
            } else {
                k_he_sort<true><<<kSortWaves / kSortWavesPerWg, kSortWavesPerWg * 64, hist_bytes, st>>>(edges, grad, e, (int)n, r,
                                                                                                      counts, r2s, pairs_done ? r2 : nullptr);
            }
            GN_LAUNCH_CHECK();
            const gn_status rc = launch_seg_lds(r2s, counts, kSortWaves, tp, tk, n, z, ld_z, n, d, ld_d, r, f, part, dz, ld_dz, st, true,
                                                both ? &cz : nullptr);
            if (rc != GN_OK) return rc;
        } else if (lds_dz) {
            k_half_recs<<<gn::stream_grid(e, 256), 256, 0, st>>>(u, v, et, grad, e, n, r, k2, r2);
            GN_LAUNCH_CHECK();
            GN_HIP(rocprim::radix_sort_pairs(ws + ll.sort_tmp, sort2, k2, k2s, r2, r2s, (size_t)(2 * e), 0, bits_for(n + 1), st));
            k_key_offsets<<<(int)gn::ceil_div(n + 1, 256), 256, 0, st>>>(k2s, 2 * e, (int)n, rp);
            GN_LAUNCH_CHECK();
            const gn_status rc = launch_seg_lds(r2s, rp, 1, tp, tk, n, z, ld_z, n, d, ld_d, r, f, part, dz, ld_dz, st, false, both ? &cz : nullptr);
            if (rc != GN_OK) return rc;
        }
        if (lds_dd) {
            const bool sorted = sorted_types;
            if (!pairs_done) {
                k_pair_recs<<<gn::stream_grid(e, 256), 256, 0, st>>>(u, v, et, grad, e, n, r, sorted ? nullptr : k2, r2);
                GN_LAUNCH_CHECK();
            }
            const uint64_t* recs_dd = r2;
            const int32_t* rp_dd = rp;
            if (sorted && type_offsets) {
                rp_dd = type_offsets;                          // the caller keeps them with its static edge_type
            } else if (sorted) {
                k_key_offsets64<<<(int)gn::ceil_div(r + 1, 256), 256, 0, st>>>(et, e, (int)r, rp);
            } else {
                GN_HIP(rocprim::radix_sort_pairs(ws + ll.sort_tmp, sort2, k2, k2s, r2, r2s, (size_t)e, 0, bits_for(r + 1), st));
                k_key_offsets<<<(int)gn::ceil_div(r + 1, 256), 256, 0, st>>>(k2s, e, (int)r, rp);
                recs_dd = r2s;
            }
            GN_LAUNCH_CHECK();
            int32_t* tp_dd = tp;
            int4* tk_dd = tk;
            if (listed) {
                const TypeTasks t = type_tasks_layout(r, e);
                tp_dd = const_cast<int32_t*>(type_offsets) + t.taskptr;
                tk_dd = reinterpret_cast<int4*>(const_cast<int32_t*>(type_offsets) + t.tasks);
            }
            gn_status rc = launch_seg_lds(recs_dd, rp_dd, 1, tp_dd, tk_dd, r, z, ld_z, n, z, ld_z, n, f, part_dd, dd, ld_dd, st, listed,
                                          both ? &cd : nullptr);
            if (rc == GN_OK && both) {
                cz.add = dz_add; cz.ld_add = ld_dz_add; cd.add = dd_add; cd.ld_add = ld_dd_add;
                rc = combine_both(cz, n, cd, r, f, st);
            }
            if (rc != GN_OK) return rc;
        }
    } else if (dz_add || dd_add) {
        return gn::fail(GN_ERR_UNSUPPORTED, "addends ride on the common combine launch of the two LDS reductions: not taken for these shapes");
    }
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 ? lds_dd : lds_dz) continue;       // done above
        const int64_t rows = mode == 2 ? r : n;
        const float* A = z;                              // a is always a node id
        const int64_t ld_a = ld_z;
        const float* B = mode == 2 ? z : d;              // b: relation row of D, or the other endpoint's z row
        const int64_t ld_b = mode == 2 ? ld_z : ld_d;
        float* out = mode == 2 ? dd : dz;
        const int64_t ld_out = mode == 2 ? ld_dd : ld_dz;
        const int accumulate = mode == 1;                // dz = u-side pass, then + v-side pass
        k_make_recs<<<gn::stream_grid(e, 256), 256, 0, st>>>(mode, u, v, et, grad, e, n, r, (uint32_t)rows, keys, recs);
        GN_LAUNCH_CHECK();
        GN_HIP(rocprim::radix_sort_pairs(ws + l.sort_tmp, sort_bytes, keys, keys_sorted, recs, recs_sorted, (size_t)e, 0,
                                         bits_for(rows + 1), st));
        k_key_offsets<<<(int)gn::ceil_div(rows + 1, 256), 256, 0, st>>>(keys_sorted, e, (int)rows, rowptr);
        GN_LAUNCH_CHECK();
        if (vec) {
            for (int c0 = 0; c0 < f; c0 += 64) {
                const int width = (int)std::min<int64_t>(64, f - c0);
                if (width > 32)
                    k_seg_reduce<16><<<chunks, 256, 0, st>>>(rowptr, keys_sorted, recs_sorted, e, (int)rows, A, ld_a, B, ld_b,
                                                             out, ld_out, partial, c0, width, accumulate);
                else if (width > 16)
                    k_seg_reduce<8><<<chunks, 256, 0, st>>>(rowptr, keys_sorted, recs_sorted, e, (int)rows, A, ld_a, B, ld_b,
                                                            out, ld_out, partial, c0, width, accumulate);
                else
                    k_seg_reduce<4><<<chunks, 256, 0, st>>>(rowptr, keys_sorted, recs_sorted, e, (int)rows, A, ld_a, B, ld_b,
                                                            out, ld_out, partial, c0, width, accumulate);
                GN_LAUNCH_CHECK();
                k_seg_combine<<<(unsigned)rows, 64, 0, st>>>(rowptr, keys_sorted, e, (int)rows, partial, out, ld_out, c0, width,
                                                            accumulate);
                GN_LAUNCH_CHECK();
            }
        } else {
            k_seg_reduce_scalar<<<(unsigned)rows, 256, 0, st>>>(rowptr, recs_sorted, A, ld_a, B, ld_b, out, ld_out, (int)f, accumulate);
            GN_LAUNCH_CHECK();
        }
    }
    return GN_OK;
}
}  // namespace

extern "C" gn_status gn_distmult_backward_ex_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                                 const int64_t* v, const int64_t* et, const float* d, int64_t ld_d,
                                                 int64_t r, int64_t e, const float* grad_logit, float* dz, int64_t ld_dz,
                                                 float* dd, int64_t ld_dd, int flags, const float* sigmoid_scores,
                                                 const int32_t* type_offsets, void* workspace, size_t workspace_bytes,
                                                 void* stream) {
    return backward_impl(z, ld_z, n, f, EdgeSrc{u, v, et, nullptr, nullptr}, d, ld_d, r, e, grad_logit, dz, ld_dz, dd, ld_dd, flags,
                         sigmoid_scores, type_offsets, workspace, workspace_bytes, stream);
}

extern "C" gn_status gn_distmult_backward_packed_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const uint32_t* packed_uv,
                                                     const uint16_t* rel16, const float* d, int64_t ld_d, int64_t r, int64_t e,
                                                     const float* grad_logit, float* dz, int64_t ld_dz, float* dd, int64_t ld_dd,
                                                     int flags, const float* sigmoid_scores, const int32_t* type_offsets,
                                                     void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(e == 0 || (packed_uv && rel16), "packed pairs or relation ids are null");
    GN_REQUIRE(n <= 65536 && r <= 65536, "packed pairs hold ids of 16 bits");
    return backward_impl(z, ld_z, n, f, EdgeSrc{nullptr, nullptr, nullptr, packed_uv, rel16}, d, ld_d, r, e, grad_logit, dz, ld_dz, dd,
                         ld_dd, flags, sigmoid_scores, type_offsets, workspace, workspace_bytes, stream);
}

extern "C" gn_status gn_distmult_backward_loss_packed_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const uint32_t* packed_uv,
                                                          const uint16_t* rel16, const float* d, int64_t ld_d, int64_t r, int64_t e,
                                                          const gn_link_loss_grad* loss, const float* sigmoid_scores, float* dz, int64_t ld_dz,
                                                          float* dd, int64_t ld_dd, int flags, const int32_t* type_offsets,
                                                          const float* dz_add, int64_t ld_dz_add, const float* dd_add, int64_t ld_dd_add,
                                                          void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(loss != nullptr && (e == 0 || sigmoid_scores), "the loss source and the forward's probabilities are required");
    GN_REQUIRE((!dz_add || (ld_dz_add >= f && ld_dz_add % 4 == 0 && (reinterpret_cast<uintptr_t>(dz_add) & 15) == 0)) &&
               (!dd_add || (ld_dd_add >= f && ld_dd_add % 4 == 0 && (reinterpret_cast<uintptr_t>(dd_add) & 15) == 0)) && (e > 0 || (!dz_add && !dd_add)),
               "addends: 16-byte aligned rows of at least num_features floats, and a non-empty list");
    GN_REQUIRE(e == 0 || (packed_uv && rel16), "packed pairs or relation ids are null");
    GN_REQUIRE(n <= 65536 && r <= 65536, "packed pairs hold ids of 16 bits");
    return backward_impl(z, ld_z, n, f, EdgeSrc{nullptr, nullptr, nullptr, packed_uv, rel16}, d, ld_d, r, e, nullptr, dz, ld_dz, dd,
                         ld_dd, flags, sigmoid_scores, type_offsets, workspace, workspace_bytes, stream, loss, dz_add, ld_dz_add, dd_add, ld_dd_add);
}

extern "C" gn_status gn_distmult_backward_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                              const int64_t* v, const int64_t* et, const float* d, int64_t ld_d,
                                              int64_t r, int64_t e, const float* grad_logit, float* dz, int64_t ld_dz,
                                              float* dd, int64_t ld_dd, void* workspace, size_t workspace_bytes,
                                              void* stream) {
    return gn_distmult_backward_ex_f32(z, ld_z, n, f, u, v, et, d, ld_d, r, e, grad_logit, dz, ld_dz, dd, ld_dd, 0, nullptr, nullptr, workspace,
                                       workspace_bytes, stream);
}

namespace {

constexpr uint32_t kNoPair = 0xffffffffu;

// g of a pair = the sum of its two triples' gradients (each with its own sigmoid factor); own == null: no pairing, triple i
// is pair i.  The pair's record of the dD pass (list order) is completed in the same pass.
__global__ void k_pair_grad(const uint32_t* __restrict__ own, const uint32_t* __restrict__ mir, int64_t n, GradSrc gs,
                            float* __restrict__ out, const uint64_t* __restrict__ pr_static, uint64_t* __restrict__ pr) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t rec = pr_static[i];
        float g;
        if (own) {
            const uint32_t m = mir[i];
            g = gs.at(own[i]);
            if (m != kNoPair) g += gs.at(m);
        } else {
            g = gs.at(i);
        }
        out[i] = g;
        pr[i] = (rec & 0xffffffffull) | ((uint64_t)__float_as_uint(g) << 32);
    }
}

// he[i] = (the static half of record i, the gradient of the pair it came from): four records per thread and trip, the
// gathers of all four in flight
__global__ __launch_bounds__(256) void k_place_g(const uint64_t* __restrict__ he_static, int64_t n, const float* __restrict__ g,
                                                 uint64_t* __restrict__ he) {
    const int64_t step = (int64_t)gridDim.x * 256;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += 4 * step) {
        uint64_t rec[4];
        float gv[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rec[k] = he_static[min(i + k * step, n - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) gv[k] = g[rec[k] >> 32];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * step < n) he[i + k * step] = (rec[k] & 0xffffffffull) | ((uint64_t)__float_as_uint(gv[k]) << 32);
    }
}

__global__ void k_is_sorted64(const int64_t* __restrict__ x, int64_t n, int* __restrict__ unsorted) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i + 1 < n; i += (int64_t)gridDim.x * blockDim.x)
        if (x[i] > x[i + 1]) *unsorted = 1;
}

struct TmpBufs {
    std::vector<void*> ptrs;
    ~TmpBufs() { for (void* q : ptrs) (void)hipFree(q); }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        void* q = nullptr;
        hipError_t err = hipMalloc(&q, (count ? count : 1) * sizeof(T));
        if (err == hipSuccess) ptrs.push_back(q);
        *out = static_cast<T*>(q);
        return err;
    }
};

// The records of the plan's triples, sorted once: the counting sort's scatter pass with the triple's position in place of
// its gradient (GradSrc index mode).  A step then only places its gradients (k_pair_grad, k_place_g).
gn_status place_static_records(gn_distmult_bwd_plan* p, const int64_t* u, const int64_t* v, const int64_t* et, TmpBufs& tmp, hipStream_t st) {
    const int64_t E = p->e, n = p->n, R = p->r;
    GN_HIP(p->he_static.alloc((size_t)(2 * E + 64)));
    GN_HIP(p->pr_static.alloc((size_t)(2 * E + 64)));          // (the scatter's spare slots sit at 2 E)
    const GradSrc index = {nullptr, nullptr};
    const size_t hist_bytes = (size_t)kSortWavesPerWg * n * sizeof(int32_t);
    const int64_t per_wave = gn::ceil_div(E, kSortWaves);
    const size_t stage_cap = (size_t)2 * per_wave * kSortWavesPerWg;
    const size_t staged_bytes = hist_bytes + (2 * (size_t)n + 2 + 1) * sizeof(int32_t) + stage_cap * sizeof(uint64_t);
    if (staged_bytes <= 127 * 1024) {
        { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_he_scatter_staged), 128 * 1024); if (lds_status != GN_OK) return lds_status; }
        k_he_scatter_staged<<<kSortWaves / kSortWavesPerWg, kSortWavesPerWg * 64, staged_bytes, st>>>(EdgeSrc{u, v, et, nullptr, nullptr}, index, E, (int)n, R, p->offsets.p,
                                                                                                  p->he_static.p, p->pr_static.p, (int)stage_cap);
    } else {
        // the unstaged pass advances its offsets in place: it works on a copy
        const size_t cells = (size_t)n * kSortWaves + 1;
        int32_t* copy;
        GN_HIP(tmp.get(&copy, cells));
        GN_HIP(hipMemcpyAsync(copy, p->offsets.p, cells * sizeof(int32_t), hipMemcpyDeviceToDevice, st));
        { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_he_sort<true>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
        k_he_sort<true><<<kSortWaves / kSortWavesPerWg, kSortWavesPerWg * 64, hist_bytes, st>>>(EdgeSrc{u, v, et, nullptr, nullptr}, index, E, (int)n, R, copy, p->he_static.p,
                                                                                              p->pr_static.p);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// The node-major records of a static list, batch by batch of their tasks, in the bank-balanced order (host_layout.hpp): the
// partner's row of z and the relation's row of D have four different indices mod 4 in every lane group where the batch allows it.
gn_status balance_static_records(gn_distmult_bwd_plan* p, hipStream_t st) {
    const int64_t n_rec = 2 * p->e;
    int32_t n_tasks = 0;
    GN_HIP(hipMemcpyAsync(&n_tasks, p->he_taskptr.p + p->n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    if (n_tasks <= 0 || n_rec <= 0) return GN_OK;
    std::vector<uint64_t> recs((size_t)n_rec);
    std::vector<int32_t> tasks((size_t)n_tasks * 4);
    GN_HIP(hipMemcpyAsync(recs.data(), p->he_static.p, recs.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipMemcpyAsync(tasks.data(), p->he_tasks.p, tasks.size() * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    uint64_t tmp[64];
    for (int32_t t = 0; t < n_tasks; ++t) {
        const int64_t begin = tasks[(size_t)t * 4 + 1], end = tasks[(size_t)t * 4 + 2];
        if (begin < 0 || end > n_rec) return gn::fail(GN_ERR_INVALID_ARG, "task %d of the decoder gradient plan lies outside its records", (int)t);
        for (int64_t b0 = begin; b0 + 64 <= end; b0 += 64) {
            uint8_t cls[64];
            int order[64];
            for (int i = 0; i < 64; ++i) {
                const uint32_t w = (uint32_t)recs[(size_t)(b0 + i)];
                cls[i] = (uint8_t)((w & 3u) | (((w >> 16) & 3u) << 2));
            }
            gn_layout::balance_batch64(cls, order);
            for (int i = 0; i < 64; ++i) tmp[i] = recs[(size_t)(b0 + order[i])];
            for (int i = 0; i < 64; ++i) recs[(size_t)(b0 + i)] = tmp[i];
        }
    }
    GN_HIP(hipMemcpyAsync(p->he_static.p, recs.data(), recs.size() * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));
    return GN_OK;
}

gn_status build_bwd_plan(gn_distmult_bwd_plan* p, const int64_t* u, const int64_t* v, const int64_t* et, hipStream_t st) {
    const int64_t E = p->e, n = p->n, R = p->r;
    TmpBufs tmp;
    int32_t* rp;
    int* unsorted;
    const size_t cells = (size_t)n * kSortWaves + 1;
    size_t scan_bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, scan_bytes, (int32_t*)nullptr, (int32_t*)nullptr, 0, cells, rocprim::plus<int32_t>(), (hipStream_t)0);
    char* scratch;
    GN_HIP(tmp.get(&rp, (size_t)R + 2));
    GN_HIP(tmp.get(&unsorted, 1));
    GN_HIP(tmp.get(&scratch, scan_bytes));
    GN_HIP(hipMemsetAsync(unsorted, 0, sizeof(int), st));
    k_is_sorted64<<<gn::stream_grid(E, 256), 256, 0, st>>>(et, E, unsorted);
    GN_LAUNCH_CHECK();
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_he_sort<false>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    GN_HIP(p->offsets.alloc(cells));
    const GradSrc none = {nullptr, nullptr};
    const size_t hist_bytes = (size_t)kSortWavesPerWg * n * sizeof(int32_t);
    k_he_sort<false><<<kSortWaves / kSortWavesPerWg, kSortWavesPerWg * 64, hist_bytes, st>>>(EdgeSrc{u, v, et, nullptr, nullptr}, none, E, (int)n, R, p->offsets.p, nullptr,
                                                                                           nullptr);
    GN_LAUNCH_CHECK();
    GN_HIP(rocprim::exclusive_scan(scratch, scan_bytes, p->offsets.p, p->offsets.p, 0, cells, rocprim::plus<int32_t>(), st));
    int32_t placed = 0;
    int is_unsorted = 0;
    GN_HIP(hipMemcpyAsync(&placed, p->offsets.p + cells - 1, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipMemcpyAsync(&is_unsorted, unsorted, sizeof(int), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    if ((int64_t)placed != 2 * E)
        return gn::fail(GN_ERR_INDEX_RANGE, "%lld of the %lld edges have a node or relation id outside its table",
                        (long long)(E - placed / 2), (long long)E);
    if (is_unsorted)
        return gn::fail(GN_ERR_UNSUPPORTED, "edge_type is not sorted: the dD records need a sort per call (gn_distmult_backward_f32)");
    p->he_tasks_max = lds_max_tasks(2 * E, n);
    p->pr_tasks_max = lds_max_tasks(E, R);
    GN_HIP(p->he_taskptr.alloc((size_t)n + 2));
    GN_HIP(p->pr_taskptr.alloc((size_t)R + 2));
    GN_HIP(p->he_tasks.alloc((size_t)p->he_tasks_max * 4));
    GN_HIP(p->pr_tasks.alloc((size_t)p->pr_tasks_max * 4));
    k_task_ptr<<<1, 1024, 0, st>>>(p->offsets.p, (int64_t)kSortWaves, (int)n, p->he_taskptr.p, reinterpret_cast<int4*>(p->he_tasks.p));
    GN_LAUNCH_CHECK();
    k_key_offsets64<<<(int)gn::ceil_div(R + 1, 256), 256, 0, st>>>(et, E, (int)R, rp);
    GN_LAUNCH_CHECK();
    k_task_ptr<<<1, 1024, 0, st>>>(rp, 1, (int)R, p->pr_taskptr.p, reinterpret_cast<int4*>(p->pr_tasks.p));
    GN_LAUNCH_CHECK();
    const gn_status rc = place_static_records(p, u, v, et, tmp, st);
    if (rc != GN_OK) return rc;
    GN_HIP(hipStreamSynchronize(st));       // scratch goes out of scope
    return GN_OK;
}

}  // namespace

extern "C" gn_status gn_distmult_bwd_plan_create(const int64_t* u, const int64_t* v, const int64_t* edge_type, int64_t num_edges,
                                                 int64_t num_nodes, int64_t num_relations, void* stream, gn_distmult_bwd_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(num_edges >= 0 && num_nodes >= 0 && num_relations >= 0, "negative size");
    GN_REQUIRE(num_edges == 0 || (u && v && edge_type), "edge pointers are null");
    if (gn::fast_paths_disabled() || num_nodes < 1 || num_relations < 1 || num_nodes > kSortMaxKeys || num_relations > 65535 ||
        !lds_dz_fits(num_nodes, num_relations) || 2 * num_edges + 64 >= (1ll << 31))
        return gn::fail(GN_ERR_UNSUPPORTED, "node and relation tables do not fit the LDS path (or it is disabled): use gn_distmult_backward_f32");
    hipStream_t st = gn::as_stream(stream);
    const int64_t E = num_edges;
    // pair up the two directions of an edge on the host (once per static list)
    std::vector<int64_t> hu(E), hv(E), hr(E);
    if (E > 0) {
        GN_HIP(hipMemcpyAsync(hu.data(), u, E * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipMemcpyAsync(hv.data(), v, E * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipMemcpyAsync(hr.data(), edge_type, E * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipStreamSynchronize(st));
    }
    for (int64_t e = 0; e < E; ++e) {
        if ((uint64_t)hu[e] >= (uint64_t)num_nodes || (uint64_t)hv[e] >= (uint64_t)num_nodes || (uint64_t)hr[e] >= (uint64_t)num_relations)
            return gn::fail(GN_ERR_INDEX_RANGE, "edge %lld = (%lld, %lld, type %lld) is outside [0,%lld) x [0,%lld) x [0,%lld)",
                            (long long)e, (long long)hu[e], (long long)hv[e], (long long)hr[e], (long long)num_nodes,
                            (long long)num_nodes, (long long)num_relations);
        if (e > 0 && hr[e] < hr[e - 1])
            return gn::fail(GN_ERR_UNSUPPORTED, "edge_type is not sorted: the dD records need a sort per call (gn_distmult_backward_f32)");
    }
    std::vector<int64_t> eu, ev, er;
    std::vector<uint32_t> own, mir;
    {
        std::unordered_map<uint64_t, int64_t> open;       // key -> index into eu of a triple still without a partner
        open.reserve((size_t)E);
        for (int64_t e = 0; e < E; ++e) {
            const uint64_t lo = (uint64_t)std::min(hu[e], hv[e]), hi = (uint64_t)std::max(hu[e], hv[e]);
            const uint64_t key = ((uint64_t)hr[e] << 32) | (lo << 16) | hi;      // ids < 65536 on this path
            auto it = open.find(key);
            if (it != open.end()) {
                mir[(size_t)it->second] = (uint32_t)e;
                open.erase(it);
            } else {
                open.emplace(key, (int64_t)eu.size());
                eu.push_back(hu[e]); ev.push_back(hv[e]); er.push_back(hr[e]);
                own.push_back((uint32_t)e); mir.push_back(kNoPair);
            }
        }
    }
    // The pairs of a relation may stand in any order: inside every full 64-record batch of the relation-major reduction's tasks
    // they are placed so that the four rows of an LDS lane group have four different indices mod 4, for u and for v (host_layout.hpp)
    {
        const int64_t P = (int64_t)eu.size();
        std::vector<int64_t> tu(64), tv(64);
        std::vector<uint32_t> to(64), tm(64);
        for (int64_t s0 = 0; s0 < P;) {
            int64_t s1 = s0;
            while (s1 < P && er[(size_t)s1] == er[(size_t)s0]) ++s1;
            for (int64_t t0 = s0; t0 < s1; t0 += kTaskRecs)
                for (int64_t b0 = t0; b0 + 64 <= std::min(s1, t0 + kTaskRecs); b0 += 64) {
                    uint8_t cls[64];
                    int order[64];
                    for (int i = 0; i < 64; ++i) cls[i] = (uint8_t)((eu[(size_t)(b0 + i)] & 3) | ((ev[(size_t)(b0 + i)] & 3) << 2));
                    gn_layout::balance_batch64(cls, order);
                    for (int i = 0; i < 64; ++i) {
                        const size_t from = (size_t)(b0 + order[i]);
                        tu[i] = eu[from]; tv[i] = ev[from]; to[i] = own[from]; tm[i] = mir[from];
                    }
                    for (int i = 0; i < 64; ++i) {
                        const size_t at = (size_t)(b0 + i);
                        eu[at] = tu[i]; ev[at] = tv[i]; own[at] = to[i]; mir[at] = tm[i];
                    }
                }
            s0 = s1;
        }
    }
    gn_distmult_bwd_plan* p = new (std::nothrow) gn_distmult_bwd_plan();
    GN_REQUIRE(p != nullptr, "out of host memory");
    p->e_list = E; p->e = (int64_t)eu.size(); p->n = num_nodes; p->r = num_relations;
    if (p->e > 0) {
        auto up = [&](auto& buf, const auto& host) -> hipError_t {
            hipError_t err = buf.alloc(host.size());
            if (err != hipSuccess) return err;
            return hipMemcpyAsync(buf.p, host.data(), host.size() * sizeof(host[0]), hipMemcpyHostToDevice, st);
        };
        hipError_t err = hipSuccess;
        if ((err = up(p->eu, eu)) != hipSuccess || (err = up(p->ev, ev)) != hipSuccess || (err = up(p->er, er)) != hipSuccess ||
            (err = up(p->own, own)) != hipSuccess || (err = up(p->mir, mir)) != hipSuccess ||
            (err = hipStreamSynchronize(st)) != hipSuccess) {
            bwd_plan_free(p);
            return gn::fail(GN_ERR_HIP, "decoder gradient plan upload failed: %s", hipGetErrorString(err));
        }
        gn_status rc = build_bwd_plan(p, p->eu.p, p->ev.p, p->er.p, st);
        if (rc == GN_OK) rc = balance_static_records(p, st);
        if (rc != GN_OK) { bwd_plan_free(p); return rc; }
    }
    *out = p;
    return GN_OK;
}

extern "C" void gn_distmult_bwd_plan_destroy(gn_distmult_bwd_plan* plan) { bwd_plan_free(plan); }

extern "C" size_t gn_distmult_bwd_plan_workspace_bytes(const gn_distmult_bwd_plan* plan, int64_t num_features) {
    if (!plan || num_features <= 0 || plan->e == 0) return 0;
    return plan_ws(plan, num_features).total;
}

static gn_status backward_planned_impl(const gn_distmult_bwd_plan* plan, const float* z, int64_t ld_z, int64_t f,
                                       const float* d, int64_t ld_d, const float* grad_logit,
                                       const float* sigmoid_scores, float* dz, int64_t ld_dz, float* dd,
                                       int64_t ld_dd, void* workspace, size_t workspace_bytes, void* stream, const gn_link_loss_grad* loss);

extern "C" gn_status gn_distmult_backward_planned_f32(const gn_distmult_bwd_plan* plan, const float* z, int64_t ld_z, int64_t f,
                                                      const float* d, int64_t ld_d, const float* grad_logit,
                                                      const float* sigmoid_scores, float* dz, int64_t ld_dz, float* dd,
                                                      int64_t ld_dd, void* workspace, size_t workspace_bytes, void* stream) {
    return backward_planned_impl(plan, z, ld_z, f, d, ld_d, grad_logit, sigmoid_scores, dz, ld_dz, dd, ld_dd, workspace, workspace_bytes, stream, nullptr);
}

extern "C" gn_status gn_distmult_backward_loss_planned_f32(const gn_distmult_bwd_plan* plan, const float* z, int64_t ld_z, int64_t f,
                                                           const float* d, int64_t ld_d, const gn_link_loss_grad* loss,
                                                           const float* sigmoid_scores, float* dz, int64_t ld_dz, float* dd,
                                                           int64_t ld_dd, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(loss != nullptr && sigmoid_scores != nullptr, "the loss source and the forward's probabilities are required");
    return backward_planned_impl(plan, z, ld_z, f, d, ld_d, nullptr, sigmoid_scores, dz, ld_dz, dd, ld_dd, workspace, workspace_bytes, stream, loss);
}

static gn_status backward_planned_impl(const gn_distmult_bwd_plan* plan, const float* z, int64_t ld_z, int64_t f,
                                       const float* d, int64_t ld_d, const float* grad_logit,
                                       const float* sigmoid_scores, float* dz, int64_t ld_dz, float* dd,
                                       int64_t ld_dd, void* workspace, size_t workspace_bytes, void* stream, const gn_link_loss_grad* loss) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(f >= 0 && f < (1ll << 31), "bad feature count");
    GN_REQUIRE(f == 0 || (dz && dd && ld_dz >= f && ld_dd >= f), "gradient output pointer is null or its leading dimension too small");
    hipStream_t st = gn::as_stream(stream);
    const int64_t n = plan->n, r = plan->r, e = plan->e;
    if (f == 0) return GN_OK;
    if (e == 0) {
        GN_HIP(hipMemset2DAsync(dz, ld_dz * sizeof(float), 0, f * sizeof(float), n, st));
        GN_HIP(hipMemset2DAsync(dd, ld_dd * sizeof(float), 0, f * sizeof(float), r, st));
        return GN_OK;
    }
    GN_REQUIRE(z && d && (loss || grad_logit) && ld_z >= f && ld_d >= f, "operand pointer is null or a leading dimension too small");
    if (f % 4 != 0 || ld_z % 4 != 0 || ld_d % 4 != 0 || ld_dz % 4 != 0 || ld_dd % 4 != 0 ||
        ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(dd)) & 15) != 0)
        return gn::fail(GN_ERR_UNSUPPORTED, "rows are not 16-byte aligned float4 columns: use gn_distmult_backward_f32");
    const PlanWs w = plan_ws(plan, f);
    GN_REQUIRE(workspace && workspace_bytes >= w.total, "workspace too small: need %zu bytes", w.total);
    char* ws = static_cast<char*>(workspace);
    uint64_t* he = reinterpret_cast<uint64_t*>(ws + w.he);
    uint64_t* pr = reinterpret_cast<uint64_t*>(ws + w.pr);
    float* part = reinterpret_cast<float*>(ws + w.partial);
    float* gpair = reinterpret_cast<float*>(ws + w.g);
    k_pair_grad<<<gn::stream_grid(e, 256), 256, 0, st>>>(plan->own.p, plan->mir.p, e, make_grad_src(grad_logit, sigmoid_scores, loss, plan->e_list), gpair, plan->pr_static.p, pr);
    GN_LAUNCH_CHECK();
    k_place_g<<<(unsigned)std::min<int64_t>(gn::ceil_div(2 * e, 4 * 256), 4096), 256, 0, st>>>(plan->he_static.p, 2 * e, gpair, he);
    GN_LAUNCH_CHECK();
    CombineSet cz, cd;
    gn_status rc = launch_seg_lds(he, nullptr, 0, plan->he_taskptr.p, reinterpret_cast<int4*>(plan->he_tasks.p), n, z, ld_z, n, d, ld_d, r, f,
                                  part, dz, ld_dz, st, true, &cz);
    if (rc != GN_OK) return rc;
    rc = launch_seg_lds(pr, nullptr, 0, plan->pr_taskptr.p, reinterpret_cast<int4*>(plan->pr_tasks.p), r, z, ld_z, n, z, ld_z, n, f,
                        part + (size_t)plan->he_tasks_max * f, dd, ld_dd, st, true, &cd);
    if (rc != GN_OK) return rc;
    return combine_both(cz, n, cd, r, f, st);
}
