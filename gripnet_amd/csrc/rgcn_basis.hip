// Multi-relational internal layer (myRGCN, gripnet/layers.py:165-197) for LARGE supervertices: O(E) memory, basis space.
//
//   out[i] * deg_i = sum_{e: dst=i} x[src_e] W_{r(e)},   W_r = sum_b att[r,b] basis[b]                (layers.py:172-189)
//                  = sum_b U_i[b,:] basis[b],             U_i[b,:] = sum_{e: dst=i} att[r(e),b] x[src_e,:]
//
// The reference forms W_r and multiplies per relation slice (O(E) memory); the table form H[r] = X W_r of rgcn.hip's
// first general path needs [R, N, out] floats (2.4 GB and 80 GFLOP per layer for the all-nodes baseline
// baselines/LP_baselines/rgcn_pose.py:53-77 with N = 19,726, R = 964).  Here neither W_r nor H exists: a wave owns a
// destination row; per FOUR incoming edges it issues the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 with
// A = att[r(e), bases] (one float per lane: M = bases) and B = x[src_e, features] (N = features), K = the four edges,
// so that U_i (bases x in) accumulates in registers over the row's edges - every product and sum is an fp32 FMA, the
// same arithmetic as the reference's fp32 matmuls.  The row leaves as one line  [ U_i / max(1, deg_i) | x_i | 0 ]  of a
// slab of rows, and ONE dense product per slab with [basis ; root] (gn_gemm_f32: K = (B + 1) in, fp32-faithful) adds
// the root term, the bias and the activation.  Workspace: the slab (<= 256 MB) + the stacked weights, independent of
// R and N; the edges come from the plan's destination-major list (key = relation * N + source), 4 bytes per edge.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
// loads of a lane's NT consecutive columns: dword-aligned only when NT is odd (global loads need no more)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

// (make VARIANT=nomfma VFLAGS=-DGN_BASIS_NO_MFMA: the same launches with every matrix instruction replaced by one FMA that keeps
// its operands' loads alive - wrong results, for timing what the gathers cost on their own: profiles/r06_experiments.md)
#ifdef GN_BASIS_NO_MFMA
__device__ __forceinline__ f32x4 basis_mfma(float a, float b, f32x4 c) { c[0] += a * b; return c; }
#else
__device__ __forceinline__ f32x4 basis_mfma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
#endif

constexpr int kBasisThreads = 512, kBasisWaves = kBasisThreads / 64;   // eight waves: a destination row per wave at a time, or a heavy row per workgroup
constexpr int kHeavyEdges = gn_layout::kBasisHeavyEdges;               // rows with more incoming edges are walked by a whole workgroup
constexpr int64_t kSlabBytes = 256ll << 20;        // rows of U in flight between the gather and the dense product (round 6: 64 MB cut the all-nodes
                                                   // baseline's first layer - 19,726 rows of 4.3 KB - into two slabs: two gather launches that each
                                                   // drew every work item of all rows, two products)

struct BasisArgs {
    const float* x; int64_t ld_x; int fin;
    const float* att; int bases;
    const int32_t* rowptr; const uint32_t* key; const float* indeg; const int32_t* order;
    const uint32_t* skey;                          // the rows' edges as source * R + relation, sorted by (source, relation) inside a row; or null
    uint32_t R;
    int mode;                                      // 0: per edge, relation order (round 5); 1: pair sums on the hub rows; 2: per edge, source order
    uint32_t n;                                    // nodes (key = relation * n + source)
    int n_rows;                                    // entries of `order` (all rows; with several slabs: the slab's own rows, its heavy ones first)
    const int32_t* slab_heavy;                     // (several slabs) the slab's heavy rows, counted on the device; or null: n_heavy
    int row_lo, row_hi;                            // this launch's slab of destination rows
    float* u; int64_t ld_u;                        // [row_hi - row_lo][ld_u]: U_i (bases outermost) | x_i | zeros up to kp
    int kp;
    int scale, with_x, vec;
    int fast_addr;                                 // vec, and every byte offset into x and att fits 32 bits
    int n_heavy;                                   // rows of more than kHeavyEdges edges: the first entries of `order`
    unsigned int* next_item;                       // this launch's work counter (zeroed by k_basis_weights)
    gn_side_copy side;
};

// One chunk of up to 64 edges, any shape: lane L holds edge L's (source, relation); step j contracts edges 4 j + kg.
template <int BT, int NT>
__device__ __forceinline__ void chunk_any(const BasisArgs& a, f32x4 (&acc)[BT][NT], uint32_t src, uint32_t rel, int cnt, int c, int kg) {
    const int steps = (cnt + 3) >> 2;
    for (int j0 = 0; j0 < steps; j0 += 4) {
        float av[4][BT], xv[4][NT];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int sel = 4 * (j0 + q) + kg;                                  // edge of lane group kg in step j0 + q
            const uint32_t s = (uint32_t)__shfl((int)src, sel & 63), r = (uint32_t)__shfl((int)rel, sel & 63);
            const bool live = sel < cnt;
            const float* __restrict__ xr = a.x + (size_t)s * a.ld_x + NT * c;
            const float* __restrict__ ar = a.att + (size_t)r * a.bases + c;
#pragma unroll
            for (int t = 0; t < NT; ++t) xv[q][t] = (live && NT * c + t < a.fin) ? xr[t] : 0.f;
#pragma unroll
            for (int jm = 0; jm < BT; ++jm) av[q][jm] = (live && 16 * jm + c < a.bases) ? ar[16 * jm] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int jm = 0; jm < BT; ++jm)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[jm][t] = basis_mfma(av[q][jm], xv[q][t], acc[jm][t]);
    }
}

// A FULL chunk of 64 edges on the fast addressing (input width a multiple of 16, rows 16-byte aligned, tables below 4 GB):
// lane L turns its edge into two 32-bit byte offsets once; a step is two ds_bpermute (the offsets of lane group kg's edge,
// lane and step folded into the instruction's immediate), two adds of this lane's column offset, the loads in the
// scalar-base form and the matrix instructions - sixteen steps unrolled, four steps' loads requested together.
template <int BT, int NT>
__device__ __forceinline__ void chunk_full(const BasisArgs& a, f32x4 (&acc)[BT][NT], uint32_t xo, uint32_t ao, uint32_t sel0,
                                           uint32_t lane_x, const uint32_t (&lane_a)[BT], const bool (&base_ok)[BT]) {
    const char* __restrict__ xb = reinterpret_cast<const char*>(a.x);
    const char* __restrict__ ab = reinterpret_cast<const char*>(a.att);
    // steps whose operands are requested together: eight while they fit the registers next to the accumulators (a row's
    // time is a chain of memory round trips - few rows are in flight per compute unit - so fewer, fatter requests win)
#ifndef GN_BASIS_G16_LIMIT
#define GN_BASIS_G16_LIMIT 0
#endif
    constexpr int G = (BT + NT) * 16 + BT * NT * 4 <= GN_BASIS_G16_LIMIT ? 16 : ((BT + NT) * 8 + BT * NT * 4 <= 72 ? 8 : 4);
#pragma unroll
    for (int g = 0; g < 16 / G; ++g) {
        float av[G][BT], xv[G][NT];
#pragma unroll
        for (int q = 0; q < G; ++q) {
            const uint32_t addr = sel0 + 16u * (uint32_t)(G * g + q);           // byte address of lane 4 j + kg
            const uint32_t xs = (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)xo) + lane_x;
            const uint32_t as = (uint32_t)__builtin_amdgcn_ds_bpermute((int)addr, (int)ao);
#pragma unroll
            for (int t4 = 0; t4 < NT / 4; ++t4) {
                const f32x4 v = *reinterpret_cast<const f32x4u*>(xb + (xs + 16u * t4));
                xv[q][4 * t4] = v[0]; xv[q][4 * t4 + 1] = v[1]; xv[q][4 * t4 + 2] = v[2]; xv[q][4 * t4 + 3] = v[3];
            }
            if constexpr (NT % 4 == 2) {
                const f32x2 v = *reinterpret_cast<const f32x2u*>(xb + (xs + 4u * (NT - 2)));
                xv[q][NT - 2] = v[0]; xv[q][NT - 1] = v[1];
            } else if constexpr (NT % 4 == 1) {
                xv[q][NT - 1] = *reinterpret_cast<const float*>(xb + (xs + 4u * (NT - 1)));
            } else if constexpr (NT % 4 == 3) {
                const f32x2 v = *reinterpret_cast<const f32x2u*>(xb + (xs + 4u * (NT - 3)));
                xv[q][NT - 3] = v[0]; xv[q][NT - 2] = v[1];
                xv[q][NT - 1] = *reinterpret_cast<const float*>(xb + (xs + 4u * (NT - 1)));
            }
#pragma unroll
            for (int jm = 0; jm < BT; ++jm) {
                const float v = *reinterpret_cast<const float*>(ab + (as + lane_a[jm]));
                av[q][jm] = base_ok[jm] ? v : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < G; ++q)
#pragma unroll
            for (int jm = 0; jm < BT; ++jm)
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    acc[jm][t] = basis_mfma(av[q][jm], xv[q][t], acc[jm][t]);
    }
}

// Round 6: a chunk of up to 64 edges of a row whose edges are sorted by SOURCE (the plan's skey list).  The edges of one (destination,
// source) pair are neighbours: the att rows of a pair's edges are summed first (fixed order), x[source] is fetched ONCE per pair, and
// the matrix instruction contracts four PAIRS per step instead of four edges - on the hub rows of the all-nodes baseline (the 645
// drugs: 4.8 edges per pair) the row gathers of x and the matrix instructions shrink 4.8-fold.  A pair that straddles two chunks is two
// partial pairs (the sums are linear).  Pair starts: a ballot of "my source differs from my left neighbour's"; the start lanes write
// their lane number into the wave's table in LDS, a lane group reads its pair's bounds from there.
template <int BT, int NT>
__device__ __forceinline__ void chunk_pairs(const BasisArgs& a, f32x4 (&acc)[BT][NT], uint32_t xo, uint32_t ao, uint32_t src, int cnt, int lane, int kg,
                                            uint32_t lane_x, const uint32_t (&lane_a)[BT], const bool (&base_ok)[BT], int* __restrict__ tab) {
    const char* __restrict__ xb = reinterpret_cast<const char*>(a.x);
    const char* __restrict__ ab = reinterpret_cast<const char*>(a.att);
    const uint32_t left = (uint32_t)__shfl_up((int)src, 1);
    const bool live = lane < cnt;
    const bool start = live && (lane == 0 || src != left);
    const unsigned long long starts = __ballot(start);
    const int np = __popcll(starts);
    const int rank = __popcll(starts & ((2ull << lane) - 1ull)) - 1;
    if (start) tab[rank] = lane;
    if (lane == 0) tab[np] = cnt;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                      // (the wave's own LDS writes, then its reads: in order)
    __builtin_amdgcn_wave_barrier();
    const int steps = (np + 3) >> 2;
    for (int j = 0; j < steps; ++j) {
        const int q = 4 * j + kg;
        const bool has = q < np;
        const int s0 = has ? tab[q] : 0, s1 = has ? tab[q + 1] : 0;
        const int len = s1 - s0;
        int longest = max(len, __shfl_xor(len, 16));
        longest = max(longest, __shfl_xor(longest, 32));                       // uniform: the longest of the step's four pairs
        const uint32_t xs = (uint32_t)__shfl((int)xo, s0) + lane_x;            // the pair's source row (requested first: it travels during the sums)
        float xv[NT];
#pragma unroll
        for (int t4 = 0; t4 < NT / 4; ++t4) {
            const f32x4 v = has ? *reinterpret_cast<const f32x4u*>(xb + (xs + 16u * t4)) : (f32x4)(0.f);
            xv[4 * t4] = v[0]; xv[4 * t4 + 1] = v[1]; xv[4 * t4 + 2] = v[2]; xv[4 * t4 + 3] = v[3];
        }
        if constexpr (NT % 4 == 2) {
            const f32x2 v = has ? *reinterpret_cast<const f32x2u*>(xb + (xs + 4u * (NT - 2))) : (f32x2)(0.f);
            xv[NT - 2] = v[0]; xv[NT - 1] = v[1];
        } else if constexpr (NT % 4 == 1) {
            xv[NT - 1] = has ? *reinterpret_cast<const float*>(xb + (xs + 4u * (NT - 1))) : 0.f;
        } else if constexpr (NT % 4 == 3) {
            const f32x2 v = has ? *reinterpret_cast<const f32x2u*>(xb + (xs + 4u * (NT - 3))) : (f32x2)(0.f);
            xv[NT - 3] = v[0]; xv[NT - 2] = v[1];
            xv[NT - 1] = has ? *reinterpret_cast<const float*>(xb + (xs + 4u * (NT - 1))) : 0.f;
        }
        float av[BT];
#pragma unroll
        for (int jm = 0; jm < BT; ++jm) av[jm] = 0.f;
        for (int t0 = 0; t0 < longest; t0 += 4) {                              // four edges of every pair requested together, added in edge order
            float v[4][BT];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = t0 + u < len;
                const uint32_t as = (uint32_t)__shfl((int)ao, ok ? s0 + t0 + u : 0);
#pragma unroll
                for (int jm = 0; jm < BT; ++jm) {
                    const float w = *reinterpret_cast<const float*>(ab + (as + lane_a[jm]));
                    v[u][jm] = (ok && base_ok[jm]) ? w : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int jm = 0; jm < BT; ++jm) av[jm] += v[u][jm];
        }
#pragma unroll
        for (int jm = 0; jm < BT; ++jm)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[jm][t] = basis_mfma(av[jm], xv[t], acc[jm][t]);
    }
    __builtin_amdgcn_wave_barrier();                                           // (the table is rewritten by the next chunk)
}

// acc += sum over the edges of chunks first, first + step, ... (64 edges each) of [e0, e1): K = four edges per instruction
template <int BT, int NT>
__device__ __forceinline__ void accumulate_chunks(const BasisArgs& a, f32x4 (&acc)[BT][NT], int e0, int e1, int first, int step, int lane,
                                                  int c, int kg, int* __restrict__ tab) {
    const uint32_t lane_x = (uint32_t)(NT * c) * 4u, sel0 = (uint32_t)kg * 4u;
    uint32_t lane_a[BT];
    bool base_ok[BT];
#pragma unroll
    for (int jm = 0; jm < BT; ++jm) {
        base_ok[jm] = 16 * jm + c < a.bases;
        lane_a[jm] = base_ok[jm] ? (uint32_t)(16 * jm + c) * 4u : 0u;          // (a lane without a base reads the row's first: in bounds)
    }
    const uint32_t ldx4 = (uint32_t)a.ld_x * 4u, b4 = (uint32_t)a.bases * 4u;
    int eb = e0 + 64 * first;
    // the edges by source: sums per (destination, source) pair (uniform branch).  Only the rows a whole workgroup walks - the hubs,
    // where pairs have several edges; on light rows (a pair is mostly one edge) the pair bookkeeping costs more than it saves:
    // 196 against 148 us per layer with every row on this path, 164 with the hub rows only: the pair's sums are a chain of
    // dependent round trips (bounds from the table, shuffle, att row, add) where the per-edge form requests eight steps' operands at once
    if (a.skey && a.fast_addr && a.mode == 1 && step > 1) {
        uint32_t k_next = (eb < e1 && eb + lane < e1) ? a.skey[eb + lane] : 0u;
        for (; eb < e1; eb += 64 * step) {
            const int cnt = min(64, e1 - eb);
            const uint32_t k = k_next;
            const int en = eb + 64 * step + lane;
            k_next = en < e1 ? a.skey[en] : 0u;
            const uint32_t src = k / a.R, rel = k - src * a.R;
            chunk_pairs<BT, NT>(a, acc, src * ldx4, rel * b4, src, cnt, lane, kg, lane_x, lane_a, base_ok, tab);
        }
        return;
    }
    // (mode 2: the per-edge contraction on the edges in SOURCE order - a row's edges from one source are neighbours, their x rows hit the L1)
    const bool by_source = a.skey && a.mode == 2;
    const uint32_t* __restrict__ keys = by_source ? a.skey : a.key;
    uint32_t k_next = (eb < e1 && eb + lane < e1) ? keys[eb + lane] : 0u;
    for (; eb < e1; eb += 64 * step) {
        const int cnt = min(64, e1 - eb);
        const uint32_t k = k_next;
        const int en = eb + 64 * step + lane;                                  // the next chunk's keys travel while this one is contracted
        k_next = en < e1 ? keys[en] : 0u;
        uint32_t rel = k / a.n, src = k - rel * a.n;                           // (one division per lane and 64 edges)
        if (by_source) { src = k / a.R; rel = k - src * a.R; }
        if (cnt == 64 && a.fast_addr) chunk_full<BT, NT>(a, acc, src * ldx4, rel * b4, sel0, lane_x, lane_a, base_ok);
        else chunk_any<BT, NT>(a, acc, src, rel, cnt, c, kg);
    }
}

// the tail of a row of the slab: x_i (the root term's operand) and the zero padding up to the dense product's K
__device__ __forceinline__ void write_tail(const BasisArgs& a, float* __restrict__ ur, int row, int lane) {
    const int tail = a.bases * a.fin;
    if (a.with_x)
        for (int col = lane; col < a.fin; col += 64) ur[tail + col] = a.x[(size_t)row * a.ld_x + col];
    for (int col = tail + (a.with_x ? a.fin : 0) + lane; col < a.kp; col += 64) ur[col] = 0.f;
}

template <int BT, int NT>
__global__ __launch_bounds__(kBasisThreads) void k_rgcn_basis(BasisArgs a) {
    __shared__ f32x4 red[kBasisWaves][64];                                     // one accumulator tile of every wave (heavy rows)
    __shared__ int pair_tab[kBasisWaves][68];                                  // every wave's pair starts of its current chunk
    if (a.side.dst) {                                                          // concat slot 0 (layers.py:264-266), by the whole grid
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = (int64_t)blockIdx.x * kBasisThreads + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBasisThreads) {
            const int64_t i = t / a.side.cols, cc = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + cc];
            a.side.dst[i * a.side.ld_dst + cc] = a.side.mode ? fabsf(v) : v;
        }
    }
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int wv = threadIdx.x >> 6;
    // ---- work items, drawn from a counter in the plan's order (rows by in-degree, largest first: the greedy longest-first
    // deal): item h < n_heavy is a heavy row (more than kHeavyEdges incoming edges) walked by the whole workgroup - its
    // waves take the row's 64-edge chunks in turn, their accumulators meet in LDS tile by tile and are added in wave order
    // (a hub of the all-nodes graph has thousands of edges: one wave would walk them for 200 us); the items behind are
    // groups of kBasisWaves light rows, a row per wave.  Every row is summed by one wave (or one workgroup) in a fixed
    // order, so the results do not depend on who drew what. ----
    __shared__ unsigned int item_s;
    const int n_rows = a.n_rows;
    const int n_heavy = a.slab_heavy ? *a.slab_heavy : a.n_heavy;
    const unsigned int n_items = (unsigned int)(n_heavy + (n_rows - n_heavy + kBasisWaves - 1) / kBasisWaves);
    for (;;) {
        __syncthreads();                                                       // (item_s and red of the previous item have been read)
        if (threadIdx.x == 0) item_s = atomicAdd(a.next_item, 1u);
        __syncthreads();
        const unsigned int item = item_s;
        if (item >= n_items) break;
        if ((int)item < n_heavy) {
            const int row = a.order[item];
            if (row < a.row_lo || row >= a.row_hi) continue;                   // (uniform over the workgroup)
            const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
            f32x4 acc[BT][NT];
#pragma unroll
            for (int jm = 0; jm < BT; ++jm)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[jm][t] = (f32x4)(0.f);
            accumulate_chunks<BT, NT>(a, acc, e0, e1, wv, kBasisWaves, lane, c, kg, pair_tab[wv]);
            const float inv = a.scale ? 1.0f / fmaxf(a.indeg[row], 1.0f) : 1.0f;
            float* __restrict__ ur = a.u + (size_t)(row - a.row_lo) * a.ld_u;
#pragma unroll
            for (int jm = 0; jm < BT; ++jm)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    __syncthreads();                                           // (the previous tile has been read)
                    red[wv][lane] = acc[jm][t];
                    __syncthreads();
                    if (wv == (jm * NT + t) % kBasisWaves) {                   // the tile's owner adds the waves' shares, in wave order
                        f32x4 s = red[0][lane];
#pragma unroll
                        for (int w = 1; w < kBasisWaves; ++w) s += red[w][lane];
                        if (NT * c + t < a.fin)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int b = 16 * jm + 4 * kg + i;
                                if (b < a.bases) ur[(size_t)b * a.fin + NT * c + t] = s[i] * inv;
                            }
                    }
                }
            if (wv == 0) write_tail(a, ur, row, lane);
            continue;
        }
        const int idx = n_heavy + (int)(item - (unsigned int)n_heavy) * kBasisWaves + wv;
        if (idx >= n_rows) continue;
        const int row = a.order[idx];
        if (row < a.row_lo || row >= a.row_hi) continue;
        const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
        f32x4 acc[BT][NT];
#pragma unroll
        for (int jm = 0; jm < BT; ++jm)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[jm][t] = (f32x4)(0.f);
        accumulate_chunks<BT, NT>(a, acc, e0, e1, 0, 1, lane, c, kg, pair_tab[wv]);
        // acc[jm][t][i] = U_i[base 16 jm + 4 kg + i][feature NT c + t]
        const float inv = a.scale ? 1.0f / fmaxf(a.indeg[row], 1.0f) : 1.0f;
        float* __restrict__ ur = a.u + (size_t)(row - a.row_lo) * a.ld_u;
#pragma unroll
        for (int jm = 0; jm < BT; ++jm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = 16 * jm + 4 * kg + i;
                if (b >= a.bases) continue;
                float* __restrict__ q = ur + (size_t)b * a.fin + NT * c;
                if (NT % 4 == 0 && a.vec) {
#pragma unroll
                    for (int t4 = 0; t4 < NT / 4; ++t4)
                        *reinterpret_cast<f32x4*>(q + 4 * t4) = (f32x4){acc[jm][4 * t4][i] * inv, acc[jm][4 * t4 + 1][i] * inv,
                                                                        acc[jm][4 * t4 + 2][i] * inv, acc[jm][4 * t4 + 3][i] * inv};
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (NT * c + t < a.fin) q[t] = acc[jm][t][i] * inv;
                }
            }
        write_tail(a, ur, row, lane);
    }
}

// Several slabs: every slab's own rows in the plan's order (in-degree, largest first; heavy rows in front), so that a slab's launch
// draws its own items only - with the plan's whole order every launch drew every item of all N rows (an atomic and two barriers
// each) and skipped the rows of the other slabs: O(slabs x N / 8) draws, ~8 M per layer at N = 10^6.  Slab s holds the rows
// [s slab_rows, (s + 1) slab_rows), each exactly once in `order`: its list starts at out[s slab_rows].  One workgroup per slab walks
// the order and keeps its rows, in order (a block-wide prefix sum per 1,024 entries: stable, the same list every run).
__global__ __launch_bounds__(1024) void k_slab_orders(const int32_t* __restrict__ order, int n, int n_heavy, int slab_rows,
                                                      int32_t* __restrict__ out, int32_t* __restrict__ heavy_out) {
    __shared__ int wave_total[16];
    __shared__ int running_s, heavy_s;
    const int slab = blockIdx.x, lo = slab * slab_rows, hi = min(n, lo + slab_rows);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { running_s = 0; heavy_s = 0; }
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        const int row = i < n ? order[i] : -1;
        const bool mine = row >= lo && row < hi;
        const unsigned long long m = __ballot(mine);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_total[wv] = __popcll(m);
        __syncthreads();
        int base = running_s, total = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) base += wave_total[w]; total += wave_total[w]; }
        if (mine) out[lo + base + before] = row;
        __syncthreads();
        if (threadIdx.x == 0) {
            running_s += total;
            if (i0 + 1024 <= n_heavy) heavy_s = running_s;                      // (every entry so far was a heavy row)
        }
        // the tile that straddles the end of the heavy rows: count its heavy members on their own
        if (i0 < n_heavy && i0 + 1024 > n_heavy) {
            const unsigned long long mh = __ballot(mine && i < n_heavy);
            if (lane == 0 && mh) atomicAdd(&heavy_s, __popcll(mh));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) heavy_out[slab] = heavy_s;
}

// [basis ; root ; 0]: the right-hand side of the slab's dense product
__global__ void k_basis_weights(const float* __restrict__ basis, const float* __restrict__ root, int64_t nb, int64_t nr, int64_t total,
                                float* __restrict__ w, unsigned int* __restrict__ counters, int n_counters) {
    if (blockIdx.x == 0 && (int)threadIdx.x < n_counters) counters[threadIdx.x] = 0u;       // the slabs' work counters
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        w[t] = t < nb ? basis[t] : (t < nb + nr ? root[t - nb] : 0.f);
}

struct BasisLayout {
    int kp;
    int64_t slab_rows, slabs;
    size_t w_off, q_off, o_off, u_off, total;     // o_off: (several slabs) [slabs] heavy counts, then [N] the slabs' own orders
};

BasisLayout basis_layout(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    BasisLayout l;
    l.kp = (int)(gn::ceil_div((bases + 1) * fin, 32) * 32);
    int64_t slab_bytes = kSlabBytes;
    if (const char* e = getenv("GN_RGCN_SLAB_MB")) {                       // test hook: small slabs (the same bits: a slab is a range of rows)
        const long mb = atol(e);
        if (mb >= 1 && mb <= 4096) slab_bytes = (int64_t)mb << 20;
    }
    l.slab_rows = std::min<int64_t>(std::max<int64_t>(plan->num_nodes, 1), std::max<int64_t>(256, slab_bytes / ((int64_t)l.kp * 4)));
    l.slab_rows = std::max<int64_t>(l.slab_rows, gn::ceil_div(std::max<int64_t>(plan->num_nodes, 1), 256));   // at most 256 slabs (one work counter each)
    l.slabs = gn::ceil_div(std::max<int64_t>(plan->num_nodes, 1), l.slab_rows);
    l.w_off = 0;
    l.q_off = ((size_t)l.kp * fout * sizeof(float) + 255) & ~size_t(255);
    l.o_off = l.q_off + (((size_t)l.slabs * sizeof(unsigned int) + 255) & ~size_t(255));
    l.u_off = l.o_off + (l.slabs > 1 ? ((((size_t)l.slabs + (size_t)plan->num_nodes) * sizeof(int32_t) + 255) & ~size_t(255)) : 0);
    l.total = l.u_off + (size_t)l.slab_rows * l.kp * sizeof(float);
    return l;
}

template <int BT>
gn_status launch_bt(int nt, const BasisArgs& a, int grid, hipStream_t st) {
#define GN_BASIS_CASE(N) case N: if constexpr (BT * N <= 16) { k_rgcn_basis<BT, N><<<grid, kBasisThreads, 0, st>>>(a); break; } else return gn::fail(GN_ERR_UNSUPPORTED, "unsupported shape")
    switch (nt) {
        GN_BASIS_CASE(1); GN_BASIS_CASE(2); GN_BASIS_CASE(3); GN_BASIS_CASE(4);
        GN_BASIS_CASE(5); GN_BASIS_CASE(6); GN_BASIS_CASE(7); GN_BASIS_CASE(8);
        default: return gn::fail(GN_ERR_UNSUPPORTED, "unsupported shape");
    }
#undef GN_BASIS_CASE
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

// ---- the layer's weight gradient for large supervertices -----------------------------------------------------------------
//   dw[r] = sum_{e in r} x[src_e]^T gm[dst_e]        ([R][in][out]; then dbasis = att^T dw, datt = dw basis^T: layers.py:172-173)
// Where the fused kernel of rel_grad.hip does not apply (its gradient table lives in LDS: a few thousand nodes) the reference's
// own formulation is the only O(E) one: per relation, the outer products of the edges' endpoint rows.  A wave owns an item
// (<= kRelDwItemEdges edges of one relation, contiguous in the caller's type-sorted list); per FOUR edges it issues
// v_mfma_f32_16x16x4_f32 with A = x[src_e, features] (M = in), B = gm[dst_e, outputs] (N = out), K = the four edges: dw[r]
// accumulates in registers.  A relation that is one item is stored as it is; the parts of a larger one go to slots of the
// workspace and k_rel_dw_combine adds them in part order - no atomics, the same bits every run.
struct RelDwArgs {
    const int64_t* src; const int64_t* dst;
    const float* x; int64_t ld_x; int fin;
    const float* gm; int64_t ld_g; int fout;
    const int32_t* items; int n_items;
    float* dw;                                     // [R][fin * fout]
    float* parts;                                  // [n_parts][fin * fout]
    int vec;
};

template <int MT, int NT>
__global__ __launch_bounds__(256) void k_rel_dw(RelDwArgs a) {
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), W = gridDim.x * 4;
    const int64_t ff = (int64_t)a.fin * a.fout;
    for (int it = wave; it < a.n_items; it += W) {
        const int32_t* __restrict__ d = a.items + 4 * (size_t)it;
        const int rel = d[0], e0 = d[1], e1 = d[2], slot = d[3];
        f32x4 acc[MT][NT];
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int tn = 0; tn < NT; ++tn) acc[tm][tn] = (f32x4)(0.f);
        int64_t s_next = e0 + lane < e1 ? a.src[e0 + lane] : 0, d_next = e0 + lane < e1 ? a.dst[e0 + lane] : 0;
        for (int eb = e0; eb < e1; eb += 64) {
            const int cnt = min(64, e1 - eb);
            const int64_t s_mine = s_next, d_mine = d_next;
            const int en = eb + 64 + lane;                                     // the next chunk's ids travel while this one is contracted
            s_next = en < e1 ? a.src[en] : 0; d_next = en < e1 ? a.dst[en] : 0;
            const int steps = (cnt + 3) >> 2;
            for (int j0 = 0; j0 < steps; j0 += 4) {
                float xv[4][MT], gv[4][NT];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sel = 4 * (j0 + q) + kg;                          // edge of lane group kg in step j0 + q
                    const int64_t s = __shfl(s_mine, sel & 63), t = __shfl(d_mine, sel & 63);
                    const bool live = sel < cnt;
                    const float* __restrict__ xr = a.x + s * a.ld_x + MT * c;   // features MT c .. of the edge's source
                    const float* __restrict__ gr = a.gm + t * a.ld_g + NT * c;  // outputs NT c .. of its destination
                    if (a.vec && MT % 4 == 0) {
#pragma unroll
                        for (int t4 = 0; t4 < MT / 4; ++t4) {
                            const f32x4 v = live ? *reinterpret_cast<const f32x4u*>(xr + 4 * t4) : (f32x4)(0.f);
                            xv[q][4 * t4] = v[0]; xv[q][4 * t4 + 1] = v[1]; xv[q][4 * t4 + 2] = v[2]; xv[q][4 * t4 + 3] = v[3];
                        }
                    } else {
#pragma unroll
                        for (int tm = 0; tm < MT; ++tm) xv[q][tm] = (live && MT * c + tm < a.fin) ? xr[tm] : 0.f;
                    }
#pragma unroll
                    for (int tn = 0; tn < NT; ++tn) gv[q][tn] = (live && NT * c + tn < a.fout) ? gr[tn] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int tm = 0; tm < MT; ++tm)
#pragma unroll
                        for (int tn = 0; tn < NT; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_16x16x4f32(xv[q][tm], gv[q][tn], acc[tm][tn], 0, 0, 0);
            }
        }
        // acc[tm][tn][i] = dw[feature MT (4 kg + i) + tm][output NT c + tn]
        float* __restrict__ out = slot >= 0 ? a.parts + (size_t)slot * ff : a.dw + (size_t)rel * ff;
#pragma unroll
        for (int tm = 0; tm < MT; ++tm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int f = MT * (4 * kg + i) + tm;
                if (f >= a.fin) continue;
#pragma unroll
                for (int tn = 0; tn < NT; ++tn)
                    if (NT * c + tn < a.fout) out[(size_t)f * a.fout + NT * c + tn] = acc[tm][tn][i];
            }
    }
}

// dw[r] of a relation cut into parts = its parts' sums, in part order; relations without edges in the shard are zero
__global__ __launch_bounds__(256) void k_rel_dw_combine(const int32_t* __restrict__ multi, int n_multi, const float* __restrict__ parts,
                                                        float* __restrict__ dw, int64_t ff) {
    for (int m = blockIdx.x; m < n_multi; m += gridDim.x) {
        const int rel = multi[4 * m], slot0 = multi[4 * m + 1], np = multi[4 * m + 2];
        for (int64_t o = threadIdx.x; o < ff; o += 256) {
            float s = 0.f;
            int p = 0;
            for (; p + 4 <= np; p += 4) {                                       // four parts requested together, added in order
                const float v0 = parts[(size_t)(slot0 + p) * ff + o], v1 = parts[(size_t)(slot0 + p + 1) * ff + o];
                const float v2 = parts[(size_t)(slot0 + p + 2) * ff + o], v3 = parts[(size_t)(slot0 + p + 3) * ff + o];
                s += v0; s += v1; s += v2; s += v3;
            }
            for (; p < np; ++p) s += parts[(size_t)(slot0 + p) * ff + o];
            dw[(size_t)rel * ff + o] = s;
        }
    }
}

template <int MT>
gn_status launch_dw(int nt, const RelDwArgs& a, int grid, hipStream_t st) {
#define GN_DW_CASE(N) case N: if constexpr (MT * N <= 16) { k_rel_dw<MT, N><<<grid, 256, 0, st>>>(a); break; } else return gn::fail(GN_ERR_UNSUPPORTED, "unsupported shape")
    switch (nt) {
        GN_DW_CASE(1); GN_DW_CASE(2); GN_DW_CASE(3); GN_DW_CASE(4);
        default: return gn::fail(GN_ERR_UNSUPPORTED, "unsupported shape");
    }
#undef GN_DW_CASE
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// ---- called from rgcn.hip ----
bool gn_rgcn_basis_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    if (!plan || !plan->row_order.p || fin < 1 || fin > 128 || bases < 1 || bases > 64 || fout < 1) return false;
    const int64_t bt = gn::ceil_div(bases, 16), nt = gn::ceil_div(fin, 16);
    return bt * nt <= 16;
}

size_t gn_rgcn_basis_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    return basis_layout(plan, fin, fout, bases).total;
}

extern "C" {

size_t gn_rgcn_weight_grad_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout) {
    if (!plan || fin < 1 || fout < 1) return 0;
    return (size_t)std::max<int64_t>(plan->n_dw_parts, 1) * fin * fout * sizeof(float);
}

int gn_rgcn_weight_grad_supported(const gn_rgcn_plan* plan, int64_t fin, int64_t fout) {
    if (!plan || fin < 1 || fout < 1 || fin > 128 || fout > 64) return 0;
    if (plan->shard_edges > 0 && plan->n_dw_items == 0) return 0;                 // (a relation too long for the item list)
    return gn::ceil_div(fin, 16) * gn::ceil_div(fout, 16) <= 16;
}

gn_status gn_rgcn_weight_grad_f32(const gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst, const float* x, int64_t ld_x,
                                  int64_t fin, const float* gm, int64_t ld_g, int64_t fout, float* dw, void* workspace,
                                  size_t workspace_bytes, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    if (!gn_rgcn_weight_grad_supported(plan, fin, fout))
        return gn::fail(GN_ERR_UNSUPPORTED, "the general relational weight gradient covers up to 128 input and 64 output features");
    const int64_t R = plan->num_relations, ff = fin * fout;
    if (R == 0) return GN_OK;
    GN_REQUIRE(dw && (plan->shard_edges == 0 || (src && dst && x && gm)) && ld_x >= fin && ld_g >= fout, "operand pointer is null or a leading dimension too small");
    GN_REQUIRE(plan->n_dw_parts == 0 || (workspace && workspace_bytes >= gn_rgcn_weight_grad_workspace_bytes(plan, fin, fout)),
               "workspace too small: need %zu bytes", gn_rgcn_weight_grad_workspace_bytes(plan, fin, fout));
    hipStream_t st = gn::as_stream(stream);
    GN_HIP(hipMemsetAsync(dw, 0, (size_t)R * ff * sizeof(float), st));         // relations without edges in the shard
    if (plan->n_dw_items == 0) return GN_OK;
    RelDwArgs a;
    a.src = src; a.dst = dst; a.x = x; a.ld_x = ld_x; a.fin = (int)fin; a.gm = gm; a.ld_g = ld_g; a.fout = (int)fout;
    a.items = plan->dw_items.p; a.n_items = (int)plan->n_dw_items; a.dw = dw; a.parts = static_cast<float*>(workspace);
    a.vec = (fin % 16 == 0) && (ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int mt = (int)gn::ceil_div(fin, 16), nt = (int)gn::ceil_div(fout, 16);
    const int grid = (int)std::min<int64_t>((int64_t)gn::compute_units() * 4, gn::ceil_div(plan->n_dw_items, 4));
    gn_status s;
    switch (mt) {
        case 1: s = launch_dw<1>(nt, a, grid, st); break;
        case 2: s = launch_dw<2>(nt, a, grid, st); break;
        case 3: s = launch_dw<3>(nt, a, grid, st); break;
        case 4: s = launch_dw<4>(nt, a, grid, st); break;
        case 5: s = launch_dw<5>(nt, a, grid, st); break;
        case 6: s = launch_dw<6>(nt, a, grid, st); break;
        case 7: s = launch_dw<7>(nt, a, grid, st); break;
        default: s = launch_dw<8>(nt, a, grid, st); break;
    }
    if (s != GN_OK) return s;
    if (plan->n_dw_multi > 0) {
        k_rel_dw_combine<<<(int)std::min<int64_t>(plan->n_dw_multi, 1024), 256, 0, st>>>(plan->dw_multi.p, (int)plan->n_dw_multi,
                                                                                         static_cast<const float*>(workspace), dw, ff);
        GN_LAUNCH_CHECK();
    }
    return GN_OK;
}

}  // extern "C"

gn_status gn_rgcn_basis_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* basis,
                                const float* att, int64_t bases, const float* root, const float* bias, int64_t fout, int relu,
                                int partial, int fast_arith, float* out, int64_t ld_out, const gn_side_copy& side, void* ws,
                                size_t ws_bytes, hipStream_t st) {
    const BasisLayout l = basis_layout(plan, fin, fout, bases);
    GN_REQUIRE(ws && ws_bytes >= l.total, "workspace too small: the basis-space path needs %zu bytes, got %zu", l.total, ws_bytes);
    const int64_t N = plan->num_nodes;
    float* W = reinterpret_cast<float*>(static_cast<char*>(ws) + l.w_off);
    float* U = reinterpret_cast<float*>(static_cast<char*>(ws) + l.u_off);
    const int64_t nb = bases * fin * fout, nr = partial ? 0 : fin * fout, total = (int64_t)l.kp * fout;
    unsigned int* counters = reinterpret_cast<unsigned int*>(static_cast<char*>(ws) + l.q_off);
    GN_REQUIRE(l.slabs <= 256, "too many slabs of rows (%lld)", (long long)l.slabs);
    k_basis_weights<<<gn::stream_grid(total, 256), 256, 0, st>>>(basis, root, nb, nr, total, W, counters, (int)l.slabs);
    GN_LAUNCH_CHECK();
    BasisArgs a;
    a.x = x; a.ld_x = ld_x; a.fin = (int)fin; a.att = att; a.bases = (int)bases;
    a.rowptr = plan->rowptr.p; a.key = plan->key.p; a.indeg = plan->indeg.p; a.order = plan->row_order.p;
    // Measurement hook (round 6, profiles/r06_experiments.md 5): GN_RGCN_BASIS_ORDER=pairs - the att rows of a (destination, source) pair's
    // edges summed first, x once per pair, on the hub rows; =source - the per-edge contraction on source-ordered edges.  Both need the
    // plan's source-ordered list (built when the variable is set at plan creation) and both measured SLOWER or equal: the default stays
    // the per-edge contraction on the relation-ordered list.
    const char* order = getenv("GN_RGCN_BASIS_ORDER");
    a.skey = plan->skey.p;
    a.mode = (a.skey && order && order[0] == 'p') ? 1 : ((a.skey && order && order[0] == 's') ? 2 : 0);
    a.R = (uint32_t)plan->num_relations;
    a.n = (uint32_t)N; a.u = U; a.ld_u = l.kp; a.kp = l.kp; a.n_heavy = (int)plan->heavy_rows;
    a.n_rows = (int)N; a.slab_heavy = nullptr;
    int32_t* slab_heavy = reinterpret_cast<int32_t*>(static_cast<char*>(ws) + l.o_off);
    int32_t* slab_order = slab_heavy + l.slabs;
    if (l.slabs > 1) {
        k_slab_orders<<<(int)l.slabs, 1024, 0, st>>>(plan->row_order.p, (int)N, (int)plan->heavy_rows, (int)l.slab_rows, slab_order, slab_heavy);
        GN_LAUNCH_CHECK();
    }
    a.scale = partial ? 0 : 1; a.with_x = partial ? 0 : 1;
    a.vec = (fin % 16 == 0) && (ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    // (measured and dropped: the att table staged in LDS - a fifth of the gathered bytes - made the layer 8 % SLOWER: the
    // fill costs more than the L2-resident 64-byte rows; what bounds the kernel is the rate of row gathers from L2 / the
    // Infinity Cache, 6-7 TB/s of 128- and 256-byte pieces of x)
    a.fast_addr = a.vec && (uint64_t)N * (uint64_t)ld_x * 4 < (1ull << 32) && (uint64_t)plan->num_relations * (uint64_t)bases * 4 < (1ull << 32);
    const int bt = (int)gn::ceil_div(bases, 16), nt = (int)gn::ceil_div(fin, 16);
    // four waves per SIMD on every compute unit where the registers allow (<= 128): two workgroups of eight waves
    const int grid = (int)std::min<int64_t>((int64_t)gn::compute_units() * 2, std::max<int64_t>(plan->heavy_rows, gn::ceil_div(N, kBasisWaves)));
    for (int64_t r0 = 0; r0 < N; r0 += l.slab_rows) {
        const int64_t r1 = std::min(N, r0 + l.slab_rows);
        a.row_lo = (int)r0; a.row_hi = (int)r1;
        a.next_item = counters + r0 / l.slab_rows;
        if (l.slabs > 1) { a.order = slab_order + r0; a.n_rows = (int)(r1 - r0); a.slab_heavy = slab_heavy + r0 / l.slab_rows; }
        a.side = r0 == 0 ? side : gn_side_copy{nullptr, 0, nullptr, 0, 0, 0, 0};
        gn_status s;
        switch (bt) {
            case 1: s = launch_bt<1>(nt, a, grid, st); break;
            case 2: s = launch_bt<2>(nt, a, grid, st); break;
            case 3: s = launch_bt<3>(nt, a, grid, st); break;
            default: s = launch_bt<4>(nt, a, grid, st); break;
        }
        if (s != GN_OK) return s;
        s = gn_gemm_f32(U, l.kp, 0, nullptr, 0, W, fout, 0, out + r0 * ld_out, ld_out, 0, r1 - r0, fout, l.kp, 1,
                        partial ? nullptr : bias, ((relu && !partial) ? GN_GEMM_RELU : 0) | (fast_arith ? GN_GEMM_ARITH_FAST : 0) | GN_GEMM_SPLIT_KERNEL, st);
        if (s != GN_OK) return s;
    }
    return GN_OK;
}
