// LDS-resident multi-relational layer for small supervertices (the drug supervertex of PoSE:
// n_d = 645 nodes, 48 -> 32 features, ~10^3 relations, millions of edges).
//
//   out[i] = (sum_{e: dst=i} x[src_e] W_{r(e)}) / max(1, indeg_i) + x[i] root (+ bias)
//
// With a few hundred nodes the whole [N, out] accumulator (82.5 KB) fits in a CU's LDS, and so
// does a source tile of H_r = X W_r.  The type-sorted edge list is cut at plan time into work
// items (relation r, source tile t, <= kChunk edges); one persistent workgroup per CU walks its
// items, all items of one source tile first:
//   (1) H tile = X[tile] @ W_r on the matrix cores (v_mfma_f32_16x16x4_f32, exact fp32).  The X
//       fragments of the tile stay in registers across items; the W_r fragments (16-byte loads
//       from the transposed per-relation weights) were prefetched during the previous item.
//   (2) every edge of the item adds one 128-byte LDS row H[src] into the LDS accumulator row of
//       its destination.  The item's packed (dst, src) words are staged in LDS with coalesced
//       loads issued before (1).  Inside an item every destination belongs to exactly one of
//       the 128 eight-lane slots (assigned at plan time, heaviest destinations dealt first, so
//       the slots carry near-equal edge counts): no atomics, fixed summation order, bitwise
//       reproducible results.
// The per-workgroup accumulators leave as slabs; a second kernel sums them in a fixed order and
// applies mean / root / bias / activation (or emits the raw partial sum for the multi-GPU
// all-reduce).
//
// HBM traffic per edge is the 4-byte packed (dst, src) word; W (R x 6 KB) and X (124 KB) are
// L2-resident.  The matrix-core time of (1), 2 N_tile x in x out flops per item at the fp32 MFMA
// rate (256 flop/clk/CU), is the larger cost below ~10^4 edges per relation.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kFout = 32;                  // out_features of the layer this path is built for
constexpr int kLpr = kFout / 4;            // float4 per full feature row (slab layout, finalisation)
constexpr int kNB = 128;                   // destination slots per item = edges in flight per workgroup
constexpr int kGroups = 256;               // persistent workgroups = CUs of an MI355X
constexpr int kChunk = 4096;               // packed words per work item = LDS edge buffer (16 KB)
constexpr int kItemEdges = kChunk - 3 * 128; // edges per work item: every slot list is padded to 4 words
constexpr int kItemOverhead = 2048;        // H-tile cost in edge equivalents (LPT balancing)
constexpr size_t kLdsBudget = 159 * 1024;
constexpr size_t kLdsHalfBudget = 80 * 1024 - 512;   // two workgroups per CU (allocation granularity left over)
constexpr int kMaxRowTilesPerWave = 3;
constexpr uint32_t kEndFlag = 0x80000000u;  // packed word: [31] last edge of its destination run, [30:16] dst, [15:0] src - tile base

struct FastGeom {
    int tiles = 0;      // source tiles
    int ts = 0;         // rows per source tile, a multiple of 16 (MFMA row tile)
    int cols = 0;       // output columns per workgroup: 32 (one workgroup per CU) or 16 (two, column halves)
    size_t lds_bytes = 0;
};

constexpr int kDescWin = 32;               // work descriptors cached in LDS (1 KB window)

size_t lds_bytes_for(int64_t n, int64_t ts, int cols) {
    return (size_t)(n + ts) * cols * sizeof(float) + (size_t)(kChunk + 8) * sizeof(uint32_t) + kDescWin * 32;
}

// Two geometries.  Column halves: a workgroup of 512 threads owns 16 of the 32 output columns (its half of
// the H tile and of the accumulator), two workgroups share a CU, and while one is in its MFMA phase the
// other gathers from LDS - the two phases use different units and a single workgroup can only alternate.
// Needs 2 x LDS <= 160 KB.  Full width (1024 threads, one workgroup per CU) covers the larger node counts.
FastGeom geometry(int64_t n) {
    FastGeom g;
    for (int cols = kFout / 2; cols <= kFout; cols *= 2) {
        const int waves = kNB / (64 / (cols / 4));
        const size_t budget = cols == kFout ? kLdsBudget : kLdsHalfBudget;
        for (int t = 1; t <= 16; ++t) {
            const int64_t ts = gn::ceil_div(gn::ceil_div(n, t), 16) * 16;
            if (ts > 16 * waves * kMaxRowTilesPerWave) continue;
            const size_t bytes = lds_bytes_for(n, ts, cols);
            if (bytes <= budget) {
                g.tiles = (int)gn::ceil_div(n, ts); g.ts = (int)ts; g.cols = cols; g.lds_bytes = bytes;
                return g;
            }
        }
    }
    return g;
}

// One work item as the kernel reads it (32 bytes, wave-uniform scalar load).
struct alignas(32) WorkDesc {
    int32_t rel, tile, start, count, item, pad0, pad1, pad2;
};

struct FastArgs {
    const float* x; int64_t ld_x; int n;
    const float* wt;                // [R, FOUT, FIN]: W_r transposed (k contiguous)
    const uint32_t* packed;         // (dst << 16) | (src - tile * ts), slot-sorted inside each item
    const int32_t* slot_off;        // [n_items * kNB + 1]
    const WorkDesc* work;           // per workgroup, (tile, item) order
    const int32_t* wg_begin;        // [groups + 1] ranges into work
    int ts;
    float* slabs;                   // [n, groups, FOUT]
};

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt(0), which
// would expose the latency of every global load prefetched across the barrier.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Pointer parameters are passed one by one and __restrict__-qualified: the work descriptors are
// then read with scalar loads, and every prefetch below is unconditional (clamped indices) so that
// the compiler's vmcnt bookkeeping stays exact and no wait drains more than it needs.
struct FastDims { int64_t ld_x; int n; int ts; int groups; };

#ifdef GN_STAMPS
// Diagnostic build only (make STAMPS=1): per-workgroup phase times, never part of the product library.
__device__ unsigned long long g_stamps[2 * kGroups][8];
__device__ unsigned long long g_wave_stamps[2 * kGroups][16][4];   // per wave: gather, top, trips, mfma
#define GN_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime()
#else
#define GN_STAMP(var)
#endif

template <int FIN, int RT, int COLS>
__global__ __launch_bounds__(32 * COLS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_rgcn_lds(const float* __restrict__ x, const float* __restrict__ wt,
                                                        const uint32_t* __restrict__ packed,
                                                        const int32_t* __restrict__ slot_off,
                                                        const WorkDesc* __restrict__ work,
                                                        const int32_t* __restrict__ wg_begin,
                                                        float* __restrict__ slabs, FastDims a) {
    constexpr int KC = FIN / 16;             // 16-deep K chunks
    constexpr int CT = COLS / 16;            // 16-column tiles of this workgroup's column block
    constexpr int LPR = COLS / 4;            // lanes per feature row (float4 each)
    constexpr int SLOTS = 64 / LPR;          // edges in flight per wave
    constexpr int THREADS = 32 * COLS;       // kNB slots x LPR lanes
    constexpr int WAVES = THREADS / 64;
    constexpr int EREGS = kChunk / THREADS;
    extern __shared__ f32x4 lds4[];
    f32x4* acc4 = lds4;                                          // [n][LPR]
    f32x4* h4 = lds4 + (size_t)a.n * LPR;                        // [ts][LPR]
    float* hf = reinterpret_cast<float*>(h4);
    uint32_t* ebuf = reinterpret_cast<uint32_t*>(h4 + (size_t)a.ts * LPR);    // [kChunk + 8]
    int4* wdesc = reinterpret_cast<int4*>(ebuf + kChunk + 8);                  // [kDescWin][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int slot = lane / LPR, j = lane % LPR;
    // workgroups grp and grp + groups share the work list (and, 256 apart, an XCD's L2 for it)
    const int grp = blockIdx.x % a.groups, col0 = (blockIdx.x / a.groups) * COLS;
    // row tiles are dealt to waves in opposite orders by the two column halves: the waves that hold
    // one row tile more than the rest then sit on different SIMDs of the CU
    const int wrow = col0 ? WAVES - 1 - wave : wave;

    for (int i = tid; i < a.n * LPR; i += THREADS) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid < 8) ebuf[kChunk + tid] = 0u;

    int wi = wg_begin[grp];
    const int wend = wg_begin[grp + 1];
    f32x4 afrag[RT][KC];                 // X fragments of the current source tile
    f32x4 bfrag[KC][CT];                 // W_rel fragments of the current item
    int a_tile = -1;                     // source tile whose X fragments sit in afrag
    int h_rel = -1, h_tile = -1;         // (relation, tile) whose H tile sits in LDS
    // This workgroup's work descriptors sit in LDS (a scalar or global load per item would put an
    // L2 / HBM round trip on every item's critical path); a window of kDescWin, re-based when it
    // runs out.
    int wbase = wi;
    auto fill_window = [&](int base) {
        const int cnt = min(kDescWin, wend - base) * 2;
        const int4* __restrict__ src = reinterpret_cast<const int4*>(work + base);
        for (int i = tid; i < cnt; i += THREADS) wdesc[i] = src[i];
    };
    struct Desc { int rel, tile, start, count, item; };
    auto read_desc = [&](int w) {
        const int k = min(w, wend - 1) - wbase;
        const int4 lo = wdesc[2 * k];
        const int it = wdesc[2 * k + 1].x;
        Desc r;
        r.rel = __builtin_amdgcn_readfirstlane(lo.x); r.tile = __builtin_amdgcn_readfirstlane(lo.y);
        r.start = __builtin_amdgcn_readfirstlane(lo.z); r.count = __builtin_amdgcn_readfirstlane(lo.w);
        r.item = __builtin_amdgcn_readfirstlane(it);
        return r;
    };
    fill_window(wbase);
    __syncthreads();                     // accumulator is zeroed, descriptors are in place
    Desc d = {0, 0, 0, 0, 0}, dn = d;
    auto load_b = [&](int rel, f32x4 (&b)[KC][CT]) {
        // lane (c16, q) holds W[k = 16 kc + 4 q + jj][16 ct + c16], jj = 0..3  (k permuted like A)
        const float* __restrict__ wr = wt + (size_t)rel * (FIN * kFout) + 4 * q;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
                b[kc][ct] = *reinterpret_cast<const f32x4*>(wr + (col0 + 16 * ct + c16) * FIN + 16 * kc);
    };
    if (wi < wend) {
        d = read_desc(wi);
        load_b(d.rel, bfrag);
    }

#ifdef GN_STAMPS
    unsigned long long st_top = 0, st_mfma = 0, st_barb = 0, st_gather = 0, st_bara = 0, st_items = 0, st_trips = 0;
    const unsigned long long st_begin = __builtin_amdgcn_s_memrealtime();
#endif
    for (; wi < wend; ++wi) {
        GN_STAMP(t0);
        if (wi + 1 - wbase >= kDescWin && wi + 1 < wend) {   // workgroup-uniform, rare: re-base the window
            lds_barrier();
            wbase = wi;
            fill_window(wbase);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
        dn = read_desc(wi + 1);
        // ---- item prologue: these loads stay in flight across the barrier and the H-tile build ----
        if (d.tile != a_tile) {                          // wave-uniform; a few times per kernel
            a_tile = d.tile;                             // issued first: the MFMAs wait for these only
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int lrow = (wrow + r * WAVES) * 16 + c16;
                const int grow = d.tile * a.ts + lrow;
                const bool valid = lrow < a.ts && grow < a.n;
                const float* __restrict__ xr = x + (int64_t)(valid ? grow : 0) * a.ld_x + 4 * q;
                const float keep = valid ? 1.0f : 0.0f;
#pragma unroll
                for (int kc = 0; kc < KC; ++kc) afrag[r][kc] = *reinterpret_cast<const f32x4*>(xr + 16 * kc) * keep;
            }
        }
        uint32_t e[EREGS];
#pragma unroll
        for (int i = 0; i < EREGS; ++i) {
            const int idx = tid + i * THREADS;
            e[i] = packed[d.start + min(idx, d.count - 1)];      // entries past the item are never read back
        }
        const int sidx = d.item * kNB + wave * SLOTS + slot;
        int g4 = (slot_off[sidx] - d.start) >> 2;                // this slot's list, in groups of 4 words
        const int g4end = (slot_off[sidx + 1] - d.start) >> 2;
        f32x4 bnext[KC][CT];
        load_b(dn.rel, bnext);                           // dn == d on the last item
        const bool build_h = d.rel != h_rel || d.tile != h_tile;   // chunks of one segment share the H tile
        GN_STAMP(t1);
        lds_barrier();            // previous item's gather is done with the H tile and the edge buffer
        GN_STAMP(t2);
        // ---- (1) H tile = X[tile rows] @ W_rel ------------------------------------------------
        if (build_h) {
            h_rel = d.rel; h_tile = d.tile;
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int rt = wrow + r * WAVES;
                if (rt * 16 < a.ts) {                    // wave-uniform
                    f32x4 acc[CT];
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                            for (int ct = 0; ct < CT; ++ct)
                                acc[ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(afrag[r][kc][jj], bfrag[kc][ct][jj],
                                                                               acc[ct], 0, 0, 0);
                    // D fragment: lane (c16, q) holds H[rt*16 + 4q + i][16 ct + c16], i = 0..3
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                        for (int i = 0; i < 4; ++i) hf[(rt * 16 + 4 * q + i) * COLS + 16 * ct + c16] = acc[ct][i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < EREGS; ++i) ebuf[tid + i * THREADS] = e[i];
        GN_STAMP(t3);
        lds_barrier();
        GN_STAMP(t4);
        // ---- (2) gather-accumulate: this slot owns the destinations of slot (wave, slot) ----------
        // The wave's issue slots are the scarce resource of this phase (four waves share a SIMD), so
        // the loop is kept to ~1.5 instructions per edge: a slot's list is padded to whole groups of
        // four words (one ds_read_b128 per group, no bounds arithmetic), the plan marks the last
        // edge of every destination run (bit 31), the run's sum lives in registers and is added to
        // the accumulator row only there.  The slot is the only owner of its destinations inside
        // an item, so the read-modify-write needs no atomics and the order of adds is fixed.
        const uint4* ebuf4 = reinterpret_cast<const uint4*>(ebuf);
        const f32x4* hrow = h4 + j;                       // this lane's float4 of an H row / accumulator row
        f32x4* arow = acc4 + j;
        auto src_of = [](uint32_t w) { return __builtin_amdgcn_ubfe(w, 0, 16) * LPR; };
        auto dst_of = [](uint32_t w) { return __builtin_amdgcn_ubfe(w, 16, 15) * LPR; };
        // sum_k = sum_{k-1} * keep_{k-1} + H[src_k], keep = 0 after the last edge of a run: the run sum restarts
        // inside the FMA instead of with a reset under a branch (x * 1 + y and x * 0 + y are exact).
        f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
        float keep = 0.f;
        uint4 W = ebuf4[min(g4, kChunk / 4)];
        while (__any(g4 < g4end)) {
            const uint4 Wn = ebuf4[min(g4 + 1, kChunk / 4)];          // next group, in flight during the adds
            if (g4 < g4end) {
                const f32x4 r0 = hrow[src_of(W.x)], r1 = hrow[src_of(W.y)], r2 = hrow[src_of(W.z)], r3 = hrow[src_of(W.w)];
                // accumulator rows of the runs that end in this group: read now, with the H rows, so that the
                // adds below wait for LDS once per group instead of once per run (left undefined otherwise:
                // they are only consumed under the same flag)
                const bool e0 = (int32_t)W.x < 0, e1 = (int32_t)W.y < 0, e2 = (int32_t)W.z < 0, e3 = (int32_t)W.w < 0;
                const int i0 = dst_of(W.x), i1 = dst_of(W.y), i2 = dst_of(W.z), i3 = dst_of(W.w);
                f32x4 a0, a1, a2, a3;
                asm volatile("" : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3));   // defined-as-anything, no zeroing moves
                if (e0) a0 = arow[i0];
                if (e1) a1 = arow[i1];
                if (e2) a2 = arow[i2];
                if (e3) a3 = arow[i3];
                const f32x4 s0 = sum * keep + r0;
                if (e0) arow[i0] = a0 + s0;
                const f32x4 s1 = s0 * (e0 ? 0.f : 1.f) + r1;
                if (e1) arow[i1] = a1 + s1;
                const f32x4 s2 = s1 * (e1 ? 0.f : 1.f) + r2;
                if (e2) arow[i2] = a2 + s2;
                const f32x4 s3 = s2 * (e2 ? 0.f : 1.f) + r3;
                if (e3) arow[i3] = a3 + s3;
                sum = s3;
                keep = e3 ? 0.f : 1.f;
            }
            W = Wn;
            ++g4;
#ifdef GN_STAMPS
            st_trips += 1;
#endif
        }
#ifdef GN_STAMPS
        {
            unsigned long long t5 = __builtin_amdgcn_s_memtime();
            st_top += t1 - t0; st_bara += t2 - t1; st_mfma += t3 - t2; st_barb += t4 - t3; st_gather += t5 - t4;
            st_items += 1;
        }
#endif
        d = dn;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) bfrag[kc][ct] = bnext[kc][ct];
    }
#ifdef GN_STAMPS
    if (lane == 0) {
        unsigned long long* w = g_wave_stamps[blockIdx.x][wave];
        w[0] = st_gather; w[1] = st_top; w[2] = st_trips; w[3] = st_mfma;
    }
    if (lane == 0 && (wave == 0)) {
        unsigned long long* o = g_stamps[blockIdx.x];
        o[0] = st_top; o[1] = st_bara; o[2] = st_mfma; o[3] = st_barb; o[4] = st_gather; o[5] = st_items;
        o[6] = st_begin; o[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    __syncthreads();
    // slabs are destination-major, [n][groups][FOUT]: the reduction reads one contiguous run per row
    f32x4* slab = reinterpret_cast<f32x4*>(slabs) + (size_t)grp * kLpr + col0 / 4;
    for (int i = tid; i < a.n * LPR; i += THREADS)
        slab[(size_t)(i / LPR) * a.groups * kLpr + (i % LPR)] = acc4[i];
}

// Wt[r][col * fin + k] = sum_b att[r, b] * basis[b, k, col]       (layers.py:172-173, transposed)
// One wave = 16 relations x (kWtK k x 16 col): the B fragments are 64-byte row segments of basis,
// and every lane ends up with kWtK consecutive k of one (relation, col) -> 16-byte stores.
constexpr int kWtK = 4;

struct WtArgs {
    const float* __restrict__ att; const float* __restrict__ basis; float* __restrict__ wt;
    int relations, bases, fin, fout;
};

__global__ __launch_bounds__(256) void k_rgcn_weights_t(WtArgs g) {
    // block = 16 relations x (16 k x 16 col); wave w owns k0 + 4w .. k0 + 4w + 3.  The tile is transposed
    // through LDS so that every 64-byte run of Wt (16 consecutive k of one (relation, col)) leaves in
    // one store instruction: 16-byte pieces scattered over four waves cost a memory transaction each.
    __shared__ float tile[16][16][17];                       // [rel][col][k], padded
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c16 = lane & 15, q = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int cblocks = g.fout / 16;
    const int kb16 = (blockIdx.y / cblocks) * 16, col0 = (blockIdx.y % cblocks) * 16;
    const int k0 = kb16 + wave * kWtK;
    const int arow = row0 + c16;
    f32x4 acc[kWtK];
#pragma unroll
    for (int t = 0; t < kWtK; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // All loads are unconditional (clamped indices, zeroed by select afterwards): a conditional load
    // would be waited for one by one instead of being batched ahead of the MFMA chain.
    const int arow_c = min(arow, g.relations - 1);
    // two K steps (32 bases) per trip: all of their loads are in flight before the first MFMA
    for (int b0 = 0; b0 < g.bases; b0 += 32) {
        float av[2][4], bv[2][kWtK][4];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int b = b0 + 16 * h + 4 * q + jj, bc = min(b, g.bases - 1);
                av[h][jj] = g.att[(int64_t)arow_c * g.bases + bc];
                const float* __restrict__ bp = g.basis + ((int64_t)bc * g.fin + k0) * g.fout + col0 + c16;
#pragma unroll
                for (int t = 0; t < kWtK; ++t) bv[h][t][jj] = bp[t * g.fout];
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const bool live = (b0 + 16 * h + 4 * q + jj) < g.bases;
                av[h][jj] = (live && arow < g.relations) ? av[h][jj] : 0.f;
#pragma unroll
                for (int t = 0; t < kWtK; ++t) bv[h][t][jj] = live ? bv[h][t][jj] : 0.f;
            }
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int t = 0; t < kWtK; ++t)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h][jj], bv[h][t][jj], acc[t], 0, 0, 0);
    }
    // lane (c16, q), element i: W[row0 + 4q + i][k0 + t][col0 + c16], t = 0..kWtK-1
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < kWtK; ++t) tile[4 * q + i][c16][wave * kWtK + t] = acc[t][i];
    __syncthreads();
    // 16 rel x 16 col x 4 float4 along k = 1024 float4, four per thread; four lanes cover one 64-byte run
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int idx = it * 256 + threadIdx.x;
        const int k4 = idx & 3, col = (idx >> 2) & 15, rel = idx >> 6;
        const int row = row0 + rel;
        if (row < g.relations) {
            const float* t = &tile[rel][col][4 * k4];
            float* o = g.wt + ((int64_t)row * g.fout + col0 + col) * g.fin + kb16 + 4 * k4;
            *reinterpret_cast<f32x4*>(o) = (f32x4){t[0], t[1], t[2], t[3]};
        }
    }
}

// out[i, c] = act( (sum_g slab[g][i][c]) / max(1, indeg) + x[i] . root[:, c] + bias[c] )   (partial: raw sum)
struct FinArgs {
    const float* __restrict__ slabs; int groups; int n;
    const float* __restrict__ indeg; const float* __restrict__ x; int64_t ld_x; int fin;
    const float* __restrict__ root; const float* __restrict__ bias;
    int relu; int partial; float* out; int64_t ld_out;
    gn_side_copy side;
};

__global__ __launch_bounds__(256) void k_rgcn_slab_finalize(FinArgs a) {
    // one destination row per block.  Slab sum: 256 threads = 8 float4 columns x 32 slab partitions;
    // root term: 256 threads = 32 columns x 8 K partitions.  Every partial sum runs in a fixed
    // order (bitwise reproducible), and every load is issued before the first add.
    __shared__ f32x4 part[32][kLpr];
    __shared__ float xpart[8][kFout];
    __shared__ float row[kFout];
    const int f = threadIdx.x & (kLpr - 1), pp = threadIdx.x >> 3;
    const int i = blockIdx.x;
    if (a.side.dst && i < a.side.rows) {                       // concat slot 0: this block copies its row
        for (int c = threadIdx.x; c < a.side.cols; c += 256) {
            const float v = a.side.src[(int64_t)i * a.side.ld_src + c];
            a.side.dst[(int64_t)i * a.side.ld_dst + c] = a.side.mode ? fabsf(v) : v;
        }
    }
    // x[i] . root[:, c], K split 8 ways (unconditional clamped loads, zeroed by select)
    float xr = 0.f;
    if (!a.partial) {
        const int c = threadIdx.x & (kFout - 1), kp = threadIdx.x >> 5;
        const float* __restrict__ xi = a.x + (int64_t)i * a.ld_x;
        float xv[8], rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = min(kp + 8 * u, a.fin - 1);
            xv[u] = xi[k];
            rv[u] = a.root[k * kFout + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xr += (kp + 8 * u < a.fin) ? xv[u] * rv[u] : 0.f;
    }
    const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.slabs) + (size_t)i * a.groups * kLpr + f;
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int g = pp; g < a.groups; g += 32) s += src[(size_t)g * kLpr];
    part[pp][f] = s;
    xpart[threadIdx.x >> 5][threadIdx.x & (kFout - 1)] = xr;
    __syncthreads();
    if (threadIdx.x < kLpr) {
        f32x4 v = part[0][threadIdx.x];
#pragma unroll
        for (int k = 1; k < 32; ++k) v += part[k][threadIdx.x];
        *reinterpret_cast<f32x4*>(&row[4 * threadIdx.x]) = v;
    }
    __syncthreads();
    if (threadIdx.x < kFout) {
        const int c = threadIdx.x;
        float v = row[c];
        if (!a.partial) {
            v = v / fmaxf(a.indeg[i], 1.0f);
            float r = xpart[0][c];
#pragma unroll
            for (int k = 1; k < 8; ++k) r += xpart[k][c];
            v += r;
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

__device__ __forceinline__ int item_of(const int32_t* __restrict__ item_begin, int n_items, int p) {
    int a = 0, b = n_items;
    while (b - a > 1) {
        int mid = (a + b) >> 1;
        if (item_begin[mid] <= p) a = mid; else b = mid;
    }
    return a;
}

// item_id[p] = work item that holds position p;  cnt[item, dst] += 1
__global__ void k_item_dst_counts(const uint32_t* __restrict__ packed, const int32_t* __restrict__ item_begin,
                                  int n_items, int n, int nodes, int32_t* __restrict__ item_id,
                                  int32_t* __restrict__ cnt) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int it = item_of(item_begin, n_items, p);
    item_id[p] = it;
    atomicAdd(&cnt[(int64_t)it * nodes + (packed[p] >> 16)], 1);
}

// key = (item << 32) | ((0xffff - count) << 16) | dst : inside an item, heaviest destination first
__global__ void k_rank_keys(const int32_t* __restrict__ cnt, int64_t total, int nodes, uint64_t* __restrict__ key) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int64_t it = i / nodes;
    const uint32_t dst = (uint32_t)(i - it * nodes);
    const uint32_t c = (uint32_t)min(cnt[i], 0xffff);
    key[i] = ((uint64_t)it << 32) | ((uint64_t)(0xffffu - c) << 16) | dst;
}

// Destinations of an item are dealt to the kNB slots in boustrophedon order of their rank.
__global__ void k_deal_slots(const uint64_t* __restrict__ sorted, int64_t total, int nodes, uint8_t* __restrict__ slot_of) {
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= total) return;
    const uint64_t k = sorted[i];
    const int64_t it = (int64_t)(k >> 32);
    const uint32_t dst = (uint32_t)(k & 0xffffu);
    const int rank = (int)(i - it * nodes);
    const int r = rank % (2 * kNB);
    slot_of[it * nodes + dst] = (uint8_t)(r < kNB ? r : 2 * kNB - 1 - r);
}

// key2 = (item << 32) | (slot << 16) | dst
__global__ void k_item_keys(const uint32_t* __restrict__ packed, const int32_t* __restrict__ item_id,
                            const uint8_t* __restrict__ slot_of, int n, int nodes, uint64_t* __restrict__ key2) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint32_t dst = packed[p] >> 16;
    const int it = item_id[p];
    key2[p] = ((uint64_t)it << 32) | ((uint64_t)slot_of[(int64_t)it * nodes + dst] << 16) | dst;
}

__global__ void k_slot_offsets(const uint64_t* __restrict__ sorted, int n, int n_items, int32_t* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_items * kNB) return;
    const uint64_t target = ((uint64_t)(i / kNB) << 32) | ((uint64_t)(i % kNB) << 16);
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sorted[mid] < target) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
}

// sz[i] = length of slot list i rounded up to whole groups of 4 words (0 for the terminator)
__global__ void k_pad_sizes(const int32_t* __restrict__ slot_off, int n_slots, int32_t* __restrict__ sz) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > n_slots) return;
    sz[i] = i < n_slots ? ((slot_off[i + 1] - slot_off[i] + 3) & ~3) : 0;
}

// Writes every edge word to its padded position and marks the last edge of each (item, slot, dst) run.
__global__ void k_emit_padded(const uint64_t* __restrict__ key2_sorted, const uint32_t* __restrict__ packed_sorted,
                              const int32_t* __restrict__ slot_off, const int32_t* __restrict__ pad_off, int n,
                              uint32_t* __restrict__ out) {
    int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const uint64_t k = key2_sorted[p];
    const int sg = (int)(k >> 32) * kNB + (int)((k >> 16) & 0xffffu);
    // a run also "ends" every 64 words of its slot list: no fp32 running sum is longer than 64 addends
    // before it is folded into the accumulator row (bounds the rounding bias of hub destinations)
    const int pos = p - slot_off[sg];
    const bool end = p == n - 1 || key2_sorted[p + 1] != k || (pos & 63) == 63;
    out[pad_off[sg] + pos] = (packed_sorted[p] & 0x7fffffffu) | (end ? kEndFlag : 0u);
}

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

using gn::Scratch;   // scoped device scratch (common.h)

template <typename K, typename V>
gn_status sort_pairs(Scratch& tmp, const K* kin, K* kout, const V* vin, V* vout, size_t n, int bits, hipStream_t st) {
    size_t bytes = 0;
    GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, kin, kout, vin, vout, n, 0, bits, st));
    char* scratch = nullptr;
    GN_HIP(tmp.get(&scratch, bytes));
    GN_HIP(rocprim::radix_sort_pairs(scratch, bytes, kin, kout, vin, vout, n, 0, bits, st));
    return GN_OK;
}

gn_status sort_keys(Scratch& tmp, const uint64_t* kin, uint64_t* kout, size_t n, int bits, hipStream_t st) {
    size_t bytes = 0;
    GN_HIP(rocprim::radix_sort_keys(nullptr, bytes, kin, kout, n, 0, bits, st));
    char* scratch = nullptr;
    GN_HIP(tmp.get(&scratch, bytes));
    GN_HIP(rocprim::radix_sort_keys(scratch, bytes, kin, kout, n, 0, bits, st));
    return GN_OK;
}

template <int FIN, int RT, int COLS>
gn_status launch_main(const FastArgs& a, int groups, size_t lds_bytes, hipStream_t st) {
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_rgcn_lds<FIN, RT, COLS>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    FastDims dm;
    dm.ld_x = a.ld_x; dm.n = a.n; dm.ts = a.ts; dm.groups = groups;
    k_rgcn_lds<FIN, RT, COLS><<<groups * (kFout / COLS), 32 * COLS, lds_bytes, st>>>(a.x, a.wt, a.packed, a.slot_off, a.work,
                                                                                    a.wg_begin, a.slabs, dm);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

template <int FIN, int COLS>
gn_status launch_main_rt(const FastArgs& a, int groups, size_t lds_bytes, hipStream_t st) {
    switch ((int)gn::ceil_div(a.ts / 16, COLS / 2)) {          // row tiles per wave; COLS / 2 waves
        case 1: return launch_main<FIN, 1, COLS>(a, groups, lds_bytes, st);
        case 2: return launch_main<FIN, 2, COLS>(a, groups, lds_bytes, st);
        default: return launch_main<FIN, 3, COLS>(a, groups, lds_bytes, st);
    }
}

template <int FIN>
gn_status launch_main_cols(const FastArgs& a, int cols, int groups, size_t lds_bytes, hipStream_t st) {
    return cols == kFout ? launch_main_rt<FIN, kFout>(a, groups, lds_bytes, st)
                         : launch_main_rt<FIN, kFout / 2>(a, groups, lds_bytes, st);
}

}  // namespace

// Builds the relation-major work items of the shard.  Leaves plan->fast_ok = 0 when the graph
// does not qualify (too many nodes for the LDS accumulator, too many relations for the key).
gn_status gn_rgcn_build_fast_segments(gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst,
                                      const std::vector<int64_t>& ranges, hipStream_t st) {
    plan->fast_ok = 0;
    const int64_t N = plan->num_nodes, R = plan->num_relations, E = plan->shard_edges;
    if (gn::fast_paths_disabled() || N < 1 || N > 32767 || R < 1 || E < 1) return GN_OK;
    const FastGeom g = geometry(N);
    if (g.tiles == 0 || R * g.tiles >= (1 << 20)) return GN_OK;

    Scratch tmp;
    GN_HIP(tmp.reserve((size_t)48 * (size_t)E + (size_t)8 * (size_t)(R * g.tiles + R) + ((size_t)1 << 20)));
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
    gn_status s = sort_pairs(tmp, key, key_sorted, packed, packed_sorted, (size_t)E, bits_for(n_seg), st);
    if (s != GN_OK) return s;
    k_lower_bounds_u32<<<(int)gn::ceil_div(n_seg + 1, 256), 256, 0, st>>>(key_sorted, (int)E, n_seg, seg_off);
    GN_LAUNCH_CHECK();
    std::vector<int32_t> seg(n_seg + 1);
    GN_HIP(hipMemcpyAsync(seg.data(), seg_off, (n_seg + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("  lds: segment keys + sort (sync)");

    // work items: every non-empty (relation, tile) segment, cut into chunks of <= kItemEdges edges
    std::vector<int32_t> item_rel, item_tile, item_begin;
    for (int sg = 0; sg < n_seg; ++sg) {
        for (int32_t b = seg[sg]; b < seg[sg + 1]; b += kItemEdges) {
            item_rel.push_back(sg / g.tiles);
            item_tile.push_back(sg % g.tiles);
            item_begin.push_back(b);
        }
    }
    const int n_items = (int)item_rel.size();
    item_begin.push_back((int32_t)E);
    if ((int64_t)n_items * kNB >= (1ll << 31) || (int64_t)n_items * N >= (1ll << 31)) return GN_OK;
    // Balancing unit = piece: up to `piece_chunks` consecutive items of one (relation, tile) segment.
    // A workgroup that runs them back to back builds the segment's H tile once.
    int64_t total_cost = 0;
    for (int sg = 0; sg < n_seg; ++sg)
        if (seg[sg + 1] > seg[sg]) total_cost += kItemOverhead + (seg[sg + 1] - seg[sg]);
    const int groups0 = std::min(kGroups, std::max(n_items, 1));
    const int64_t piece_cap = std::max<int64_t>(kItemOverhead + kItemEdges, total_cost / groups0 / 4);
    const int piece_chunks = (int)std::max<int64_t>(1, (piece_cap - kItemOverhead) / kItemEdges);
    std::vector<int32_t> piece_first, piece_count;
    std::vector<int64_t> piece_cost;
    for (int i = 0; i < n_items;) {
        int jn = i + 1;
        while (jn < n_items && jn - i < piece_chunks && item_rel[jn] == item_rel[i] && item_tile[jn] == item_tile[i]) ++jn;
        piece_first.push_back(i);
        piece_count.push_back(jn - i);
        piece_cost.push_back((int64_t)kItemOverhead + (item_begin[jn] - item_begin[i]));
        i = jn;
    }
    const int n_pieces = (int)piece_first.size();
    const int groups = std::min(groups0, std::max(n_pieces, 1));
    // longest-processing-time assignment of pieces to the persistent workgroups
    std::vector<int> order(n_pieces);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return piece_cost[x] > piece_cost[y]; });
    std::vector<std::vector<int32_t>> bins(groups);
    {
        std::vector<std::pair<int64_t, int>> heap;   // min-heap over (load, group)
        for (int gidx = 0; gidx < groups; ++gidx) heap.emplace_back(0, gidx);
        auto cmp = [](const std::pair<int64_t, int>& x, const std::pair<int64_t, int>& y) { return x > y; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (int pc : order) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto& top = heap.back();
            for (int k = 0; k < piece_count[pc]; ++k) bins[top.second].push_back(piece_first[pc] + k);
            top.first += piece_cost[pc];
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    GN_LAP("  lds: items, pieces, LPT (host)");
    // destination -> slot assignment inside every item, balanced by edge count
    const int64_t cells = (int64_t)n_items * N;
    // (everything the rest of the build asks for in one more block of the scratch)
    GN_HIP(tmp.reserve((size_t)cells * 40 + (size_t)E * 40 + (size_t)n_items * (kNB + 1) * 8 + ((size_t)2 << 20)));
    int32_t *item_begin_dev, *item_id, *cnt;
    uint64_t *rkey, *rkey_sorted, *key2, *key2_sorted;
    uint8_t* slot_of;
    GN_HIP(tmp.get(&item_begin_dev, n_items + 1));
    GN_HIP(tmp.get(&item_id, E));
    GN_HIP(tmp.get(&cnt, cells));
    GN_HIP(tmp.get(&rkey, cells));
    GN_HIP(tmp.get(&rkey_sorted, cells));
    GN_HIP(tmp.get(&slot_of, cells));
    GN_HIP(tmp.get(&key2, E));
    GN_HIP(tmp.get(&key2_sorted, E));
    GN_HIP(hipMemcpyAsync(item_begin_dev, item_begin.data(), (n_items + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemsetAsync(cnt, 0, cells * sizeof(int32_t), st));
    k_item_dst_counts<<<(int)gn::ceil_div(E, 256), 256, 0, st>>>(packed_sorted, item_begin_dev, n_items, (int)E, (int)N,
                                                                item_id, cnt);
    GN_LAUNCH_CHECK();
    k_rank_keys<<<(int)gn::ceil_div(cells, 256), 256, 0, st>>>(cnt, cells, (int)N, rkey);
    GN_LAUNCH_CHECK();
    s = sort_keys(tmp, rkey, rkey_sorted, (size_t)cells, 32 + bits_for(n_items), st);
    if (s != GN_OK) return s;
    k_deal_slots<<<(int)gn::ceil_div(cells, 256), 256, 0, st>>>(rkey_sorted, cells, (int)N, slot_of);
    GN_LAUNCH_CHECK();
    k_item_keys<<<(int)gn::ceil_div(E, 256), 256, 0, st>>>(packed_sorted, item_id, slot_of, (int)E, (int)N, key2);
    GN_LAUNCH_CHECK();
    uint32_t* packed_slot;
    int32_t *slot_off, *pad_sz;
    const int n_slots = n_items * kNB;
    GN_HIP(tmp.get(&packed_slot, E));
    GN_HIP(tmp.get(&slot_off, n_slots + 1));
    GN_HIP(tmp.get(&pad_sz, n_slots + 1));
    s = sort_pairs(tmp, key2, key2_sorted, packed_sorted, packed_slot, (size_t)E, 32 + bits_for(n_items), st);
    if (s != GN_OK) return s;
    k_slot_offsets<<<(int)gn::ceil_div((int64_t)n_slots + 1, 256), 256, 0, st>>>(key2_sorted, (int)E, n_items, slot_off);
    GN_LAUNCH_CHECK();
    // every slot list padded to whole groups of 4 words: padded offsets = exclusive scan of the sizes
    k_pad_sizes<<<(int)gn::ceil_div((int64_t)n_slots + 1, 256), 256, 0, st>>>(slot_off, n_slots, pad_sz);
    GN_LAUNCH_CHECK();
    GN_HIP(plan->seg_begin.alloc((size_t)n_slots + 1));
    {
        size_t bytes = 0;
        GN_HIP(rocprim::exclusive_scan(nullptr, bytes, pad_sz, plan->seg_begin.p, 0, (size_t)n_slots + 1,
                                       rocprim::plus<int32_t>(), st));
        char* scratch = nullptr;
        GN_HIP(tmp.get(&scratch, bytes));
        GN_HIP(rocprim::exclusive_scan(scratch, bytes, pad_sz, plan->seg_begin.p, 0, (size_t)n_slots + 1,
                                       rocprim::plus<int32_t>(), st));
    }
    std::vector<int32_t> item_pad(n_items + 1);
    GN_HIP(hipMemcpy2DAsync(item_pad.data(), sizeof(int32_t), plan->seg_begin.p, kNB * sizeof(int32_t), sizeof(int32_t),
                            (size_t)n_items + 1, hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("  lds: slots, two sorts, scan (sync)");
    const int64_t padded = item_pad[n_items];
    GN_HIP(plan->packed.alloc(padded + 8));
    GN_HIP(hipMemsetAsync(plan->packed.p, 0, (padded + 8) * sizeof(uint32_t), st));   // padding word: dst 0, src 0, no flag
    k_emit_padded<<<(int)gn::ceil_div(E, 256), 256, 0, st>>>(key2_sorted, packed_slot, slot_off, plan->seg_begin.p, (int)E,
                                                            plan->packed.p);
    GN_LAUNCH_CHECK();
    // per workgroup: all items of one source tile together (the X fragments stay in registers),
    // relation order inside a tile (W_r reuse in L2)
    std::vector<int32_t> wg_begin(groups + 1, 0);
    std::vector<WorkDesc> work;
    work.reserve(n_items);
    for (int gidx = 0; gidx < groups; ++gidx) {
        std::sort(bins[gidx].begin(), bins[gidx].end(), [&](int32_t x, int32_t y) {
            return item_tile[x] != item_tile[y] ? item_tile[x] < item_tile[y] : x < y;
        });
        for (int32_t it : bins[gidx]) {
            WorkDesc w = {item_rel[it], item_tile[it], item_pad[it], item_pad[it + 1] - item_pad[it], it, 0, 0, 0};
            work.push_back(w);
        }
        wg_begin[gidx + 1] = (int32_t)work.size();
    }

    GN_HIP(plan->wg_begin.alloc(groups + 1));
    GN_HIP(plan->wg_items.alloc(work.size() * (sizeof(WorkDesc) / sizeof(int32_t))));
    GN_HIP(hipMemcpyAsync(plan->wg_begin.p, wg_begin.data(), (groups + 1) * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->wg_items.p, work.data(), work.size() * sizeof(WorkDesc), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));       // host vectors go out of scope after this
    GN_LAP("  lds: emit + descriptors (sync)");
    plan->n_seg = n_items;
    plan->fast_groups = groups;
    plan->fast_ts = g.ts;
    plan->fast_ts_pad = g.ts;
    plan->fast_lds_bytes = g.lds_bytes;
    plan->fast_cols = g.cols;
    plan->fast_ok = 1;
    return GN_OK;
}

// Multi-GPU finalisation: the all-reduced [n, 32] sum is one "slab"; mean / root / bias / activation and the
// concat slot in a single launch (the general path needs a GEMM launch for the root term first).
bool gn_rgcn_fast_finalize_applicable(int64_t fin, int64_t fout, int64_t ld_summed, const void* summed) {
    return !gn::fast_paths_disabled() && fout == kFout && fin >= 1 && fin <= 64 && ld_summed == kFout &&
           (reinterpret_cast<uintptr_t>(summed) & 15) == 0;
}

gn_status gn_rgcn_fast_finalize(const gn_rgcn_plan* plan, const float* summed, const float* x, int64_t ld_x, int64_t fin,
                                const float* root, const float* bias, int relu, float* out, int64_t ld_out,
                                const gn_side_copy& side, hipStream_t st) {
    FinArgs f;
    f.slabs = summed; f.groups = 1; f.n = (int)plan->num_nodes; f.indeg = plan->indeg.p; f.x = x; f.ld_x = ld_x;
    f.fin = (int)fin; f.root = root; f.bias = bias; f.relu = relu; f.partial = 0; f.out = out; f.ld_out = ld_out;
    f.side = side;
    k_rgcn_slab_finalize<<<(int)plan->num_nodes, 256, 0, st>>>(f);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

bool gn_rgcn_fast_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    if (!plan->fast_ok || gn::fast_paths_disabled()) return false;
    return fout == kFout && (fin == 16 || fin == 32 || fin == 48 || fin == 64) && bases >= 1;
}

size_t gn_rgcn_fast_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    const size_t w = ((size_t)plan->num_relations * fin * fout * sizeof(float) + 255) & ~size_t(255);
    return w + (size_t)plan->fast_groups * plan->num_nodes * fout * sizeof(float);
}

static gn_status gn_rgcn_fast_weights(const gn_rgcn_plan* plan, int64_t fin, const float* basis, const float* att,
                                      int64_t bases, int64_t fout, void* ws, hipStream_t st) {
    GN_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "workspace must be 16-byte aligned");
    // K7: W_r = sum_b att[r,b] basis[b], stored transposed per relation   (layers.py:172-173)
    WtArgs wa;
    wa.att = att; wa.basis = basis; wa.wt = static_cast<float*>(ws); wa.relations = (int)plan->num_relations;
    wa.bases = (int)bases; wa.fin = (int)fin; wa.fout = (int)fout;
    dim3 wgrid((unsigned)gn::ceil_div(plan->num_relations, 16), (unsigned)((fin / 16) * (fout / 16)));
    k_rgcn_weights_t<<<wgrid, 256, 0, st>>>(wa);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_rgcn_fast_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin,
                               const float* basis, const float* att, int64_t bases, const float* root,
                               const float* bias, int64_t fout, int relu, int partial, float* out,
                               int64_t ld_out, const gn_side_copy& side, void* ws, size_t ws_bytes, hipStream_t st) {
    GN_REQUIRE(ld_x % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned with ld_x %% 4 == 0");
    const int64_t N = plan->num_nodes, R = plan->num_relations;
    GN_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "workspace must be 16-byte aligned");
    float* Wt = static_cast<float*>(ws);
    const size_t w_bytes = ((size_t)R * fin * fout * sizeof(float) + 255) & ~size_t(255);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + w_bytes);
    {
        gn_status ws_status = gn_rgcn_fast_weights(plan, fin, basis, att, bases, fout, ws, st);
        if (ws_status != GN_OK) return ws_status;
    }
    FastArgs a;
    a.x = x; a.ld_x = ld_x; a.n = (int)N; a.wt = Wt; a.packed = plan->packed.p; a.slot_off = plan->seg_begin.p;
    a.work = reinterpret_cast<const WorkDesc*>(plan->wg_items.p); a.wg_begin = plan->wg_begin.p;
    a.ts = plan->fast_ts; a.slabs = slabs;
    const int groups = plan->fast_groups;
    gn_status s;
    switch (fin) {
        case 16: s = launch_main_cols<16>(a, plan->fast_cols, groups, plan->fast_lds_bytes, st); break;
        case 32: s = launch_main_cols<32>(a, plan->fast_cols, groups, plan->fast_lds_bytes, st); break;
        case 48: s = launch_main_cols<48>(a, plan->fast_cols, groups, plan->fast_lds_bytes, st); break;
        default: s = launch_main_cols<64>(a, plan->fast_cols, groups, plan->fast_lds_bytes, st); break;
    }
    if (s != GN_OK) return s;
    FinArgs f;
    f.slabs = slabs; f.groups = groups; f.n = (int)N; f.indeg = plan->indeg.p; f.x = x; f.ld_x = ld_x;
    f.fin = (int)fin; f.root = root; f.bias = bias; f.relu = relu; f.partial = partial; f.out = out; f.ld_out = ld_out;
    f.side = side;
    k_rgcn_slab_finalize<<<(int)N, 256, 0, st>>>(f);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

#ifdef GN_STAMPS
extern "C" __attribute__((visibility("default"))) int gn_debug_read_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 2 * kGroups * 8);
}
extern "C" __attribute__((visibility("default"))) int gn_debug_read_wave_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wave_stamps), sizeof(unsigned long long) * 2 * kGroups * 16 * 4);
}
#endif
