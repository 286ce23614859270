// Multi-relational layer for small supervertices, aggregate-then-transform with the sums kept in the
// matrix-core accumulators (the drug supervertex of PoSE: n_d = 645 nodes, 48 -> 32 features, ~10^3
// relations, millions of edges).
//
//   out[i] = (sum_r (sum_{e in r, dst=i} x[src_e]) W_r) / max(1, indeg_i) + x[i] root (+ bias)
//          = (sum_{e: dst=i} x[src_e] W_{r(e)}) / ...                     (gripnet/layers.py:165-197)
//
// The node table x ([n, in], 124 KB at 645 x 48) lives in every CU's LDS.  A wave owns kTpg tiles
// of 16 DESTINATION rows for the whole kernel and walks a list of (relation, row group) units:
//   (1) gather: the four lanes 4 r .. 4 r + 3 add x[src] for every edge (relation, src -> row r of
//       the tile) from LDS into registers, 16 rows at a time; one ds_read_b128 per lane and 64
//       features of a row, the four lanes of a row covering one 64-byte bank slot;
//   (2) the 16 x in matrix A_r of per-destination sums is moved to the A-operand layout of the
//       matrix instruction once per tile (ds_bpermute) and transformed: acc[tile] += A_r W_r.  The
//       accumulator registers carry the sum over relations, so nothing is scattered, nothing is
//       read-modify-written and no barrier sits between a workgroup's prologue and its epilogue.
//       The product runs as three bf16 MFMAs on split operands (hi.hi + hi.lo + lo.hi, fp32
//       accumulate; GN_ACC_EXACT=1 selects the fp32 matrix instruction): the fp32 MFMA runs at the
//       fp32 vector rate and shares the SIMD's issue with the gather's adds.
// Edge lists reach the waves as one private, contiguous stream per wave of 64-byte blocks: 2
// "iterations" x 16 rows of uint16 source ids (GN_ACC_BLOCK_ITERS; 4 = 128-byte blocks); rows without an edge in an iteration point at one of
// four zero rows of the LDS table (no branches, no bounds).  Which edge of a row goes into which
// iteration is decided at plan time for the LDS: per iteration, the four rows of a ds_read_b128
// access group read four different bank slots wherever the graph allows it (conflict cycles are
// ~16 % of the LDS cycles on pose0-syn; a random order gives ~3x the conflict-free time).  The
// stream is staged through a 1 KB LDS window per wave, refilled from registers that were loaded a
// window ahead.
// Workgroups = row groups x slabs: the waves of a workgroup share a row group and split its units
// (longest-processing-time in two levels at plan time); they fold their accumulators through LDS
// in wave order, and the per-workgroup slabs are summed in slab order by k_rgcn_slab_finalize:
// fixed summation order, bitwise reproducible.
//
// HBM traffic: 2 bytes per edge slot (the stream) + the W_r fragments, which every XCD's L2 fetches
// once.  Matrix-core time of the fp32 form: 2 n in out flops per relation = 1.9 GFLOP / 157 TFLOP/s
// = 12 us on pose0-syn; of the split form 3/16 of that.  What bounds the kernel now is the LDS
// bandwidth of the gather (192 bytes per edge slot, ~2.4 slots per edge after padding).
#include "rgcn_weights.cuh"

#include <rocprim/device/device_radix_sort.hpp>

#include <algorithm>
#include <numeric>
#include <vector>

// rgcn_tf.hip
// rgcn_fast.hip
gn_status gn_rgcn_slab_finalize_launch(const gn_rgcn_plan* plan, const float* slabs, int groups, const float* x,
                                       int64_t ld_x, int64_t fin, const float* root, const float* bias, int relu,
                                       int partial, float* out, int64_t ld_out, const gn_side_copy& side, hipStream_t st);

namespace {

using gn_rw::f32x4;
using gn_rw::u32x4;
using gn_rw::u32x2;
using gn_rw::split2;
using gn_rw::WeightsFragArgs;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#ifndef GN_ACC_TPG
#define GN_ACC_TPG 3
#endif
#ifndef GN_ACC_WAVES
#define GN_ACC_WAVES 12
#endif
constexpr int kTpg = GN_ACC_TPG;            // 16-row tiles per wave = accumulator tiles kept in registers
constexpr int kWaves = GN_ACC_WAVES;         // waves per workgroup (one workgroup per CU: the node table fills its LDS)
constexpr int kThreads = kWaves * 64;
#ifndef GN_ACC_ITER_CAP
#define GN_ACC_ITER_CAP 32
#endif
constexpr int kIterCap = GN_ACC_ITER_CAP;       // iterations per unit and tile: longer (relation, row) lists are cut into chunks
#ifndef GN_ACC_BLOCK_ITERS
#define GN_ACC_BLOCK_ITERS 2
#endif
constexpr int kBI = GN_ACC_BLOCK_ITERS;      // gather iterations per stream block (4: 128-byte blocks, 2: 64-byte blocks; a tile's
                                             // iterations are rounded up to whole blocks: 2 pads less, 101.5 vs 102.3 us per pose0-syn step; 1: no further gain)
static_assert(kBI == 4 || kBI == 2 || kBI == 1, "stream blocks hold one, two or four iterations");
constexpr int kStage = 8;          // 128-byte units per LDS window of a wave (1 KB)
constexpr int kStageBlocks = kStage * 4 / kBI;   // stream blocks per window
constexpr int kFoutAcc = 32;       // out_features (slab layout of k_rgcn_slab_finalize)
constexpr size_t kLdsBudget = 159 * 1024;
// cost model of the plan-time balancing, in cycles of the wave's SIMD
#ifndef GN_ACC_TILE_COST
#define GN_ACC_TILE_COST 2200
#define GN_ACC_BLOCK_COST 800
#define GN_ACC_EDGE_COST 0
#endif
// (a least-squares fit of the waves' loop times: 0.33 us per block, 0.92 us per non-empty tile - the tile pays for the
// wait on its unit's W_r fragments as well as for the move to the MFMA layout, the split and the 10 bf16 MFMAs; tile
// costs of 550 / 1000 / 1500 / 2200 / 3000 cycles gave kernels of 26.5 / 25.9 / 25.5 / 25.45 / 25.6 us on pose0-syn)
constexpr int kTileCost = GN_ACC_TILE_COST;
constexpr int kBlockCost = GN_ACC_BLOCK_COST;  // one stream block: 12 LDS reads + adds, with the workgroup's other waves on the LDS
constexpr int kEdgeCost = GN_ACC_EDGE_COST;    // what a real edge adds to its block (nothing since the plan orders edges for the LDS)

// Diagnostic builds only (make MODE=n -> libgripnet_hip_mode<n>.so, never the product library):
// bit 0 = no MFMA phase, bit 1 = no table gather (the stream is still read), bit 2 = conflict-free gather.
#ifndef GN_ACC_MODE
#define GN_ACC_MODE 0
#endif

#ifdef GN_STAMPS
// Diagnostic build only (make STAMPS=1): per-wave times, never part of the product library.
__device__ unsigned long long g_acc_stamps[4096][8];
#endif

struct alignas(16) AccUnit {
    int32_t rel;
    uint8_t blocks[4];             // stream blocks (4 iterations each) of the unit's tiles; 0 = tile has no edge
    int32_t pad0, pad1;
};
static_assert(kTpg <= 4, "AccUnit holds four tile block counts");

// Workgroup index -> (row group, slab).  Workgroups go to the eight XCDs round-robin (XCD = index % 8) and every XCD has
// its own L2: slab s < 8 (G / 8) sits on XCD s % 8, so the plan can keep a relation's units on the slabs of one XCD
// and its W_r fragments are fetched into one L2 instead of eight (62 of the 85 MB the entry point moved).  The
// G % 8 slabs that are left take what does not fit.
__host__ __device__ inline void acc_block_to_group(int block, int Q, int G, int& qg, int& slab) {
    const int g8 = (G / 8) * 8, main_blocks = Q * g8;
    if (block < main_blocks) {
        const int x = block & 7, t = block >> 3;
        qg = t % Q;
        slab = (t / Q) * 8 + x;
    } else {
        const int r = block - main_blocks;
        qg = r % Q;
        slab = g8 + r / Q;
    }
}
__host__ inline int acc_group_to_block(int qg, int slab, int Q, int G) {
    const int g8 = (G / 8) * 8;
    if (slab < g8) return (((slab >> 3) * Q + qg) << 3) + (slab & 7);
    return Q * g8 + (slab - g8) * Q + qg;
}

struct AccDims { int64_t ld_x; int n; int tiles; int q_groups; int slabs; };


// SPLIT = false: the transform runs on v_mfma_f32_16x16x4_f32 (exact fp32, the fp32 vector rate).
// SPLIT = true:  both operands are split into bf16 pairs (v = hi + lo, |v - hi - lo| <= 2^-18 |v|) and the
//                transform is three v_mfma_f32_16x16x32_bf16 products hi.hi + hi.lo + lo.hi accumulated in fp32
//                (the dropped lo.lo term and the residuals are ~2^-17 of a product: a few 1e-6 on the layer's
//                output, against the 1e-4 contract) at 3/16 of the fp32 matrix time.
template <int FIN, int NT, bool SPLIT>
__global__ __launch_bounds__(kThreads) void k_rgcn_acc(const float* __restrict__ x, const f32x4* __restrict__ wfrag,
                                                      const u32x2* __restrict__ stream,
                                                      const AccUnit* __restrict__ units,
                                                      const int32_t* __restrict__ wave_units,
                                                      const uint32_t* __restrict__ wave_stream,
                                                      float* __restrict__ slabs, AccDims a) {
    constexpr int KQ = FIN / 4;              // features per lane quarter
    constexpr int KP = KQ / 4;               // 16-byte pieces of them
    constexpr int M = (KQ + 7) / 8;          // bf16 MFMAs (8 k per lane) that cover a quarter
    constexpr int BV = SPLIT ? NT * M * 2 : NT * KP;   // 16-byte W fragments per lane and relation
    constexpr int XL = FIN / 4;              // float4 per row of x
    // LDS rows are whole 64-byte bank slots, an odd number of them per row: the four lanes that gather one edge
    // read one slot (see below), and the plan orders every row's edges so that the four edges a 16-lane access
    // group of ds_read_b128 works on sit in four different slots wherever the graph allows it.
    constexpr int XS = 4 * ((FIN / 16) | 1);
    extern __shared__ f32x4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;       // MFMA layout: row lane % 16, k group lane / 16
    const int grow = lane >> 2, gq = lane & 3;        // gather layout: row lane / 4, quarter lane % 4
    int qg, slab;
    acc_block_to_group((int)blockIdx.x, a.q_groups, a.slabs, qg, slab);
    const int wv = __builtin_amdgcn_readfirstlane(blockIdx.x * kWaves + wave);

#ifdef GN_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_gather = 0, st_mfma = 0, st_blocks = 0;
#endif
    // ---- first loads of the table fill: issued before everything else, because vector loads retire in order and
    //      the stream / W loads below come from further away (HBM / Infinity Cache) than the L2-resident table ----
    // (every workgroup reads the same table at the same time: each starts at its own offset, so that they do not all
    // queue on the same L2 channel)
    constexpr int FILL = 12;
    const int fill_total = a.n * XL, fill_rot = (int)((blockIdx.x * 977u) % (unsigned)fill_total);
    auto fill_index = [&](int i) { i += fill_rot; return i < fill_total ? i : i - fill_total; };
    f32x4 fill[FILL];
#pragma unroll
    for (int k = 0; k < FILL; ++k) {
        const int i = fill_index(min(k * kThreads + tid, fill_total - 1));
        const int r = i / XL, c = i - r * XL;
        fill[k] = *reinterpret_cast<const f32x4*>(x + (int64_t)r * a.ld_x + 4 * c);
    }
    // ---- this wave's work list and the head of its stream: in flight while the table is filled ----
    int u = wave_units[wv];
    const int u_end = wave_units[wv + 1];
    // The stream is staged through a private LDS window of kStage blocks per wave: one 16-byte load per lane
    // brings 8 blocks, the next window waits in registers while the current one is consumed (a register
    // ring rotated with moves would have to wait for its youngest load on every trip).
    constexpr int SV = kStage / 8;                                // 16-byte loads per lane and window
    const u32x4* __restrict__ gp = reinterpret_cast<const u32x4*>(stream) + (size_t)wave_stream[wv] * (2 * kBI) + lane;
    u32x4 pre[SV];
#pragma unroll
    for (int i = 0; i < SV; ++i) pre[i] = gp[i * 64];             // the stream ends with two spare windows
    gp += kStage * 8;
    auto load_b = [&](int rel, f32x4 (&b)[BV]) {
#if GN_ACC_MODE & 32     // diagnostic: every unit reads one of eight W_r (all of them L2-resident)
        rel &= 7;
#endif
        const f32x4* __restrict__ wr = wfrag + (size_t)rel * (BV * 64) + lane;
#pragma unroll
        for (int i = 0; i < BV; ++i) b[i] = wr[i * 64];
    };
    AccUnit d = units[u < u_end ? u : 0];
    f32x4 bfrag[BV];                         // fp32: [nt][p];  split: [nt][m][hi, lo]
    load_b(d.rel, bfrag);

    // ---- node table -> LDS, plus the zero row that padded slots point at ----
    // (the first FILL loads per thread were issued at the top - the whole table at n = 645; a load-store loop would
    // pay one L2 round trip per trip)
    for (int base = 0; base < fill_total; base += FILL * kThreads) {
        if (base > 0) {
#pragma unroll
            for (int k = 0; k < FILL; ++k) {
                const int i = fill_index(min(base + k * kThreads + tid, fill_total - 1));
                const int r = i / XL, c = i - r * XL;
                fill[k] = *reinterpret_cast<const f32x4*>(x + (int64_t)r * a.ld_x + 4 * c);
            }
        }
#pragma unroll
        for (int k = 0; k < FILL; ++k) {
            const int j = base + k * kThreads + tid;
            const int i = fill_index(min(j, fill_total - 1));
            const int r = i / XL, c = i - r * XL;
            if (j < fill_total) lds4[r * XS + c] = fill[k];
        }
    }
    if (tid < 4 * XS) lds4[a.n * XS + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};   // four zero rows, one per bank slot, for padded edge slots
    u32x4* stage = reinterpret_cast<u32x4*>(lds4 + (a.n + 4) * XS) + wave * (kStage * 8);
    const u32x2* stage2 = reinterpret_cast<const u32x2*>(stage) + grow;          // kBI == 4: 8 bytes per row and block
    const uint32_t* stage1 = reinterpret_cast<const uint32_t*>(stage) + grow;    // kBI == 2: 4 bytes per row and block
    const uint16_t* stage0 = reinterpret_cast<const uint16_t*>(stage) + grow;    // kBI == 1: 2 bytes per row and block
    __syncthreads();
#ifdef GN_STAMPS
    const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
    const int st_u0 = u;
#endif

    f32x4 acc[kTpg][NT];
#pragma unroll
    for (int t = 0; t < kTpg; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // Gather layout: the four lanes 4 r .. 4 r + 3 work on the edge of row r and read the four consecutive 16-byte
    // pieces 4 p + gq of x[src] (one 64-byte bank slot per instruction and edge, so a 16-lane access group touches
    // four slots); lane (r, gq) therefore sums features 16 p + 4 gq + c.  The sums are moved to the MFMA layout
    // once per tile (ds_bpermute), not once per edge.
    const f32x4* __restrict__ xq = lds4 + gq;
    const int from_lane4 = 4 * (4 * n16 + kq);                    // MFMA lane (row n16, k group kq) takes gather lane 4 n16 + kq
    // wnext is the stream word of the next block to consume, read from the window one block ahead (the stream
    // is contiguous across tiles and units, so the look-ahead never stops); pos is its block in the window.
    int pos = 0;
    auto refill = [&]() {                                         // publish the waiting window, fetch the one after
#pragma unroll
        for (int i = 0; i < SV; ++i) stage[i * 64 + lane] = pre[i];
#pragma unroll
        for (int i = 0; i < SV; ++i) pre[i] = gp[i * 64];
        gp += kStage * 8;
        pos = 0;
    };
    refill();
    u32x2 wnext;                                                  // LDS is in order per wave: no wait after the stores
    if constexpr (kBI == 4) wnext = stage2[0]; else if constexpr (kBI == 2) wnext = (u32x2){stage1[0], 0u}; else wnext = (u32x2){stage0[0], 0u};

    for (; u < u_end; ++u) {
        const AccUnit dn = units[u + 1 < u_end ? u + 1 : u];     // scalar load, a whole unit ahead
#pragma unroll
        for (int t = 0; t < kTpg; ++t) {
            const int nb = d.blocks[t];                           // wave-uniform
            if (nb == 0) continue;
            f32x4 s[KP];
#pragma unroll
            for (int p = 0; p < KP; ++p) s[p] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef GN_STAMPS
            const unsigned long long st_a = __builtin_amdgcn_s_memtime();
            st_blocks += nb;
#endif
            for (int k = 0; k < nb; ++k) {
                const u32x2 w = wnext;
                if (++pos == kStageBlocks) refill();
                if constexpr (kBI == 4) wnext = stage2[pos * 16]; else if constexpr (kBI == 2) wnext.x = stage1[pos * 16]; else wnext.x = stage0[pos * 16];
#if GN_ACC_MODE & 4      // every lane reads the same table row: the gather without bank conflicts
                const uint32_t s0 = pos, s1 = pos + 1, s2 = pos + 2, s3 = pos + 3 + (w.x & w.y & 1u);
#else
                const uint32_t s0 = w.x & 0xffffu, s1 = w.x >> 16, s2 = w.y & 0xffffu, s3 = w.y >> 16;
#endif
                const f32x4* __restrict__ r0 = xq + s0 * XS;
                const f32x4* __restrict__ r1 = xq + s1 * XS;
                const f32x4* __restrict__ r2 = xq + s2 * XS;
                const f32x4* __restrict__ r3 = xq + s3 * XS;
#if GN_ACC_MODE & 2
                s[0][0] += __uint_as_float(s0 + s1 + s2 + s3);
#else
                if constexpr (kBI == 4) {
#pragma unroll
                    for (int p = 0; p < KP; ++p) s[p] += (r0[4 * p] + r1[4 * p]) + (r2[4 * p] + r3[4 * p]);
                } else if constexpr (kBI == 2) {
#pragma unroll
                    for (int p = 0; p < KP; ++p) s[p] += r0[4 * p] + r1[4 * p];
                } else {
#pragma unroll
                    for (int p = 0; p < KP; ++p) s[p] += r0[4 * p];
                }
#endif
            }
#ifdef GN_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
#pragma unroll
            for (int p = 0; p < KP; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    s[p][c] = __int_as_float(__builtin_amdgcn_ds_bpermute(from_lane4, __float_as_int(s[p][c])));
#if GN_ACC_MODE & 1
#pragma unroll
            for (int p = 0; p < KP; ++p) acc[t][0] += s[p] * bfrag[p];
#else
            if constexpr (SPLIT) {
                // lane (row, kg) supplies A[row][k = 8 kg + j] = s[8 m + j] = feature 16 ((8 m + j) / 4) + 4 kg + (8 m + j) % 4.
                // A quarter that ends half way through
                // its last MFMA (in = 16, 48) packs that one as A = {hi, lo} against B = {hi, hi} and B = {lo, 0}:
                // hi.hi + lo.hi in one instruction, hi.lo in the other.
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const bool half = 8 * m + 4 == KQ;
                    u32x4 ah, al;
#pragma unroll
                    for (int h = 0; h < 4; ++h) {
                        const int e = 8 * m + 2 * h;              // flat index into the quarter
                        uint32_t hi = 0u, lo = 0u;
                        if (e < KQ) split2(s[e / 4][e % 4], s[e / 4][e % 4 + 1], hi, lo);
                        ah[h] = hi; al[h] = lo;
                    }
                    if (half) { ah[2] = al[0]; ah[3] = al[1]; }
                    const bf16x8 xh = __builtin_bit_cast(bf16x8, ah), xl = __builtin_bit_cast(bf16x8, al);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const bf16x8 bh = __builtin_bit_cast(bf16x8, bfrag[(nt * M + m) * 2]);
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, bfrag[(nt * M + m) * 2 + 1]);
                        if (!half) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh, acc[t][nt], 0, 0, 0);
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl, acc[t][nt], 0, 0, 0);
                        acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh, acc[t][nt], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int p = 0; p < KP; ++p)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(s[p][jj], bfrag[nt * KP + p][jj], acc[t][nt], 0, 0, 0);
            }
#endif
#ifdef GN_STAMPS
            asm volatile("s_nop 0" :: "v"(acc[t][0]), "v"(acc[t][NT - 1]) : "memory");   // the chain has retired
            const unsigned long long st_c = __builtin_amdgcn_s_memtime();
            st_gather += st_b - st_a; st_mfma += st_c - st_b;
#endif
        }
        d = dn;
        // W fragments of the next unit: they land during its first gather (vmcnt retires in order, so a second
        // register set loaded a unit ahead would be waited for at the same place and only cost registers)
        load_b(d.rel, bfrag);
    }

#ifdef GN_STAMPS
    const unsigned long long st_t2 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- fold the waves of the workgroup in wave order, leave as one slab ----
    __syncthreads();                                              // every wave is done with the table
    float* red = reinterpret_cast<float*>(lds4);                  // [wave][tile][16 rows][NT * 16]
    constexpr int RW = NT * 16;
#pragma unroll
    for (int t = 0; t < kTpg; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 4; ++i)                            // D fragment: lane (n16, kq) holds row 4 kq + i, column n16
                red[((wave * kTpg + t) * 16 + 4 * kq + i) * RW + nt * 16 + n16] = acc[t][nt][i];
    __syncthreads();
    constexpr int OUT4 = kTpg * 16 * (RW / 4);
    if (tid < OUT4) {
        const int row = tid / (RW / 4), c4 = tid % (RW / 4);
        f32x4 v = lds4[tid];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) v += lds4[w * OUT4 + tid];
        const int grow = qg * (kTpg * 16) + row;
        if (grow < a.n) reinterpret_cast<f32x4*>(slabs)[((size_t)grow * a.slabs + slab) * (RW / 4) + c4] = v;
    }
#ifdef GN_STAMPS
    if (lane == 0 && wv < 4096) {
        unsigned long long* o = g_acc_stamps[wv];
        o[0] = st_t0; o[1] = st_t1; o[2] = st_t2; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = st_gather; o[5] = st_mfma; o[6] = (unsigned long long)(u_end - st_u0); o[7] = st_blocks;
    }
#endif
}

__global__ __launch_bounds__(256) void k_rgcn_weights_frag(WeightsFragArgs g) { gn_rw::rgcn_weights_frag_body(g, (int)blockIdx.x); }

// ---- plan construction ----------------------------------------------------------------------------
__global__ void k_acc_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                           const int64_t* __restrict__ range_start, int R, int64_t lo, int64_t hi, int64_t N,
                           uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
    for (int64_t e = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < hi; e += (int64_t)gridDim.x * blockDim.x) {
        int a = 0, b = R;
        while (b - a > 1) {
            int mid = (a + b) >> 1;
            if (range_start[mid] <= e) a = mid; else b = mid;
        }
        key[e - lo] = (uint32_t)((int64_t)a * N + dst[e]);       // ids validated by the general plan builder
        val[e - lo] = (uint32_t)src[e];
    }
}

__global__ void k_acc_rowptr(const uint32_t* __restrict__ sorted, int n, int64_t count, int32_t* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i > count) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (sorted[mid] < (uint32_t)i) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
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

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}


// 16-byte W fragments per lane and relation (see k_rgcn_acc)
size_t acc_w_bytes(int64_t relations, int64_t fin, int64_t fout) {
    const int64_t kq = fin / 4, m = (kq + 7) / 8;
    const int64_t frags = std::max<int64_t>((fout / 16) * m * 2, (fout / 16) * (kq / 4));     // either layout fits
    return ((size_t)relations * frags * 64 * 16 + 255) & ~size_t(255);
}

bool acc_disabled() {
    if (gn::fast_paths_disabled()) return true;
    const char* e = getenv("GN_DISABLE_ACC");
    return e && e[0] == '1';
}

template <int FIN, bool SPLIT>
gn_status launch_acc(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, const f32x4* wfrag, float* slabs,
                     size_t lds_bytes, hipStream_t st) {
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_rgcn_acc<FIN, kFoutAcc / 16, SPLIT>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    AccDims dm;
    dm.ld_x = ld_x; dm.n = (int)plan->num_nodes; dm.tiles = plan->acc_tiles; dm.q_groups = plan->acc_q; dm.slabs = plan->acc_g;
    k_rgcn_acc<FIN, kFoutAcc / 16, SPLIT><<<plan->acc_q * plan->acc_g, kThreads, lds_bytes, st>>>(
        x, wfrag, reinterpret_cast<const u32x2*>(plan->acc_stream.p), reinterpret_cast<const AccUnit*>(plan->acc_units.p),
        plan->acc_wave_units.p, plan->acc_wave_stream.p, slabs, dm);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

// Builds the per-wave unit lists and edge streams of the shard.  Leaves plan->acc_ok = 0 when the graph does
// not qualify (node ids beyond 16 bits of stream word, nothing to do).
gn_status gn_rgcn_build_acc_plan(gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst,
                                 const std::vector<int64_t>& ranges, hipStream_t st) {
    plan->acc_ok = 0;
    const int64_t N = plan->num_nodes, R = plan->num_relations, E = plan->shard_edges;
    if (acc_disabled() || N < 1 || N > 2500 || R < 1 || E < 1) return GN_OK;
    const int tiles = (int)gn::ceil_div(N, 16);
    const int Q = (int)gn::ceil_div(tiles, kTpg);
    int cus = 256;                                                   // MI355X; asked of the device the plan is built on
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    const int G = std::max(1, cus / Q);
    if (R * N >= ((int64_t)1 << 31) || R * tiles >= ((int64_t)1 << 28)) return GN_OK;

    Scratch tmp;
    int64_t* starts_dev;
    uint32_t *key, *key_sorted, *val, *val_sorted;
    int32_t* rowptr;
    GN_HIP(tmp.get(&starts_dev, R + 1));
    GN_HIP(tmp.get(&key, E));
    GN_HIP(tmp.get(&key_sorted, E));
    GN_HIP(tmp.get(&val, E));
    GN_HIP(tmp.get(&val_sorted, E));
    GN_HIP(tmp.get(&rowptr, R * N + 1));
    std::vector<int64_t> starts(R + 1, plan->input_edges);
    for (int64_t r = 0; r < R; ++r) starts[r] = ranges[2 * r];
    GN_HIP(hipMemcpyAsync(starts_dev, starts.data(), (R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    k_acc_keys<<<gn::stream_grid(E, 256), 256, 0, st>>>(src, dst, starts_dev, (int)R, plan->edge_lo, plan->edge_hi, N, key, val);
    GN_LAUNCH_CHECK();
    {
        size_t bytes = 0;
        GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key, key_sorted, val, val_sorted, (size_t)E, 0, bits_for(R * N), st));
        char* scratch = nullptr;
        GN_HIP(tmp.get(&scratch, bytes));
        GN_HIP(rocprim::radix_sort_pairs(scratch, bytes, key, key_sorted, val, val_sorted, (size_t)E, 0, bits_for(R * N), st));
    }
    k_acc_rowptr<<<(int)gn::ceil_div(R * N + 1, 256), 256, 0, st>>>(key_sorted, (int)E, R * N, rowptr);
    GN_LAUNCH_CHECK();
    std::vector<int32_t> rp(R * N + 1);
    std::vector<uint32_t> srcs(E);                                    // sources in (relation, destination) order
    GN_HIP(hipMemcpyAsync(rp.data(), rowptr, (R * N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipMemcpyAsync(srcs.data(), val_sorted, (size_t)E * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));

    // iterations of every (relation, tile) = its longest (relation, destination) run
    std::vector<int32_t> iters((size_t)R * tiles, 0);
    for (int64_t r = 0; r < R; ++r)
        for (int64_t i = 0; i < N; ++i) {
            const int32_t c = rp[r * N + i + 1] - rp[r * N + i];
            int32_t& m = iters[r * tiles + i / 16];
            m = std::max(m, c);
        }
    // units = (relation, row group, chunk of <= kIterCap iterations per tile)
    struct Unit { int32_t rel, chunk; uint8_t blocks[4]; int64_t cost; };
    std::vector<std::vector<Unit>> per_q(Q);
    for (int64_t r = 0; r < R; ++r)
        for (int q = 0; q < Q; ++q) {
            int32_t longest = 0;
            for (int t = 0; t < kTpg; ++t) {
                const int tile = q * kTpg + t;
                if (tile < tiles) longest = std::max(longest, iters[r * tiles + tile]);
            }
            const int64_t row0 = (int64_t)q * kTpg * 16, row1 = std::min<int64_t>(N, row0 + kTpg * 16);
            for (int c = 0; c * kIterCap < longest; ++c) {
                Unit un = {(int32_t)r, c, {0, 0, 0, 0}, 0};
                int64_t edges = 0;
                for (int64_t i = row0; i < row1; ++i)
                    edges += std::min(kIterCap, std::max(0, rp[r * N + i + 1] - rp[r * N + i] - c * kIterCap));
                for (int t = 0; t < kTpg; ++t) {
                    const int tile = q * kTpg + t;
                    const int32_t it = tile < tiles ? std::min(kIterCap, std::max(0, iters[r * tiles + tile] - c * kIterCap)) : 0;
                    un.blocks[t] = (uint8_t)((it + kBI - 1) / kBI);
                    if (it > 0) un.cost += kTileCost + (int64_t)un.blocks[t] * (kBlockCost * kBI / 4);
                }
                un.cost += edges * kEdgeCost;
                per_q[q].push_back(un);
            }
        }
    // Longest-processing-time assignment in two levels: a row group's units to its G workgroups (the LDS is
    // the shared resource of the gather, so whole workgroups have to carry equal loads), then a workgroup's
    // units to its waves.  A wave walks its units in relation order (all waves sweep the relations together:
    // W_r stays L2-resident).
    const int n_waves = Q * G * kWaves;
    std::vector<std::vector<Unit>> per_wave(n_waves);
    auto lpt = [](const std::vector<Unit>& us, int bins, std::vector<std::vector<Unit>>& out) {
        out.assign(bins, {});
        std::vector<int> order(us.size());
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return us[x].cost > us[y].cost; });
        std::vector<std::pair<int64_t, int>> heap;
        for (int w = 0; w < bins; ++w) heap.emplace_back(0, w);
        auto cmp = [](const std::pair<int64_t, int>& x, const std::pair<int64_t, int>& y) { return x > y; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (int idx : order) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto& top = heap.back();
            out[top.second].push_back(us[idx]);
            top.first += us[idx].cost;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    };
    // Relations are dealt to eight classes (one per XCD) by their total cost, once for all row groups; a unit may go
    // to the slabs of its relation's class (slab % 8 = class, slab < 8 (G / 8)) or to one of the G % 8 spare slabs,
    // whichever is least loaded.  With fewer than eight slabs there are no classes.
    const int g8 = (G / 8) * 8;
    std::vector<int> rel_class(R, 0);
    if (g8 > 0) {
        std::vector<int64_t> rel_cost(R, 0);
        for (int q = 0; q < Q; ++q)
            for (const Unit& un : per_q[q]) rel_cost[un.rel] += un.cost;
        std::vector<int> order(R);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return rel_cost[x] > rel_cost[y]; });
        int64_t load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r : order) {
            const int c = (int)(std::min_element(load, load + 8) - load);
            rel_class[r] = c;
            load[c] += rel_cost[r];
        }
    }
    for (int q = 0; q < Q; ++q) {
        std::vector<std::vector<Unit>> per_wg(G), per_w;
        if (g8 == 0) {
            lpt(per_q[q], G, per_wg);
        } else {
            const std::vector<Unit>& us = per_q[q];
            std::vector<int> order(us.size());
            std::iota(order.begin(), order.end(), 0);
            std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return us[x].cost > us[y].cost; });
            std::vector<int64_t> load(G, 0);
            for (int idx : order) {
                const int c = rel_class[us[idx].rel];
                int best = c;
                for (int sl = c + 8; sl < g8; sl += 8)
                    if (load[sl] < load[best]) best = sl;
                for (int sl = g8; sl < G; ++sl)
                    if (load[sl] < load[best]) best = sl;
                per_wg[best].push_back(us[idx]);
                load[best] += us[idx].cost;
            }
        }
        for (int slab = 0; slab < G; ++slab) {
            lpt(per_wg[slab], kWaves, per_w);
            const int block = acc_group_to_block(q, slab, Q, G);
            for (int wi = 0; wi < kWaves; ++wi) per_wave[block * kWaves + wi] = std::move(per_w[wi]);
        }
    }
    std::vector<AccUnit> units;
    std::vector<int32_t> wave_units(n_waves + 1, 0);
    std::vector<uint32_t> wave_stream(n_waves, 0);
    // (relation, tile) -> first entry of its per-chunk block offsets
    std::vector<int32_t> chunk_base((size_t)R * tiles + 1, 0);
    for (int64_t k = 0; k < R * tiles; ++k) chunk_base[k + 1] = chunk_base[k] + (int32_t)gn::ceil_div(iters[k], kIterCap);
    std::vector<uint32_t> chunk_off(chunk_base[R * tiles] + 1, 0);
    uint64_t blocks_total = 0;
    for (int w = 0; w < n_waves; ++w) {
        auto& us = per_wave[w];
        std::sort(us.begin(), us.end(), [](const Unit& x, const Unit& y) { return x.rel != y.rel ? x.rel < y.rel : x.chunk < y.chunk; });
        int q, slab_of_w;
        acc_block_to_group(w / kWaves, Q, G, q, slab_of_w);
        wave_stream[w] = (uint32_t)blocks_total;
        for (const Unit& un : us) {
            AccUnit au = {un.rel, {un.blocks[0], un.blocks[1], un.blocks[2], un.blocks[3]}, 0, 0};
            units.push_back(au);
            for (int t = 0; t < kTpg; ++t) {
                if (un.blocks[t] == 0) continue;
                const int tile = q * kTpg + t;
                chunk_off[chunk_base[(int64_t)un.rel * tiles + tile] + un.chunk] = (uint32_t)blocks_total;
                blocks_total += un.blocks[t];
            }
        }
        wave_units[w + 1] = (int32_t)units.size();
    }
    if (blocks_total + 3 * kStageBlocks >= ((uint64_t)1 << 26)) return GN_OK;      // 64 B-word index must fit 32 bits
    if (units.empty()) units.push_back(AccUnit{0, {0, 0, 0, 0}, 0, 0});

    // ---- the edge streams.  Which edge of a row goes into which iteration is free (the order of a sum), so it is
    //      chosen for the LDS: in the gather layout of k_rgcn_acc the four lanes of a row read one 64-byte bank
    //      slot of x[src] per instruction, slot = (src * odd) mod 4, and ds_read_b128 serves the wave in four
    //      access groups of four rows each.  Per iteration and access group the rows pick edges whose sources
    //      fall into different slots where they can; rows with edges to spare sit an iteration out (their slot
    //      then points at a zero row in a free slot) rather than collide. ----
    const int64_t words = (int64_t)(blocks_total + 3 * kStageBlocks) * (8 * kBI);   // uint32 words: 16 rows x kBI / 2 per block
    std::vector<uint16_t> stream16((size_t)words * 2);
    static const int kGroupRows[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};   // rows = lanes / 4 of the b128 access groups
    int pos_in_group[16];
    for (int g = 0; g < 4; ++g)
        for (int k = 0; k < 4; ++k) pos_in_group[kGroupRows[g][k]] = k;
    for (size_t i = 0; i < stream16.size(); ++i) stream16[i] = (uint16_t)(N + pos_in_group[(i / kBI) & 15]);   // padding: four zero rows, one per slot
    // (every (relation, tile) writes its own stream blocks: the relations are spread over the plan builders' threads)
    gn::parallel_for(R, 4, [&](int64_t rel0, int64_t rel1) {
        int cur[16][4], end[16][4], rem[16];
        std::vector<uint16_t> rowbuf[16];
        for (int64_t r = rel0; r < rel1; ++r)
            for (int tile = 0; tile < tiles; ++tile) {
                const int32_t L = iters[r * tiles + tile];
                if (L == 0) continue;
                for (int i = 0; i < 16; ++i) {
                    const int64_t d = (int64_t)tile * 16 + i;
                    rowbuf[i].clear();
                    rem[i] = 0;
                    for (int c = 0; c < 4; ++c) cur[i][c] = end[i][c] = 0;
                    if (d >= N) continue;
                    const int32_t b0 = rp[r * N + d], b1 = rp[r * N + d + 1];
                    rem[i] = b1 - b0;
                    int cnt[4] = {0, 0, 0, 0};
                    for (int32_t p = b0; p < b1; ++p) ++cnt[srcs[p] & 3];
                    int off = 0;
                    for (int c = 0; c < 4; ++c) { cur[i][c] = off; off += cnt[c]; end[i][c] = off; }
                    rowbuf[i].resize(rem[i]);
                    int fillp[4] = {cur[i][0], cur[i][1], cur[i][2], cur[i][3]};
                    for (int32_t p = b0; p < b1; ++p) rowbuf[i][fillp[srcs[p] & 3]++] = (uint16_t)srcs[p];
                }
                const int32_t cb = chunk_base[r * tiles + tile];
                for (int32_t it = 0; it < L; ++it) {
                    const int32_t left = L - it;             // iterations left, this one included
                    const size_t blk = (size_t)chunk_off[cb + it / kIterCap] + (size_t)((it % kIterCap) / kBI);
                    uint16_t* out = stream16.data() + blk * (16 * kBI) + (it % kBI);         // + row * kBI
                    for (int g = 0; g < 4; ++g) {
                        int order[4] = {kGroupRows[g][0], kGroupRows[g][1], kGroupRows[g][2], kGroupRows[g][3]};
                        // rows that cannot sit out first, then the fuller ones
                        std::sort(order, order + 4, [&](int x, int y) {
                            const bool mx = rem[x] >= left, my = rem[y] >= left;
                            return mx != my ? mx : rem[x] > rem[y];
                        });
                        unsigned used = 0;
                        bool idle[4] = {false, false, false, false};
                        for (int k = 0; k < 4; ++k) {
                            const int i = order[k];
                            if (rem[i] == 0) { idle[k] = true; continue; }
                            int best = -1, bestcnt = 0;
                            for (int c = 0; c < 4; ++c) {
                                const int n_c = end[i][c] - cur[i][c];
                                if (n_c > bestcnt && !((used >> c) & 1)) { best = c; bestcnt = n_c; }
                            }
                            if (best < 0) {
                                if (rem[i] < left) { idle[k] = true; continue; }      // can wait for a free slot
                                for (int c = 0; c < 4; ++c) {
                                    const int n_c = end[i][c] - cur[i][c];
                                    if (n_c > bestcnt) { best = c; bestcnt = n_c; }
                                }
                            }
                            out[i * kBI] = rowbuf[i][cur[i][best]++];
                            --rem[i];
                            used |= 1u << best;
                        }
                        for (int k = 0; k < 4; ++k) {        // rows that sit out: a zero row in a slot nobody reads
                            if (!idle[k]) continue;
                            const int i = order[k];
                            for (int z = 0; z < 4; ++z) {
                                const unsigned c = (unsigned)(N + z) & 3u;
                                if (!((used >> c) & 1)) { out[i * kBI] = (uint16_t)(N + z); used |= 1u << c; break; }
                            }
                        }
                    }
                }
            }
    });
    GN_HIP(plan->acc_stream.alloc((size_t)words));
    GN_HIP(plan->acc_units.alloc(units.size() * (sizeof(AccUnit) / sizeof(int32_t))));
    GN_HIP(plan->acc_wave_units.alloc(wave_units.size()));
    GN_HIP(plan->acc_wave_stream.alloc(wave_stream.size()));
    GN_HIP(hipMemcpyAsync(plan->acc_stream.p, stream16.data(), (size_t)words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->acc_units.p, units.data(), units.size() * sizeof(AccUnit), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->acc_wave_units.p, wave_units.data(), wave_units.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->acc_wave_stream.p, wave_stream.data(), wave_stream.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));       // host vectors and scratch go out of scope after this
    plan->acc_tiles = tiles; plan->acc_q = Q; plan->acc_g = G;
    plan->acc_blocks = (int64_t)blocks_total;
    plan->acc_ok = 1;
    return GN_OK;
}

static size_t acc_lds_bytes(int64_t n, int64_t fin) {
    const size_t table = (size_t)(n + 4) * 16 * ((fin / 16) | 1) * sizeof(float) + (size_t)kWaves * kStage * 128;
    const size_t fold = (size_t)kWaves * kTpg * 16 * kFoutAcc * sizeof(float);
    return std::max(table, fold);
}

bool gn_rgcn_acc_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    if (!plan->acc_ok || acc_disabled()) return false;
    if (fout != kFoutAcc || !(fin == 16 || fin == 32 || fin == 48) || bases < 1) return false;
    return acc_lds_bytes(plan->num_nodes, fin) <= kLdsBudget;
}

size_t gn_rgcn_acc_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    const size_t groups = (size_t)plan->acc_g;
    return acc_w_bytes(plan->num_relations, fin, fout) + groups * plan->num_nodes * fout * sizeof(float);
}

// Arguments of the weights body for this plan and these parameters (also used by cowork.hip, which runs the body
// inside another launch); returns the number of 256-thread blocks it needs.
int gn_rgcn_acc_weights_args(const gn_rgcn_plan* plan, int64_t fin, const float* basis, const float* att, int64_t bases,
                             int64_t fout, void* ws, int exact, gn_rw::WeightsFragArgs* g) {
    const int64_t R = plan->num_relations;
    g->att = att; g->basis = basis; g->wfrag = static_cast<f32x4*>(ws);
    g->relations = (int)R; g->bases = (int)bases; g->fin = (int)fin; g->fout = (int)fout;
    g->tasks = (int)(gn::ceil_div(R, 16) * 4 * ((fin / 4 + 7) / 8) * (fout / 16));
    g->split = exact ? 0 : 1;
    return (int)gn::ceil_div(g->tasks, 4);
}

gn_status gn_rgcn_acc_weights(const gn_rgcn_plan* plan, int64_t fin, const float* basis, const float* att,
                              int64_t bases, int64_t fout, void* ws, int exact, hipStream_t st) {
    GN_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "workspace must be 16-byte aligned");
    WeightsFragArgs g;
    const int blocks = gn_rgcn_acc_weights_args(plan, fin, basis, att, bases, fout, ws, exact, &g);
    k_rgcn_weights_frag<<<blocks, 256, 0, st>>>(g);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_rgcn_acc_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* basis,
                              const float* att, int64_t bases, const float* root, const float* bias, int64_t fout,
                              int relu, int partial, int weights_ready, int exact, float* out, int64_t ld_out,
                              const gn_side_copy& side, void* ws, size_t ws_bytes, hipStream_t st) {
    GN_REQUIRE(ld_x % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0, "x must be 16-byte aligned with ld_x %% 4 == 0");
    GN_REQUIRE((reinterpret_cast<uintptr_t>(ws) & 15) == 0, "workspace must be 16-byte aligned");
    const int64_t R = plan->num_relations;
    f32x4* wfrag = static_cast<f32x4*>(ws);
    float* slabs = reinterpret_cast<float*>(static_cast<char*>(ws) + acc_w_bytes(R, fin, fout));
    const bool split = !exact;
    if (!weights_ready) {
        gn_status ws_status = gn_rgcn_acc_weights(plan, fin, basis, att, bases, fout, ws, exact, st);
        if (ws_status != GN_OK) return ws_status;
    }
    const size_t lds = acc_lds_bytes(plan->num_nodes, fin);
    gn_status s;
    switch ((int)fin * 2 + (split ? 1 : 0)) {
        case 32: s = launch_acc<16, false>(plan, x, ld_x, wfrag, slabs, lds, st); break;
        case 33: s = launch_acc<16, true>(plan, x, ld_x, wfrag, slabs, lds, st); break;
        case 64: s = launch_acc<32, false>(plan, x, ld_x, wfrag, slabs, lds, st); break;
        case 65: s = launch_acc<32, true>(plan, x, ld_x, wfrag, slabs, lds, st); break;
        case 96: s = launch_acc<48, false>(plan, x, ld_x, wfrag, slabs, lds, st); break;
        default: s = launch_acc<48, true>(plan, x, ld_x, wfrag, slabs, lds, st); break;
    }
    if (s != GN_OK) return s;
    // (summing the slabs inside the kernel, by the last workgroup of every row group behind an agent-scope hand-off,
    // was measured: +8 us on the kernel's tail against the 4.5 us launch it removes)
    return gn_rgcn_slab_finalize_launch(plan, slabs, plan->acc_g, x, ld_x, fin, root, bias, relu, partial, out, ld_out, side, st);
}

#ifdef GN_STAMPS
extern "C" __attribute__((visibility("default"))) int gn_debug_read_acc_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_acc_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif
