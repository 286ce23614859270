// Multi-relational layer for small supervertices, transform-then-gather inside one workgroup per relation
// (the drug supervertex of PoSE: n_d = 645 nodes, 48 -> 32 features, ~10^3 relations, millions of edges).
//
//   out[i] = (sum_{e: dst=i} x[src_e] W_{r(e)}) / max(1, indeg_i) + x[i] root (+ bias)      (gripnet/layers.py:165-197)
//
// k_rgcn_acc (rgcn_acc.hip) sums x[src] per (relation, destination) first and transforms the sums; every non-empty
// (relation, 16-row tile) then pays a move to the MFMA layout, a bf16 split and a wait for W_r, which measured at 60 %
// of its loop on pose0-syn (830 of the 964 relations have fewer than four edges per destination).  This kernel turns
// the order round: a workgroup takes a whole relation (or a slice of a large one),
//   (1) transform: H_r = x W_r for all nodes on the matrix cores.  The node table never enters LDS: every wave keeps
//       the A-operand fragments of its three 16-row tiles of x (bf16 hi / lo pairs) in registers for the whole kernel;
//       W_r arrives as the B fragments k_rgcn_weights_frag writes (rgcn_weights.cuh), five bf16 MFMAs per 16 x 16
//       tile (hi.hi + hi.lo + lo.hi in fp32 accumulators); H_r (n x 32 fp32, 82 KB at n = 645) goes to LDS;
//   (2) gather: the four lanes of a quad own up to three DESTINATION rows for the whole kernel (the sums over
//       relations stay in registers); per edge the quad adds the 128-byte row H_r[src] to the row's accumulator, two
//       ds_read_b128 per lane.  No per-tile work is left, an edge costs 128 bytes of LDS reads instead of 192, and
//       nothing is scattered, read-modify-written or atomically added.
// The 16 quads of a wave run in lock step: per relation and row slot the wave makes as many iterations as its longest
// (relation, destination) run, rounded up to an even number; rows without an edge in an iteration read a zero row.
// Rows are dealt to (wave, slot) cells by total in-degree, so the rows of a cell are of a kind.
//
// LDS image of H_r.  Row i, column half h (16 floats = one 64-byte bank slot) sits at
//       (i >> 1) * 256 + ((q(i) ^ (2 * (i & 1) + h)) * 64,       q(i) = (i >> 2) & 3
// so that (a) the accumulator layout of the MFMA (lane group g holds rows 4 g + j) writes four different slots per
// ds_write_b32, and (b) a quad reads half 0 at word * 64 and half 1 at (word * 64) ^ 64, word = (i >> 1) << 2 | colour,
// colour(i) = q(i) ^ 2 (i & 1): the four quads of a ds_read_b128 access group are conflict-free when their sources have
// four different colours, which the plan arranges wherever the graph allows it (the order of the addends is free).
//
// Edge lists reach a wave as 16-bit words, one per quad and iteration, 64 bytes per pair of iterations; a wave's words
// for one relation slice (at most 64 iterations = 2 KB) are fetched while the previous slice is gathered and parked in
// a private LDS window.  A workgroup leaves one slab [n x 32]; k_rgcn_slab_finalize sums the slabs in slab order: fixed
// summation order, bitwise reproducible.
#include "rgcn_weights.cuh"

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

using gn_rw::f32x4;
using gn_rw::u32x4;
using gn_rw::split2;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef GN_TF_WAVES
#define GN_TF_WAVES 16
#endif
constexpr int kWaves = GN_TF_WAVES;          // waves per workgroup, one workgroup per CU
constexpr int kThreads = kWaves * 64;
constexpr int kTpw = 48 / kWaves;            // 16-row tiles of x per wave (A operands in registers)
constexpr int kSlots = 48 / kWaves;          // destination rows per quad
static_assert(kWaves == 16 || kWaves == 12, "48 tiles over the waves");
constexpr int kQuads = kWaves * 16;
constexpr int kMaxNodes = kWaves * kTpw * 16;     // 768
constexpr int kFout = 32;
constexpr int kWinSteps = 32;                // pairs of iterations per window: 64 iterations, 2 KB per wave
constexpr int kWinBytes = kWinSteps * 64 + 192;   // + three steps of slack: the gather reads words two steps ahead
#ifndef GN_TF_CHUNK
#define GN_TF_CHUNK (GN_TF_WAVES == 16 ? 20 : 16)
#endif
constexpr int kChunk = GN_TF_CHUNK;          // edges of one (relation, destination) per slice: kSlots * kChunk <= 64 iterations
static_assert(kChunk % 2 == 0 && kSlots * kChunk <= 2 * kWinSteps, "a slice has to fit the window");
// cost model of the plan-time balancing (cycles of a CU)
#ifndef GN_TF_PART_COST
#define GN_TF_PART_COST 2000
#define GN_TF_ITER_COST 10
#define GN_TF_WAVE_ITER_COST 80
#endif

#ifdef GN_STAMPS
__device__ unsigned long long g_tf_stamps[4096][12];
#endif

struct TfDims { int64_t ld_x; int n; int tiles; int groups; };

__host__ __device__ inline uint32_t tf_colour(uint32_t i) { return ((i >> 2) & 3u) ^ (2u * (i & 1u)); }
__host__ __device__ inline uint32_t tf_word(uint32_t i) { return ((i >> 1) << 2) | tf_colour(i); }

template <int FIN>
__global__ __launch_bounds__(kThreads) void k_rgcn_tf(const float* __restrict__ x, const u32x4* __restrict__ wfrag,
                                                     const u32x4* __restrict__ stream,
                                                     const int32_t* __restrict__ wg_parts,
                                                     const int32_t* __restrict__ part_rel,
                                                     const uint2* __restrict__ part_wave,
                                                     const int32_t* __restrict__ cell_row,
                                                     float* __restrict__ slabs, TfDims a) {
    constexpr int KQ = FIN / 4;              // features per lane quarter
    constexpr int KP = KQ / 4;               // 16-byte pieces of them
    constexpr int M = (KQ + 7) / 8;          // bf16 MFMAs (8 k per lane) that cover a quarter
    constexpr int NT = kFout / 16;
    constexpr int BV = NT * M * 2;           // 16-byte W fragments per lane and relation
    extern __shared__ char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, kq = lane >> 4;        // MFMA layout: row lane % 16, k group lane / 16
    const int quad = lane >> 2, qj = lane & 3;        // gather layout: quad lane / 4, 16-byte piece lane % 4
    const int wg = blockIdx.x;
#ifdef GN_STAMPS
    const unsigned long long st_t0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_mfma = 0, st_gather = 0, st_wait = 0;
#endif

    // ---- this wave's tiles of x as A operands (held for the whole kernel) ----
    // lane (row, kg) supplies A[row][k = 8 kg + j] = feature 16 ((8 m + j) / 4) + 4 kg + (8 m + j) % 4 (the k order of
    // the W fragments).  A quarter that ends half way through its last MFMA packs that one as A = {hi, lo} against
    // B = {hi, hi} and B = {lo, 0}.
    u32x4 ah[kTpw][M], al[kTpw][M];
    int p = wg_parts[wg];
    const int p_end = wg_parts[wg + 1];
    {
        f32x4 s[kTpw][KP];
#pragma unroll
        for (int t = 0; t < kTpw; ++t) {
            const int row = (wave + kWaves * t) * 16 + n16;
            const float* __restrict__ xr = x + (int64_t)min(row, a.n - 1) * a.ld_x + 4 * kq;
#pragma unroll
            for (int p = 0; p < KP; ++p) s[t][p] = *reinterpret_cast<const f32x4*>(xr + 16 * p);
        }
#pragma unroll
        for (int t = 0; t < kTpw; ++t) {
            const bool live = (wave + kWaves * t) * 16 + n16 < a.n;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const bool half = 8 * m + 4 == KQ;
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    const int e = 8 * m + 2 * h;
                    uint32_t hi = 0u, lo = 0u;
                    if (e < KQ) split2(live ? s[t][e / 4][e % 4] : 0.f, live ? s[t][e / 4][e % 4 + 1] : 0.f, hi, lo);
                    ah[t][m][h] = hi; al[t][m][h] = lo;
                }
                if (half) { ah[t][m][2] = al[t][m][0]; ah[t][m][3] = al[t][m][1]; }
            }
        }
    }
    // zero tile behind the table: rows that padded slots point at, one per colour
    if (tid < 128) reinterpret_cast<f32x4*>(lds + (size_t)a.tiles * 2048)[tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
    char* const win = lds + (size_t)(a.tiles + 1) * 2048 + wave * kWinBytes;
    for (int i = lane; i < kWinBytes / 16; i += 64) reinterpret_cast<u32x4*>(win)[i] = (u32x4){0u, 0u, 0u, 0u};   // word 0 = row 0: every word the look-ahead can see is a row
    // W_r fragments of one relation: BV KB behind the windows, fetched by LDS-DMA (1 KB per wave instruction) by the
    // first BV waves while the previous slice is gathered; every wave reads them from there.  (The DMA operands are a
    // scalar base + one shared 32-bit lane offset, made opaque per use: hipcc would otherwise hoist a 64-bit per-lane
    // address out of the slice loop, spill it and wait on the reload at the top of every slice.)
    char* const bbuf = lds + (size_t)(a.tiles + 1) * 2048 + kWaves * kWinBytes;
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds);
    const uint32_t bb_s = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)(bbuf - lds) + (uint32_t)wave * 1024u);
    const uint32_t lane16 = (uint32_t)lane * 16u;
#define GN_TF_FETCH_W(rel)                                                                                               \
    if (wave < BV) {                                                                                                     \
        uint32_t l16 = lane16;                                                                                           \
        asm volatile("" : "+v"(l16));                                                                                    \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(wfrag + (size_t)(rel) * (BV * 64) + wave * 64) + l16), \
                                         (__attribute__((address_space(3))) void*)(uintptr_t)bb_s, 16, 0, 0);            \
    }

    f32x4 acc0[kSlots], acc1[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) { acc0[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc1[s] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
    const uint32_t gather_off = lds_base + ((uint32_t)qj << 4);
    // row 16 tile + 4 kq + i, column 16 nt + n16 of the image: (8 tile + 2 kq + i / 2) * 256 + ((kq ^ (2 (i & 1) + nt)) << 6) + 4 n16
    const uint32_t h_lane = (uint32_t)(2 * kq) * 256u + ((uint32_t)kq << 6) + (uint32_t)n16 * 4u;
    int my_tiles = 0;                                                  // this wave's tiles wave, wave + kWaves, ... below a.tiles
#pragma unroll
    for (int t = 0; t < kTpw; ++t) my_tiles += (wave + kWaves * t < a.tiles) ? 1 : 0;

    // first slice: relation, this wave's words and the W fragments
    uint2 pw = make_uint2(0u, 0u);
    u32x4 pre0, pre1;
    {
        int rel = 0;
        if (p < p_end) {
            rel = part_rel[p];
            pw = part_wave[p * kWaves + wave];
        }
        GN_TF_FETCH_W(rel);
        const u32x4* __restrict__ sp = stream + pw.x + lane;
        pre0 = sp[0];
        pre1 = sp[64];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                               // fragments and zero tile are in place
    }
#ifdef GN_STAMPS
    const unsigned long long st_t1 = __builtin_amdgcn_s_memrealtime();
#endif

    for (; p < p_end; ++p) {
        const uint32_t counts = pw.y;                                  // steps (pairs of iterations) per row slot, 8 bits each
        const uint32_t steps = (counts & 255u) + ((counts >> 8) & 255u) + ((counts >> 16) & 255u) + (counts >> 24);
        // ---- this wave's words -> its window ----
        reinterpret_cast<u32x4*>(win)[lane] = pre0;
        if (steps > 16) reinterpret_cast<u32x4*>(win)[64 + lane] = pre1;
#ifdef GN_STAMPS
        const unsigned long long st_a = __builtin_amdgcn_s_memtime();
#endif
        // ---- H_r = x W_r, this wave's tiles: one (hi, lo) fragment pair of W_r at a time, applied to every tile ----
        {
            const u32x4* __restrict__ bsrc = reinterpret_cast<const u32x4*>(bbuf) + lane;
            f32x4 cp[kTpw][NT];
#pragma unroll
            for (int t = 0; t < kTpw; ++t)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) cp[t][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const bool half = 8 * m + 4 == KQ;
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, bsrc[((nt * M + m) * 2) * 64]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, bsrc[((nt * M + m) * 2 + 1) * 64]);
#pragma unroll
                    for (int t = 0; t < kTpw; ++t) {                   // (tiles past the table are zero fragments)
                        const bf16x8 xh = __builtin_bit_cast(bf16x8, ah[t][m]), xl = __builtin_bit_cast(bf16x8, al[t][m]);
                        if (!half) cp[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh, cp[t][nt], 0, 0, 0);
                        cp[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl, cp[t][nt], 0, 0, 0);
                        cp[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh, cp[t][nt], 0, 0, 0);
                    }
                }
            // D fragment: lane (n16, kq) holds row 16 tile + 4 kq + i, column 16 nt + n16
            uint32_t hb = h_lane;                                      // opaque: 24 hoisted addresses would be spilled
            asm volatile("" : "+v"(hb));
#pragma unroll
            for (int t = 0; t < kTpw; ++t) {
                if (t >= my_tiles) continue;                           // wave-uniform
                const uint32_t tb = hb + (uint32_t)(wave + kWaves * t) * 2048u;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        *reinterpret_cast<float*>(lds + ((tb ^ (uint32_t)((2 * (i & 1) + nt) << 6)) + (uint32_t)(i >> 1) * 256u)) = cp[t][nt][i];
            }
        }
#ifdef GN_STAMPS
        const unsigned long long st_b = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();                                               // H_r is complete
#ifdef GN_STAMPS
        const unsigned long long st_c = __builtin_amdgcn_s_memtime();
#endif
        // ---- next slice: relation, words, W fragments (they land while this slice is gathered) ----
        {
            const int pn = p + 1 < p_end ? p + 1 : p;
            pw = part_wave[pn * kWaves + wave];
            GN_TF_FETCH_W(part_rel[pn]);
            const u32x4* __restrict__ sp = stream + pw.x + lane;
            pre0 = sp[0];
            pre1 = sp[64];
        }
        // ---- gather: every quad adds H_r[src] for the edges of its rows.  A step is two iterations; while one
        //      iteration is added up the reads of the next one are in flight (every row slot has an even number of
        //      iterations, so the two register sets keep their roles across the slots), and a step's words are read a
        //      step ahead.  Words past the wave's last step name valid rows (the window has slack).
        //      The LDS reads and their waits are written out: left to itself hipcc 7.2 waits for ALL outstanding reads
        //      before the first add of an iteration (lgkmcnt(0)), which serialises the two sets ----
        {
            uint32_t wp = lds_base + (uint32_t)(win - lds) + (uint32_t)quad * 4u;
            uint32_t pair, next;
            f32x4 x0, x1, y0, y1;
#define GN_TF_ISSUE(word, v0, v1)                                                                      \
            {                                                                                          \
                const uint32_t w6 = (word) << 6;                                                       \
                asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"                              \
                             : "=&v"(v0), "=&v"(v1) : "v"(w6 + gather_off), "v"((w6 ^ 64u) + gather_off)); \
            }
            asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(pair) : "v"(wp));
            GN_TF_ISSUE(pair & 0xffffu, x0, x1);
#pragma unroll
            for (int s = 0; s < kSlots; ++s) {
                const int c = (int)((counts >> (8 * s)) & 255u);       // wave-uniform
                for (int k = 0; k < c; ++k) {
                    wp += 64;
                    asm volatile("ds_read_b32 %0, %1" : "=v"(next) : "v"(wp));
                    GN_TF_ISSUE(pair >> 16, y0, y1);                   // outstanding: x0 x1 next y0 y1
                    asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(x0), "+v"(x1));
                    acc0[s] += x0;
                    acc1[s] += x1;
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(next));
                    GN_TF_ISSUE(next & 0xffffu, x0, x1);               // outstanding: y0 y1 x0 x1
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(y0), "+v"(y1));
                    acc0[s] += y0;
                    acc1[s] += y1;
                    pair = next;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x0), "+v"(x1));  // the look-ahead reads of the step after the last
#undef GN_TF_ISSUE
        }
#ifdef GN_STAMPS
        const unsigned long long st_d = __builtin_amdgcn_s_memtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the next W fragments have landed in LDS
        __syncthreads();                                               // every wave is done with H_r
#ifdef GN_STAMPS
        st_mfma += st_b - st_a; st_wait += (st_c - st_b) + (__builtin_amdgcn_s_memtime() - st_d); st_gather += st_d - st_c;
#endif
    }
#undef GN_TF_FETCH_W

    // ---- one slab per workgroup: [row][groups][32] ----
#ifdef GN_STAMPS
    const unsigned long long st_t2 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int s = 0; s < kSlots; ++s) {
        const int row = cell_row[s * kQuads + wave * 16 + quad];
        if (row >= 0) {
            f32x4* o = reinterpret_cast<f32x4*>(slabs + ((size_t)row * a.groups + wg) * kFout) + qj;
            o[0] = acc0[s];
            o[4] = acc1[s];
        }
    }
#ifdef GN_STAMPS
    const int wv = wg * kWaves + wave;
    if (lane == 0 && wv < 4096) {
        unsigned long long* o = g_tf_stamps[wv];
        o[0] = st_t0; o[1] = st_t2; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = st_mfma; o[4] = st_gather; o[5] = st_wait;
        o[6] = (unsigned long long)(p_end - wg_parts[wg]); o[7] = st_t1; o[8] = 0; o[9] = 0; o[10] = 0; o[11] = 0;
    }
#endif
}

// Opt-in (GN_RGCN_TF=1): measured at the speed of k_rgcn_acc on pose0/1/2-syn with twice its HBM traffic (256 slabs).
bool tf_disabled() {
    if (gn::fast_paths_disabled()) return true;
    const char* e = getenv("GN_RGCN_TF");
    return !(e && e[0] == '1');
}

size_t tf_lds_bytes(int tiles) { return (size_t)(tiles + 1) * 2048 + (size_t)kWaves * kWinBytes + 8 * 1024; }

template <int FIN>
gn_status launch_tf(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, const f32x4* wfrag, float* slabs, hipStream_t st) {
    const size_t lds_bytes = tf_lds_bytes(plan->tf_tiles);
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_rgcn_tf<FIN>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    TfDims dm;
    dm.ld_x = ld_x; dm.n = (int)plan->num_nodes; dm.tiles = plan->tf_tiles; dm.groups = plan->tf_g;
    k_rgcn_tf<FIN><<<plan->tf_g, kThreads, lds_bytes, st>>>(
        x, reinterpret_cast<const u32x4*>(wfrag), reinterpret_cast<const u32x4*>(plan->tf_stream.p), plan->tf_wg_parts.p,
        plan->tf_part_rel.p, reinterpret_cast<const uint2*>(plan->tf_part_wave.p), plan->tf_cell_row.p, slabs, dm);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

// Builds the workgroup lists and the per-wave word streams of the shard from its edges in (relation, destination)
// order: rp[r * N + i] .. rp[r * N + i + 1] are the positions in srcs of the sources of (relation r, destination i).
// Leaves plan->tf_ok = 0 when the graph does not qualify.
gn_status gn_rgcn_build_tf_plan(gn_rgcn_plan* plan, const std::vector<int32_t>& rp, const std::vector<uint32_t>& srcs,
                                hipStream_t st) {
    plan->tf_ok = 0;
    const int64_t N = plan->num_nodes, R = plan->num_relations, E = plan->shard_edges;
    if (tf_disabled() || N < 1 || N > kMaxNodes || R < 1 || E < 1) return GN_OK;
    const int tiles = (int)gn::ceil_div(N, 16);
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }

    // ---- destination rows -> (wave, slot, quad) cells, by total in-degree: the 16 rows of a cell run in lock step ----
    std::vector<int64_t> tot(N, 0);
    for (int64_t r = 0; r < R; ++r)
        for (int64_t i = 0; i < N; ++i) tot[i] += rp[r * N + i + 1] - rp[r * N + i];
    std::vector<int> order(N);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return tot[x] > tot[y]; });
    std::vector<int32_t> cell_row((size_t)kSlots * kQuads, -1);
    for (int j = 0; j < tiles; ++j) {
        const int s = j / kWaves, k = j % kWaves, w = (s & 1) ? kWaves - 1 - k : k;         // snake over the waves
        for (int q = 0; q < 16 && j * 16 + q < N; ++q) cell_row[(size_t)s * kQuads + w * 16 + q] = order[j * 16 + q];
    }
    // zero rows for padded slots, one per colour (the tile behind the table)
    uint16_t zero_word[4] = {0, 0, 0, 0};
    for (uint32_t k = 0; k < 16; ++k) zero_word[tf_colour((uint32_t)tiles * 16 + k)] = (uint16_t)tf_word((uint32_t)tiles * 16 + k);

    // ---- slices: (relation, chunk c) takes edges [c kChunk, (c + 1) kChunk) of every (relation, destination) run ----
    struct Part { int32_t rel, chunk; uint8_t steps[kWaves][kSlots]; int64_t cost; };
    std::vector<Part> parts;
    for (int64_t r = 0; r < R; ++r) {
        int32_t longest = 0;
        for (int64_t i = 0; i < N; ++i) longest = std::max(longest, rp[r * N + i + 1] - rp[r * N + i]);
        for (int c = 0; c * kChunk < longest; ++c) {
            Part pt;
            pt.rel = (int32_t)r; pt.chunk = c;
            int64_t sum = 0, longest_wave = 0;
            for (int w = 0; w < kWaves; ++w) {
                int64_t mine = 0;
                for (int s = 0; s < kSlots; ++s) {
                    int32_t it = 0;
                    for (int q = 0; q < 16; ++q) {
                        const int row = cell_row[(size_t)s * kQuads + w * 16 + q];
                        if (row < 0) continue;
                        const int32_t cnt = rp[r * N + row + 1] - rp[r * N + row];
                        it = std::max(it, std::min(kChunk, std::max(0, cnt - c * kChunk)));
                    }
                    pt.steps[w][s] = (uint8_t)((it + 1) / 2);
                    mine += 2 * pt.steps[w][s];
                }
                sum += mine;
                longest_wave = std::max(longest_wave, mine);
            }
            pt.cost = GN_TF_PART_COST + std::max<int64_t>(sum * GN_TF_ITER_COST, longest_wave * GN_TF_WAVE_ITER_COST);
            parts.push_back(pt);
        }
    }
    if (parts.empty()) return GN_OK;
    const int G = (int)std::min<size_t>((size_t)cus, parts.size());
    // longest-processing-time assignment of the slices to the workgroups
    std::vector<std::vector<int>> per_wg(G);
    {
        std::vector<int> by_cost(parts.size());
        std::iota(by_cost.begin(), by_cost.end(), 0);
        std::stable_sort(by_cost.begin(), by_cost.end(), [&](int x, int y) { return parts[x].cost > parts[y].cost; });
        std::vector<std::pair<int64_t, int>> heap;
        for (int w = 0; w < G; ++w) heap.emplace_back(0, w);
        auto cmp = [](const std::pair<int64_t, int>& x, const std::pair<int64_t, int>& y) { return x > y; };
        std::make_heap(heap.begin(), heap.end(), cmp);
        for (int idx : by_cost) {
            std::pop_heap(heap.begin(), heap.end(), cmp);
            auto& top = heap.back();
            per_wg[top.second].push_back(idx);
            top.first += parts[idx].cost;
            std::push_heap(heap.begin(), heap.end(), cmp);
        }
    }
    // ---- emit: workgroup lists in relation order, the per-wave counts and the word streams ----
    std::vector<int32_t> wg_parts(G + 1, 0), part_rel;
    std::vector<uint32_t> part_wave;                                  // [part][wave] {stream offset in 16-byte units, steps per slot}
    std::vector<uint16_t> words;
    part_rel.reserve(parts.size());
    part_wave.reserve(parts.size() * kWaves * 2);
    words.reserve((size_t)E * 2 + 4096);
    std::vector<uint16_t> rowbuf[16];
    int cur[16][4], end[16][4], rem[16];
    // quads (lanes / 4) of the four access groups ds_read_b128 serves a wave in (the same table as rgcn_acc.hip)
    static const int kGroupQuads[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    for (int g = 0; g < G; ++g) {
        std::vector<int>& mine = per_wg[g];
        std::sort(mine.begin(), mine.end(), [&](int x, int y) {
            return parts[x].rel != parts[y].rel ? parts[x].rel < parts[y].rel : parts[x].chunk < parts[y].chunk;
        });
        for (int idx : mine) {
            const Part& pt = parts[idx];
            const int64_t r = pt.rel;
            part_rel.push_back(pt.rel);
            for (int w = 0; w < kWaves; ++w) {
                const size_t base = words.size();                      // multiple of 32 words (64 bytes)
                part_wave.push_back((uint32_t)(base / 8));
                uint32_t packed_steps = 0;
                for (int s = 0; s < kSlots; ++s) packed_steps |= (uint32_t)pt.steps[w][s] << (8 * s);
                part_wave.push_back(packed_steps);
                size_t step0 = 0;
                for (int s = 0; s < kSlots; ++s) {
                    const int L = 2 * pt.steps[w][s];
                    if (L == 0) continue;
                    words.resize(base + (step0 + pt.steps[w][s]) * 32);
                    // the quads' sources of this slice, as words grouped by colour
                    for (int q = 0; q < 16; ++q) {
                        rowbuf[q].clear();
                        rem[q] = 0;
                        for (int c = 0; c < 4; ++c) cur[q][c] = end[q][c] = 0;
                        const int row = cell_row[(size_t)s * kQuads + w * 16 + q];
                        if (row < 0) continue;
                        const int32_t b0 = rp[r * N + row] + pt.chunk * kChunk;
                        const int32_t b1 = std::min<int32_t>(rp[r * N + row + 1], b0 + kChunk);
                        if (b1 <= b0) continue;
                        rem[q] = b1 - b0;
                        int cnt[4] = {0, 0, 0, 0};
                        for (int32_t e = b0; e < b1; ++e) ++cnt[tf_colour(srcs[e])];
                        int off = 0;
                        for (int c = 0; c < 4; ++c) { cur[q][c] = off; off += cnt[c]; end[q][c] = off; }
                        rowbuf[q].resize(rem[q]);
                        int fillp[4] = {cur[q][0], cur[q][1], cur[q][2], cur[q][3]};
                        for (int32_t e = b0; e < b1; ++e) rowbuf[q][fillp[tf_colour(srcs[e])]++] = (uint16_t)tf_word(srcs[e]);
                    }
                    for (int it = 0; it < L; ++it) {
                        const int left = L - it;                       // iterations left, this one included
                        uint16_t* out = words.data() + base + (step0 + it / 2) * 32 + (it & 1);     // + quad * 2
                        for (int ag = 0; ag < 4; ++ag) {               // the quads that share a ds_read_b128 access group
                            int ord[4] = {kGroupQuads[ag][0], kGroupQuads[ag][1], kGroupQuads[ag][2], kGroupQuads[ag][3]};
                            // quads that cannot sit out first, then the fuller ones
                            std::sort(ord, ord + 4, [&](int x, int y) {
                                const bool mx = rem[x] >= left, my = rem[y] >= left;
                                return mx != my ? mx : rem[x] > rem[y];
                            });
                            unsigned used = 0;
                            bool idle[4] = {false, false, false, false};
                            for (int k = 0; k < 4; ++k) {
                                const int q = ord[k];
                                if (rem[q] == 0) { idle[k] = true; continue; }
                                int best = -1, bestcnt = 0;
                                for (int c = 0; c < 4; ++c) {
                                    const int n_c = end[q][c] - cur[q][c];
                                    if (n_c > bestcnt && !((used >> c) & 1)) { best = c; bestcnt = n_c; }
                                }
                                if (best < 0) {
                                    if (rem[q] < left) { idle[k] = true; continue; }          // can wait for a free colour
                                    for (int c = 0; c < 4; ++c) {
                                        const int n_c = end[q][c] - cur[q][c];
                                        if (n_c > bestcnt) { best = c; bestcnt = n_c; }
                                    }
                                }
                                out[q * 2] = rowbuf[q][cur[q][best]++];
                                --rem[q];
                                used |= 1u << best;
                            }
                            for (int k = 0; k < 4; ++k) {              // quads that sit out: a zero row in a colour nobody reads
                                if (!idle[k]) continue;
                                int c = 0;
                                while (c < 3 && ((used >> c) & 1)) ++c;
                                out[ord[k] * 2] = zero_word[c];
                                used |= 1u << c;
                            }
                        }
                    }
                    step0 += pt.steps[w][s];
                }
            }
        }
        wg_parts[g + 1] = (int32_t)part_rel.size();
    }
    words.resize(words.size() + 2 * kWinBytes / 2 + 1024, zero_word[0]);     // the look-ahead loads run past the last slice
    if (words.size() / 8 >= ((size_t)1 << 31)) return GN_OK;

    const size_t stream_words = (words.size() + 1) / 2;
    GN_HIP(plan->tf_stream.alloc(stream_words));
    GN_HIP(plan->tf_wg_parts.alloc(wg_parts.size()));
    GN_HIP(plan->tf_part_rel.alloc(part_rel.size()));
    GN_HIP(plan->tf_part_wave.alloc(part_wave.size()));
    GN_HIP(plan->tf_cell_row.alloc(cell_row.size()));
    words.resize(stream_words * 2, zero_word[0]);
    GN_HIP(hipMemcpyAsync(plan->tf_stream.p, words.data(), stream_words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->tf_wg_parts.p, wg_parts.data(), wg_parts.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->tf_part_rel.p, part_rel.data(), part_rel.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->tf_part_wave.p, part_wave.data(), part_wave.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->tf_cell_row.p, cell_row.data(), cell_row.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));       // host vectors go out of scope after this
    plan->tf_tiles = tiles; plan->tf_g = G; plan->tf_parts = (int64_t)part_rel.size();
    int64_t iters = 0;
    for (const Part& pt : parts)
        for (int w = 0; w < kWaves; ++w)
            for (int s = 0; s < kSlots; ++s) iters += 2 * pt.steps[w][s];
    plan->tf_iters = iters;
    plan->tf_ok = 1;
    return GN_OK;
}

// The kernel takes over from k_rgcn_acc when its plan exists, the transform is the split-bf16 one (GN_ACC_EXACT=1 keeps
// the fp32 kernel) and the width is one of its instantiations.
bool gn_rgcn_tf_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, bool split) {
    if (!plan->tf_ok || tf_disabled() || !split) return false;
    if (fout != kFout || !(fin == 16 || fin == 32 || fin == 48)) return false;
    return tf_lds_bytes(plan->tf_tiles) <= 159 * 1024;
}

gn_status gn_rgcn_tf_launch(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const void* wfrag,
                            float* slabs, hipStream_t st) {
    const f32x4* w = static_cast<const f32x4*>(wfrag);
    switch ((int)fin) {
        case 16: return launch_tf<16>(plan, x, ld_x, w, slabs, st);
        case 32: return launch_tf<32>(plan, x, ld_x, w, slabs, st);
        default: return launch_tf<48>(plan, x, ld_x, w, slabs, st);
    }
}

#ifdef GN_STAMPS
extern "C" __attribute__((visibility("default"))) int gn_debug_read_tf_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_tf_stamps), sizeof(unsigned long long) * 4096 * 12);
}
#endif
