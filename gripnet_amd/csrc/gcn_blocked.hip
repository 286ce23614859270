// GCN-style layer with the neighbour features staged through LDS (column-blocked gather).
//
//   out[i, :] = act( sum_{e: dst(e)=i} norm_e (x W)[src(e), :] + b ),  norm_e = dis[src] w_e dis[dst]
//                                                                     (gripnet/layers.py:52-100)
//
// The wave-per-destination-row kernels (aggregate.cuh) ask the L2 for one neighbour row per edge and sit at the rate
// of the per-CU miss path (~64 lines in flight per CU: 0.15 random rows per clock whatever the row width).  Here the
// gathered table never leaves a CU once it is there.  For graphs whose stored weights are all 1 (GripNet passes
// edge_weight = ones or None, GripNet-pose.py:52,117-120) the coefficient factorises, norm_e = dis[src] dis[dst]:
//
//   out[i, :] = act( dis[i] * sum_{e: dst=i} T[src(e), :] + b ),      T[s, :] = dis[s] * (x W)[s, :]
//
// and an edge is a 16-bit source id, not a column index and a coefficient.  Two launches per layer:
//   k_col_transform  T = dis * (x W), exact fp32, written column-group-major: [out / CW][rows][CW], CW = 2 (or 1)
//                    columns per group, so that the slice of ALL nodes for one column group is contiguous and fits
//                    the 160 KB of LDS of one CU (19,081 genes x 8 bytes = 153 KB);
//   k_col_gather     one workgroup per (column group, range of destination rows): its slice of T goes straight into
//                    LDS (LDS-DMA, 1 KB per wave instruction), then a quad of lanes per destination row adds the
//                    CW-float entries its ids name - one ds_read_b64 per lane and edge - and writes its CW columns
//                    of the layer's slot of the concat buffer, scaled, biased, activated.  No partial sums cross
//                    workgroups: nothing to combine, nothing written but the result.
// The plan orders every row's edges so that the 32 lanes of a ds_read_b64 access group read 32 different bank pairs
// wherever the graph allows it (the order of addends is free), sorts the destination rows of a range by degree so
// that the 16 rows of a tile finish together, and deals the rows to the ranges so that every range holds the same
// number of edges.  Fixed summation order, no atomics: bitwise reproducible.
// Eight column groups x 32 ranges = 256 workgroups for 16 output columns.  Applies while one column group of all nodes
// fits the LDS: N <= ~19,900 nodes at CW = 2, ~39,800 at CW = 1; larger tables stay on the wave-per-row kernels.
// (Measured on the way, pose0-syn: source blocks of whole 16-float rows with per-block partial sums and a combine
// launch took 3.8 + 11.2 + 5.5 us per layer - the 10 MB of partial sums are flushed at one kernel boundary and read back
// behind the next - against 19.2 / 14.7 us for the wave-per-row kernels.)
#include "aggregate.cuh"

#include <algorithm>
#include <numeric>
#include <vector>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef GN_COL_THREADS
#define GN_COL_THREADS 1024
#endif
constexpr int kColThreads = GN_COL_THREADS;
constexpr int kColWaves = kColThreads / 64;
static_assert(kColWaves == gn_layout::kColLayoutWaves, "the plan cuts a range's tiles into the waves of one workgroup");
constexpr size_t kColLdsBytes = 160 * 1024;
constexpr int kColSlack = gn_layout::kColLayoutSlack;            // spare iterations behind the id stream: look-ahead loads never leave it
constexpr int kWaveInts = 12;            // per wave of a range: first tile, end tile, first iteration, end iteration, the ends of its first five tiles, 3 unused

bool blocked_disabled() {
    if (gn::fast_paths_disabled()) return true;
    const char* e = getenv("GN_DISABLE_BLOCKED");
    return e && e[0] == '1';
}

// ---- T = dis * (x W), column-group-major ---------------------------------------------------------------------------
// A quad of lanes per row: lane j holds the features 16 p + 4 j + c of its row (one contiguous 64-byte read per quad and
// 16 features), hands them round the quad with DPP and accumulates the output columns 16 nt + 4 j + c against W rows
// read from LDS (w == null: T = dis * x, FIN == 16 NT).
struct TransArgs {
    const float* x; int64_t ld_x;
    const float* w;                      // [FIN, 16 NT] or null
    const float* dis;                    // [rows_pad], zero beyond the last node
    float* table;                        // [16 NT / CW][rows_pad][CW]
    int n, rows_pad;
};

template <int CTRL>
__device__ __forceinline__ float quad_bcast(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int FIN, int NT, int CW>
__global__ __launch_bounds__(256) void k_col_transform(TransArgs a) {
    constexpr int FOUT = 16 * NT, P = FIN / 16;           // 16-byte pieces of a row per lane
    __shared__ f32x4 wl[FIN * FOUT / 4];
    const int tid = threadIdx.x, lane = tid & 63, q = lane >> 2, j = lane & 3;
    const int tiles = a.rows_pad / 16;
    int t = blockIdx.x * 4 + (tid >> 6);
    f32x4 xv[P];
    float d = 0.f;
    auto load_row = [&](int tt) {                          // requested before W is staged: one round trip, not two
        const int row = 16 * min(tt, tiles - 1) + q, rowc = min(row, a.n - 1);
        const float* __restrict__ px = a.x + (int64_t)rowc * a.ld_x + 4 * j;
#pragma unroll
        for (int p = 0; p < P; ++p) xv[p] = *reinterpret_cast<const f32x4*>(px + 16 * p);
        d = a.dis[row];
    };
    load_row(t);
    if (a.w) {
        for (int i = tid; i < FIN * FOUT / 4; i += 256) wl[i] = reinterpret_cast<const f32x4*>(a.w)[i];
        __syncthreads();
    }
    for (; t < tiles; t += gridDim.x * 4) {
        const int row = 16 * t + q;
        f32x4 acc[NT];
        if (a.w == nullptr) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = xv[nt < P ? nt : 0];
        } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int p = 0; p < P; ++p)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = xv[p][c];
                    const float v0 = quad_bcast<0x00>(v), v1 = quad_bcast<0x55>(v), v2 = quad_bcast<0xAA>(v), v3 = quad_bcast<0xFF>(v);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {       // lane jj of the quad holds feature 16 p + 4 jj + c
                        acc[nt] += v0 * wl[(16 * p + 0 + c) * (FOUT / 4) + 4 * nt + j];
                        acc[nt] += v1 * wl[(16 * p + 4 + c) * (FOUT / 4) + 4 * nt + j];
                        acc[nt] += v2 * wl[(16 * p + 8 + c) * (FOUT / 4) + 4 * nt + j];
                        acc[nt] += v3 * wl[(16 * p + 12 + c) * (FOUT / 4) + 4 * nt + j];
                    }
                }
        }
        const float scale = d;
        const bool real = row < a.n;                        // rows beyond the last node (the zero row among them) are exact zeros
        if (t + (int)gridDim.x * 4 < tiles) load_row(t + gridDim.x * 4);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const f32x4 v = real ? acc[nt] * scale : (f32x4){0.f, 0.f, 0.f, 0.f};     // columns 16 nt + 4 j .. + 3 = groups (16 nt + 4 j) / CW ..
            if constexpr (CW == 2) {
                float* o = a.table + ((size_t)(8 * nt + 2 * j) * a.rows_pad + row) * 2;
                *reinterpret_cast<f32x2*>(o) = (f32x2){v[0], v[1]};
                *reinterpret_cast<f32x2*>(o + (size_t)a.rows_pad * 2) = (f32x2){v[2], v[3]};
            } else {
                float* o = a.table + (size_t)(16 * nt + 4 * j) * a.rows_pad + row;
#pragma unroll
                for (int c = 0; c < 4; ++c) o[(size_t)c * a.rows_pad] = v[c];
            }
        }
    }
}

// ---- the gather ------------------------------------------------------------------------------------------------------
struct ColArgs {
    const float* table;                  // [groups][rows_pad][CW]
    const int32_t* cell;                 // [ranges][waves][kWaveInts]
    const int32_t* tile_off;             // [tiles + 3] first iteration of every tile
    const int32_t* tile_rows;            // [tiles + 4][16] destination row of every quad (-1: none)
    const float* tile_dis;               // [tiles + 4][16] dis of that row
    const u32x2* ids;                    // per iteration 64 lanes x 4 uint16 source ids (512 bytes)
    const float* dis; const float* bias;
    float* out; int64_t ld_out;
    int n, rows_pad, groups, ranges, relu;
    gn_side_copy side;
};

#ifdef GN_STAMPS
// Diagnostic build only (make STAMPS=1): per-workgroup phase times, never part of the product library.
__device__ unsigned long long g_blk_stamps[2][512][4];
#endif

template <int CW>
__global__ __launch_bounds__(kColThreads) void k_col_gather(ColArgs a, int stamp_set) {
    typedef float vec_t __attribute__((ext_vector_type(CW)));
    extern __shared__ f32x4 lds4[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 2, j = lane & 3;
    // Workgroups go to the eight XCDs round-robin and every XCD has its own L2: the column groups of one range sit on
    // one XCD (its id stream is fetched into one L2), and each XCD sees every column group.
    const int wg = blockIdx.x, per_x = (a.ranges + 7) / 8;
    const int x = wg & 7, k = wg >> 3;
    const int range = x * per_x + k / a.groups, cg = k % a.groups;
    const bool live = range < a.ranges;
#ifdef GN_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif
    // ---- this wave's tiles [c0, c1): read with scalar loads BEFORE the first vector-memory instruction (behind the
    //      DMA loads the compiler would make them vector loads, and those return in issue order: behind the table) ----
    const int32_t* __restrict__ cell = a.cell + ((size_t)(live ? range : 0) * kColWaves + wave) * kWaveInts;
    const int c0 = cell[0], c1 = cell[1];
    int it = cell[2];
    const int it_last = cell[3];
    const int e0 = cell[4], e1 = cell[5], e2 = cell[6], e3 = cell[7], e4 = cell[8];   // last iteration of the wave's first five tiles
    // ---- this column group of every node -> LDS, 1 KB per wave instruction (rows_pad * CW * 4 is a multiple of 1 KB;
    //      row n of the table is zero: padded id slots name it) ----
    if (live) {
        const int total = a.rows_pad * CW / 4;                            // float4 pieces
        const f32x4* __restrict__ src = reinterpret_cast<const f32x4*>(a.table + (size_t)cg * a.rows_pad * CW);
        for (int i = wave * 64; i < total; i += kColThreads)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i + lane),
                                             (__attribute__((address_space(3))) void*)(lds4 + i), 16, 0, 0);
    }
    do {
    if (!live) break;
    asm volatile("" ::: "memory");                                         // what follows is requested behind the DMA loads
    int tile_end = e0;
    // sixteen iterations of ids are requested before anything is consumed (a wave has ~13 on PoSE): the stream comes
    // from HBM once per range, and a loop that asks for it three iterations ahead pays that latency again and again
    const u32x2* __restrict__ sp = a.ids + (size_t)it * 64 + lane;
    u32x2 wa[8], wb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) wa[k] = sp[64 * k];
#pragma unroll
    for (int k = 0; k < 8; ++k) wb[k] = sp[512 + 64 * k];
    sp += 1024;
    // destination row and scale of this quad in the wave's first four tiles (more tiles: read when they come up)
    const int32_t* __restrict__ trp = a.tile_rows + (size_t)c0 * 16 + q;   // the arrays end with four spare tiles
    const float* __restrict__ tdp = a.tile_dis + (size_t)c0 * 16 + q;
    int row = trp[0];
    const int row1 = trp[16], row2 = trp[32], row3 = trp[48];
    float d = tdp[0];
    const float d1 = tdp[16], d2 = tdp[32], d3 = tdp[48];
    int t = c0;
    vec_t bias = (vec_t)(0.f);
    if (a.bias) bias = *reinterpret_cast<const vec_t*>(a.bias + CW * cg);
    // Everything requested so far has to be here: the LDS-DMA loads, and with them the id / row / scale loads issued
    // behind them.  (Loads return in issue order, so a count that lets exactly those later loads stay in flight would
    // release the barrier a little earlier - measured: the same total, the gather needs the first ids at once - but such
    // a count depends on how many vector loads the compiler emits for the lines above.  Zero does not.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef GN_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memrealtime();
#endif
    const vec_t* __restrict__ tab = reinterpret_cast<const vec_t*>(lds4);
    vec_t acc = (vec_t)(0.f);
    // (macros, not lambdas: with the closures hipcc 7.2 kept the kernel arguments and every captured local in scratch)
    // GN_COL_FLUSH: tile t is complete - fold the quad, scale, bias, ReLU, write; step to the next tile
#define GN_COL_FLUSH()                                                                                         \
    do {                                                                                                       \
        _Pragma("unroll") for (int c = 0; c < CW; ++c) {                                                       \
            float v = acc[c];                                                                                  \
            v += __shfl_xor(v, 1);                                                                             \
            v += __shfl_xor(v, 2);                                                                             \
            acc[c] = v;                                                                                        \
        }                                                                                                      \
        if (j == 0 && row >= 0) {                                                                              \
            vec_t r = acc * d + bias;                                                                          \
            if (a.relu) {                                                                                      \
                _Pragma("unroll") for (int c = 0; c < CW; ++c) r[c] = fmaxf(r[c], 0.f);                        \
            }                                                                                                  \
            *reinterpret_cast<vec_t*>(a.out + (int64_t)row * a.ld_out + CW * cg) = r;                          \
        }                                                                                                      \
        acc = (vec_t)(0.f);                                                                                    \
        ++t;                                                                                                   \
        const int kk = t - c0;                                                                                 \
        tile_end = kk == 1 ? e1 : (kk == 2 ? e2 : (kk == 3 ? e3 : (kk == 4 ? e4 : a.tile_off[t + 1])));       \
        if (kk < 4) {                                                                                          \
            row = kk == 1 ? row1 : (kk == 2 ? row2 : row3);                                                    \
            d = kk == 1 ? d1 : (kk == 2 ? d2 : d3);                                                            \
        } else {                                                                                               \
            row = a.tile_rows[(size_t)t * 16 + q];                                                             \
            d = a.tile_dis[(size_t)t * 16 + q];                                                                \
        }                                                                                                      \
    } while (0)
    // GN_COL_CONSUME: eight iterations out of registers; tile ends are wave-uniform (the inner while also passes empty tiles)
#define GN_COL_CONSUME(w)                                                                                      \
    _Pragma("unroll") for (int k = 0; k < 8; ++k) {                                                            \
        while (t < c1 && it == tile_end) GN_COL_FLUSH();                                                       \
        if (it < it_last) {                                                                                    \
            const uint32_t s0 = w[k].x & 0xffffu, s1 = w[k].x >> 16, s2 = w[k].y & 0xffffu, s3 = w[k].y >> 16; \
            acc += (tab[s0] + tab[s1]) + (tab[s2] + tab[s3]);                                                  \
            ++it;                                                                                              \
        }                                                                                                      \
    }
    while (it < it_last) {
        GN_COL_CONSUME(wa)
        if (it >= it_last) break;
#pragma unroll
        for (int k = 0; k < 8; ++k) wa[k] = sp[64 * k];                    // the stream has slack behind its end
        GN_COL_CONSUME(wb)
#pragma unroll
        for (int k = 0; k < 8; ++k) wb[k] = sp[512 + 64 * k];
        sp += 1024;
    }
    while (t < c1) GN_COL_FLUSH();
#undef GN_COL_CONSUME
#undef GN_COL_FLUSH
#ifdef GN_STAMPS
    if (tid == 0 && blockIdx.x < 512) {
        unsigned long long* o = g_blk_stamps[stamp_set & 1][blockIdx.x];
        o[0] = st0; o[1] = st1; o[2] = __builtin_amdgcn_s_memrealtime(); o[3] = 0;
    }
#endif
    } while (0);
    if (a.side.dst) {                                                      // concat slot: streamed by the whole grid, last
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t tt = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; tt < total; tt += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = tt / a.side.cols, c = tt - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + c];
            a.side.dst[i * a.side.ld_dst + c] = a.side.mode ? fabsf(v) : v;
        }
    }
}

template <int FIN, int NT>
void launch_transform(const TransArgs& t, int cw, hipStream_t st) {
    const int grid = (int)std::min<int64_t>(gn::ceil_div(t.rows_pad / 16, 4), GN_AGG_GRID);
    if (cw == 2) k_col_transform<FIN, NT, 2><<<grid, 256, 0, st>>>(t); else k_col_transform<FIN, NT, 1><<<grid, 256, 0, st>>>(t);
}

}  // namespace

// True when gn_graph_aggregate_f32 takes the LDS-staged path for these shapes.
bool gn_blocked_applicable(const gn_graph_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* w, int64_t fout) {
    if (!plan->blk_ok || blocked_disabled()) return false;
    if (!(fout == 16 || fout == 32) || fout > plan->blk_cols) return false;
    if (w) {
        if (!(fin == 16 || fin == 32 || fin == 64)) return false;
        if (!gn::aligned16(w)) return false;
    } else if (fin != fout) {
        return false;
    }
    return (ld_x % 4) == 0 && gn::aligned16(x);
}

gn_status gn_blocked_aggregate(const gn_graph_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* w,
                               int64_t fout, const float* bias, int relu, float* out, int64_t ld_out,
                               const gn_side_copy& side, hipStream_t st) {
    const int cw = plan->blk_cw;
    GN_REQUIRE((ld_out % cw) == 0 && (reinterpret_cast<uintptr_t>(out) % (4 * cw)) == 0 &&
               (!bias || (reinterpret_cast<uintptr_t>(bias) % (4 * cw)) == 0),
               "the LDS-staged path writes %d-float column groups: out and bias must be aligned to them", cw);
    // 1. the scaled table T = dis * (x W), column-group-major
    TransArgs t;
    t.x = x; t.ld_x = ld_x; t.w = w; t.dis = plan->blk_dis.p; t.table = plan->blk_table.p; t.n = (int)plan->rows;
    t.rows_pad = plan->blk_rows;
    switch ((int)((w ? fin : fout) * 100 + fout)) {
        case 1616: launch_transform<16, 1>(t, cw, st); break;
        case 3216: launch_transform<32, 1>(t, cw, st); break;
        case 6416: launch_transform<64, 1>(t, cw, st); break;
        case 1632: launch_transform<16, 2>(t, cw, st); break;
        case 3232: launch_transform<32, 2>(t, cw, st); break;
        case 6432: launch_transform<64, 2>(t, cw, st); break;
        default: return gn::fail(GN_ERR_UNSUPPORTED, "no LDS-staged kernel for %lld -> %lld features", (long long)fin, (long long)fout);
    }
    GN_LAUNCH_CHECK();
    // 2. gather from LDS, one workgroup per (column group, range of destination rows)
    ColArgs a;
    a.table = plan->blk_table.p; a.cell = plan->blk_cell.p; a.tile_off = plan->blk_tile_off.p; a.tile_rows = plan->blk_tile_rows.p;
    a.tile_dis = plan->blk_tile_dis.p;
    a.ids = reinterpret_cast<const u32x2*>(plan->blk_ids.p); a.dis = plan->blk_dis.p; a.bias = bias;
    a.out = out; a.ld_out = ld_out; a.n = (int)plan->rows; a.rows_pad = plan->blk_rows;
    a.groups = (int)(fout / cw); a.ranges = plan->blk_cells; a.relu = relu; a.side = side;
    const int per_x = (a.ranges + 7) / 8;
    const int grid = 8 * per_x * a.groups;
    const size_t lds = (size_t)plan->blk_rows * cw * sizeof(float);
    if (cw == 2) {
        gn_status s = gn::allow_large_lds(reinterpret_cast<const void*>(k_col_gather<2>), (int)kColLdsBytes);
        if (s != GN_OK) return s;
        k_col_gather<2><<<grid, kColThreads, lds, st>>>(a, (int)(fin == 32 ? 0 : 1));
    } else {
        gn_status s = gn::allow_large_lds(reinterpret_cast<const void*>(k_col_gather<1>), (int)kColLdsBytes);
        if (s != GN_OK) return s;
        k_col_gather<1><<<grid, kColThreads, lds, st>>>(a, (int)(fin == 32 ? 0 : 1));
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// Adds the LDS-staged encoding to a GCN plan whose stored weights are all 1 (self loops included), for layers of up to
// `cols` output features (16 or 32).  Not an error when the graph does not qualify: the plan then keeps using the
// wave-per-row kernels (gn_graph_plan_blocked_cols returns 0).  Copies the CSR to the host and synchronises.
extern "C" gn_status gn_graph_plan_build_blocked(gn_graph_plan* plan, int64_t cols, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(cols == 16 || cols == 32, "LDS-staged plans are built for layers of 16 or 32 output features, got %lld", (long long)cols);
    if (plan->blk_ok && plan->blk_cols >= cols) return GN_OK;
    if (!plan->is_gcn || !plan->unit_weights || blocked_disabled()) return GN_OK;
    const int64_t N = plan->rows, nnz = plan->nnz;
    const char* any = getenv("GN_BLOCKED_ANY");                        // tests: no size thresholds
    if (!(any && any[0] == '1') && (N < 4096 || nnz < 16 * N)) return GN_OK;    // small or very sparse graphs: the wave-per-row kernels do fine
    if (N < 1 || N >= 65535) return GN_OK;
    hipStream_t st = gn::as_stream(stream);
    // rows of the table incl. the zero row, padded so that one column group is a whole number of 1 KB DMA pieces
    int cw = 2;
    int64_t rows_pad = gn::ceil_div((N + 1) * cw * 4, 1024) * 1024 / (cw * 4);
    if (rows_pad * cw * 4 > (int64_t)kColLdsBytes) {
        cw = 1;
        rows_pad = gn::ceil_div((N + 1) * cw * 4, 1024) * 1024 / (cw * 4);
        if (rows_pad * cw * 4 > (int64_t)kColLdsBytes) return GN_OK;
    }
    const int groups = (int)(cols / cw);
    const int R = std::max(1, std::min<int>(256 / groups, (int)gn::ceil_div(N, 64)));     // ranges of destination rows

    GN_LAP(nullptr);
    gn::ArenaHold arena;                                       // (before every host array of this build: host_layout.hpp)
    std::vector<int32_t> rp(N + 1), col(nnz);
    GN_HIP(hipMemcpyAsync(rp.data(), plan->rowptr.p, (N + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipMemcpyAsync(col.data(), plan->col.p, nnz * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    std::vector<float> dis_host((size_t)rows_pad + 16, 0.f);
    GN_HIP(hipMemcpyAsync(dis_host.data(), plan->dis.p, N * sizeof(float), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));

    GN_LAP("blocked: CSR to the host");
    gn_layout::BlockedLayout bl = gn_layout::build_blocked_layout(N, R, rp, col, dis_host);
    GN_LAP("blocked: host schedule");
    if (bl.failed) return gn::fail(GN_ERR_UNSUPPORTED, "internal: an edge was not scheduled");
    if (!bl.ok) return GN_OK;
    std::vector<int32_t>& tile_off = bl.tile_off; std::vector<int32_t>& tile_rows = bl.tile_rows; std::vector<int32_t>& cell = bl.cell;
    std::vector<float>& tile_dis = bl.tile_dis; gn::RawVec<uint16_t>& ids = bl.ids;
    const int64_t iters_total = bl.iters_total;
    plan->blk_ok = 0;
    plan->blk_dis.release(); plan->blk_tile_off.release(); plan->blk_ids.release(); plan->blk_cell.release();
    plan->blk_tile_rows.release(); plan->blk_tile_dis.release(); plan->blk_table.release();
    GN_HIP(plan->blk_dis.alloc(dis_host.size()));
    GN_HIP(plan->blk_tile_off.alloc(tile_off.size()));
    GN_HIP(plan->blk_tile_rows.alloc(tile_rows.size()));
    GN_HIP(plan->blk_tile_dis.alloc(tile_dis.size()));
    GN_HIP(hipMemcpyAsync(plan->blk_tile_dis.p, tile_dis.data(), tile_dis.size() * sizeof(float), hipMemcpyHostToDevice, st));
    GN_HIP(plan->blk_ids.alloc(ids.size() / 2));
    GN_HIP(plan->blk_cell.alloc(cell.size()));
    GN_HIP(plan->blk_table.alloc((size_t)rows_pad * cols));
    GN_HIP(hipMemcpyAsync(plan->blk_dis.p, dis_host.data(), dis_host.size() * sizeof(float), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->blk_tile_off.p, tile_off.data(), tile_off.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->blk_tile_rows.p, tile_rows.data(), tile_rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->blk_ids.p, ids.data(), ids.size() * sizeof(uint16_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->blk_cell.p, cell.data(), cell.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("blocked: upload");
    plan->blk_cols = (int)cols; plan->blk_cw = cw; plan->blk_rows = (int)rows_pad; plan->blk_cells = R;
    plan->blk_iters = iters_total;
    plan->blk_ok = 1;
    return GN_OK;
}

#ifdef GN_STAMPS
extern "C" __attribute__((visibility("default"))) int gn_debug_read_blk_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_blk_stamps), sizeof(unsigned long long) * 2 * 512 * 4);
}
#endif

extern "C" int64_t gn_graph_plan_blocked_cols(const gn_graph_plan* plan) {
    return (plan && plan->blk_ok && !blocked_disabled()) ? plan->blk_cols : 0;
}
