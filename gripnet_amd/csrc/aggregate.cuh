// Destination-major gather-reduce shared by the GCN-style layers and the general RGCN path.
//
//   out[i, :] = act( (sum_{p in row i} coef[p] * T[col[p], :]) / max(1, rowdiv[i]) + addend[i, :] + bias )
//
// One 64-lane wave owns one destination row at a time.  The wave reads 64 (col, coef) pairs
// with one coalesced load each, then walks them S = 64/LPE at a time: every group of LPE
// lanes covers the feature row of one neighbour with 16-byte loads (VEC = 4), so a wave
// keeps S independent row gathers in flight.  The S partial sums are folded with cross-lane
// shuffles, in a fixed order: results are bitwise reproducible run to run.
#pragma once

#include "common.h"

// Groups of S neighbour rows a wave requests before it consumes the first one.
#ifndef GN_AGG_U
#define GN_AGG_U 2
#endif
#ifndef GN_AGG_U_WIDE
#define GN_AGG_U_WIDE 8
#endif
// Upper bound of the launch grid (blocks of four waves); rows beyond it are taken grid-stride.
#ifndef GN_AGG_GRID
#define GN_AGG_GRID (256 * 8)
#endif

// Average row length below which LPE lanes own a row (k_aggregate_short: two gathers in flight per lane; k_aggregate_group:
// GN_AGG_GROUP_U of them) instead of a wave
#ifndef GN_AGG_SHORT_MAX_DEG
#define GN_AGG_SHORT_MAX_DEG 8
#endif
#ifndef GN_AGG_GROUP_MAX_DEG
#define GN_AGG_GROUP_MAX_DEG 48
#endif
#ifndef GN_AGG_GROUP_U
#define GN_AGG_GROUP_U 8
#endif

namespace gn {

struct AggArgs {
    const int32_t* rowptr;
    const uint32_t* col;
    const float* coef;     // nullable (all ones)
    const float* table;
    int64_t ld_table;
    int features;
    const float* rowdiv;   // nullable
    const float* addend;   // nullable
    int64_t ld_addend;
    const float* bias;     // nullable
    int relu;
    float* out;
    int64_t ld_out;
    int rows;
    gn_side_copy side = {nullptr, 0, nullptr, 0, 0, 0, 0};   // optional fused row copy (dst == nullptr: none)
    gn_split_planes split = {nullptr, 0, 0, 0, 0};           // optional bf16 split planes of what the launch writes
    const uint32_t* ell_col = nullptr;   // padded rows of the plan (rows of at most 64 entries), or null
    const float* ell_coef = nullptr;
    int64_t nnz = -1;      // stored coefficients, when the caller knows them (picks the short-row kernel)
    int64_t table_rows = -1;   // rows of the gathered table, when the caller knows them (picks the LDS-table kernel)
};

// One value into the split planes of X (gn_split_planes): its three bf16 terms, cut by truncation exactly as the
// relational kernel cuts them itself (rgcn_pair.hip: split_pair), so a layer gives the same bits either way.
__device__ __forceinline__ void write_split(const gn_split_planes& sp, int64_t row, int col, float v) {
    const int cellb = 4 * ((3 * sp.nt + 1) / 2);
    const int cell = col / sp.nt, j = col - cell * sp.nt;
    unsigned short* p = reinterpret_cast<unsigned short*>(static_cast<unsigned char*>(sp.planes) + (row * 16 + cell) * cellb) + j;
    const uint32_t a = __builtin_bit_cast(uint32_t, v);
    const float r = v - __builtin_bit_cast(float, a & 0xffff0000u);
    const uint32_t b = __builtin_bit_cast(uint32_t, r);
    const float s = r - __builtin_bit_cast(float, b & 0xffff0000u);
    p[0] = (unsigned short)(a >> 16);
    p[sp.nt] = (unsigned short)(b >> 16);
    p[2 * sp.nt] = (unsigned short)(__builtin_bit_cast(uint32_t, s) >> 16);
}

template <int UNUSED = 0>   // (a template so that the header can define it in every translation unit)
__global__ __launch_bounds__(256) void k_split_planes(const float* __restrict__ src, int64_t ld_src, int64_t rows, int cols, int col0,
                                                      gn_split_planes sp) {
    const int64_t total = rows * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / cols;
        const int c = (int)(t - i * cols);
        write_split(sp, i, col0 + c, src[i * ld_src + c]);
    }
}

template <int VEC, int LPE>
__global__ __launch_bounds__(256) void k_aggregate(AggArgs a) {
    constexpr int S = kWave / LPE;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPE;
    const int j = lane % LPE;
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);

    if (a.side.dst) {                                          // concat slot: streamed up front by the whole grid
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / a.side.cols, c = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + c];
            a.side.dst[i * a.side.ld_dst + c] = a.side.mode ? fabsf(v) : v;
        }
    }
    for (int row = wave; row < a.rows; row += n_waves) {
        const int begin = a.rowptr[row], end = a.rowptr[row + 1];
        for (int cb = 0; cb * LPE * VEC < a.features; ++cb) {
            const int fcol = (cb * LPE + j) * VEC;
            const bool active = fcol < a.features;
            float acc[VEC];
#pragma unroll
            for (int t = 0; t < VEC; ++t) acc[t] = 0.f;

            for (int base = begin; base < end; base += kWave) {
                const int mine = base + lane;
                const uint32_t c = mine < end ? a.col[mine] : 0u;
                const float v = mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f;
                const int cnt = min(kWave, end - base);
                // The neighbour rows of the whole batch are requested before the first one is consumed: a loop
                // that loads and adds one group of S rows per trip pays an L2 round trip per trip.
                constexpr int IT = kWave / S, U = IT < GN_AGG_U ? IT : GN_AGG_U;   // groups of S rows per batch, U in flight
                for (int it0 = 0; it0 * S < cnt; it0 += U) {
                    float t[U][VEC], vv[U];
#pragma unroll
                    for (int it = 0; it < U; ++it) {
                        const int idx = (it0 + it) * S + slot;
                        const uint32_t cc = (uint32_t)__shfl((int)c, idx);
                        vv[it] = __shfl(v, idx);
#pragma unroll
                        for (int k = 0; k < VEC; ++k) t[it][k] = 0.f;
                        if (idx < cnt && active) {
                            const float* src = a.table + (int64_t)cc * a.ld_table + fcol;
                            if constexpr (VEC == 4) {
                                const float4 r = *reinterpret_cast<const float4*>(src);
                                t[it][0] = r.x; t[it][1] = r.y; t[it][2] = r.z; t[it][3] = r.w;
                            } else {
                                t[it][0] = src[0];
                            }
                        }
                    }
#pragma unroll
                    for (int it = 0; it < U; ++it)
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] += vv[it] * t[it][k];
                }
            }
#pragma unroll
            for (int off = LPE; off < kWave; off <<= 1) {
#pragma unroll
                for (int t = 0; t < VEC; ++t) acc[t] += __shfl_xor(acc[t], off);
            }
            if (slot == 0 && active) {
                const float div = a.rowdiv ? fmaxf(a.rowdiv[row], 1.0f) : 1.0f;
#pragma unroll
                for (int t = 0; t < VEC; ++t) {
                    float val = a.rowdiv ? acc[t] / div : acc[t];
                    if (a.addend) val += a.addend[(int64_t)row * a.ld_addend + fcol + t];
                    if (a.bias) val += a.bias[fcol + t];
                    if (a.relu) val = fmaxf(val, 0.f);
                    acc[t] = val;
                }
                float* dst = a.out + (int64_t)row * a.ld_out + fcol;
                if constexpr (VEC == 4) {
                    *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                } else {
                    dst[0] = acc[0];
                }
            }
        }
    }
}

// Rows of a few neighbours each (the (relation, source) rows of the relational layer's weight gradient: 6 x 10^5 rows
// of ~3 edges): a wave per row spends its time on row bookkeeping.  Here LPE lanes own a row - 64 / LPE rows per wave
// side by side, every lane sums its own four columns over the row's neighbours, two loads in flight - and nothing
// is folded across lanes.
template <int LPE>
__global__ __launch_bounds__(256) void k_aggregate_short(AggArgs a) {
    constexpr int S = kWave / LPE;
    const int lane = threadIdx.x & 63, slot = lane / LPE, j = lane % LPE;
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
    const int fcol = 4 * j;
    const bool active = fcol < a.features;
    if (a.side.dst) {
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / a.side.cols, c = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + c];
            a.side.dst[i * a.side.ld_dst + c] = a.side.mode ? fabsf(v) : v;
        }
    }
    for (int row0 = wave * S; row0 < a.rows; row0 += n_waves * S) {
        const int row = row0 + slot;
        const bool live = row < a.rows && active;
        const int begin = live ? a.rowptr[row] : 0, end = live ? a.rowptr[row + 1] : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = begin; __any(p < end); p += 2) {
            const bool h0 = p < end, h1 = p + 1 < end;
            const uint32_t c0 = h0 ? a.col[p] : 0u, c1 = h1 ? a.col[p + 1] : 0u;
            const float v0 = h0 ? (a.coef ? a.coef[p] : 1.0f) : 0.f, v1 = h1 ? (a.coef ? a.coef[p + 1] : 1.0f) : 0.f;
            float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
            if (h0) r0 = *reinterpret_cast<const float4*>(a.table + (int64_t)c0 * a.ld_table + fcol);
            if (h1) r1 = *reinterpret_cast<const float4*>(a.table + (int64_t)c1 * a.ld_table + fcol);
            acc.x += v0 * r0.x; acc.y += v0 * r0.y; acc.z += v0 * r0.z; acc.w += v0 * r0.w;
            acc.x += v1 * r1.x; acc.y += v1 * r1.y; acc.z += v1 * r1.z; acc.w += v1 * r1.w;
        }
        if (live) {
            float o[4] = {acc.x, acc.y, acc.z, acc.w};
            const float div = a.rowdiv ? fmaxf(a.rowdiv[row], 1.0f) : 1.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float val = a.rowdiv ? o[t] / div : o[t];
                if (a.addend) val += a.addend[(int64_t)row * a.ld_addend + fcol + t];
                if (a.bias) val += a.bias[fcol + t];
                if (a.relu) val = fmaxf(val, 0.f);
                o[t] = val;
            }
            *reinterpret_cast<float4*>(a.out + (int64_t)row * a.ld_out + fcol) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// Rows of a dozen to a few dozen neighbours (the homogeneous layers of the node-classification graphs: 5 x 10^4 rows of
// ~11 edges over a table far larger than an L2): a wave per row walks rowptr -> col -> gathered rows -> fold as four
// dependent round trips per row, six rows deep per wave.  Here LPE lanes own a row - 64 / LPE rows per wave side by
// side, one row per group and no grid-stride loop - the group reads its (col, coef) pairs LPE at a time with one
// coalesced load and requests U neighbour rows before it consumes the first: the latency chain of a row is paid once
// per wave, with 64 / LPE x U row gathers in flight.  Every lane sums its own four columns in neighbour order.
template <int LPE, int U>
__global__ __launch_bounds__(256) void k_aggregate_group(AggArgs a) {
    constexpr int S = kWave / LPE;
    const int lane = threadIdx.x & 63, slot = lane / LPE, j = lane % LPE;
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int fcol = 4 * j;
    const bool active = fcol < a.features;
    if (a.side.dst) {
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / a.side.cols, c = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + c];
            a.side.dst[i * a.side.ld_dst + c] = a.side.mode ? fabsf(v) : v;
        }
    }
    const int row = wave * S + slot;
    const bool live = row < a.rows;
    const int begin = live ? a.rowptr[row] : 0, end = live ? a.rowptr[row + 1] : 0;
    const float* __restrict__ tab = a.table + fcol;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int base = begin; __any(base < end); base += LPE) {
        const int mine = base + j;
        const uint32_t c = mine < end ? a.col[mine] : 0u;
        const float v = mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f;
        const int cnt = min(LPE, end - base);                  // of this group (<= 0 once its row is done)
        for (int t0 = 0; __any(t0 < cnt); t0 += U) {
            float4 r[U];
            float vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t cc = (uint32_t)__shfl((int)c, t0 + u, LPE);
                vv[u] = __shfl(v, t0 + u, LPE);
                r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (t0 + u < cnt && active) r[u] = *reinterpret_cast<const float4*>(tab + (int64_t)cc * a.ld_table);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                acc.x += vv[u] * r[u].x; acc.y += vv[u] * r[u].y; acc.z += vv[u] * r[u].z; acc.w += vv[u] * r[u].w;
            }
        }
    }
    if (live && active) {
        float o[4] = {acc.x, acc.y, acc.z, acc.w};
        const float div = a.rowdiv ? fmaxf(a.rowdiv[row], 1.0f) : 1.0f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float val = a.rowdiv ? o[t] / div : o[t];
            if (a.addend) val += a.addend[(int64_t)row * a.ld_addend + fcol + t];
            if (a.bias) val += a.bias[fcol + t];
            if (a.relu) val = fmaxf(val, 0.f);
            o[t] = val;
        }
        *reinterpret_cast<float4*>(a.out + (int64_t)row * a.ld_out + fcol) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// ---- aggregate, then transform: out[i, :] = act( (sum_p coef[p] * X[col[p], :]) @ W + bias ) ----------------
// A_norm (X W) = (A_norm X) W, so the dense contraction of a GCN-style layer (gripnet/layers.py:73) can run
// on the aggregated row instead of on every node beforehand: no X W launch, no [N, out] round trip through
// HBM.  The gather is bound by the number of L2 requests (one per neighbour row), not by their size, so
// gathering the wider input row costs about the same.  FIN = 4 LPE input features (one float4 per lane of
// the neighbour's group), FOUT in {16, 32}.  Epilogue: after the butterfly fold every lane holds the
// aggregated features 4j..4j+3 of its j; lane (c = lane % FOUT, kq = lane / FOUT) multiplies KPL = FIN * FOUT / 64
// of them (fetched with shuffles) by its register-resident slice W[kq*KPL .. , c] and the 64 / FOUT partial
// sums are folded with two more shuffles.
// (the body takes its block index and block count as arguments: gn_graph_aggregate_with_rgcn_weights_f32 runs it on
// the first blocks of a launch whose other blocks do something else)
template <int LPE, int FOUT>
__device__ __forceinline__ void aggregate_transform_body(const AggArgs& a, const float* __restrict__ w, int block, int n_blocks) {
    constexpr int FIN = 4 * LPE, S = kWave / LPE, G = kWave / FOUT, KPL = FIN / G;
    static_assert(KPL % 4 == 0, "K slice per lane must cover whole float4 groups");
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPE, j = lane % LPE;
    const int c = lane % FOUT, kq = lane / FOUT;
    const int wave = (int)((block * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)(((int64_t)n_blocks * blockDim.x) >> 6);
    float wreg[KPL];
#pragma unroll
    for (int i = 0; i < KPL; ++i) wreg[i] = w[(kq * KPL + i) * FOUT + c];
    const float bias = a.bias ? a.bias[c] : 0.f;

    // concat slot, by the whole grid: the first element of every thread is REQUESTED here and stored behind the rows (a
    // load -> store in front of them put its round trip in front of the rows' own three), the rest (slots longer than
    // the grid) is streamed at the end
    const int64_t side_total = a.side.dst ? a.side.rows * a.side.cols : 0;
    const int64_t side_t0 = block * (int64_t)blockDim.x + threadIdx.x;
    float side_first = 0.f;
    if (side_t0 < side_total) {
        const int64_t i = side_t0 / a.side.cols, cc = side_t0 - i * a.side.cols;
        side_first = a.side.src[i * a.side.ld_src + cc];
    }
    for (int row = wave; row < a.rows; row += n_waves) {
        // padded rows: this lane's (column, coefficient) pair sits at row * 64 + lane - no row pointers in front of it
        int begin, end;
        uint32_t cl0 = 0u;
        float v0 = 0.f;
        if (a.ell_col) {
            cl0 = a.ell_col[(size_t)row * 64 + lane];
            v0 = a.ell_coef[(size_t)row * 64 + lane];
            begin = 0;
            end = __popcll(__ballot(cl0 != 0xffffffffu));
            if (lane >= end) cl0 = 0u;
        } else {
            begin = a.rowptr[row];
            end = a.rowptr[row + 1];
        }
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int base = begin; base < end; base += kWave) {
            const int mine = base + lane;
            const uint32_t cl = a.ell_col ? cl0 : (mine < end ? a.col[mine] : 0u);
            const float v = a.ell_col ? v0 : (mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f);
            const int cnt = min(kWave, end - base);
            // U groups of S neighbour rows are requested before the first one is consumed (see k_aggregate); wide rows
            // (16 lanes each: the external layer, a few hundred destination rows of ~30 edges, latency-bound) ask for
            // eight groups = 32 rows at once
            constexpr int IT = kWave / S, UW = LPE >= 16 ? GN_AGG_U_WIDE : GN_AGG_U, U = IT < UW ? IT : UW;
            for (int it0 = 0; it0 * S < cnt; it0 += U) {
                float4 t[U];
                float vv[U];
#pragma unroll
                for (int it = 0; it < U; ++it) {
                    const int idx = (it0 + it) * S + slot;
                    const uint32_t cc = (uint32_t)__shfl((int)cl, idx);
                    vv[it] = __shfl(v, idx);
                    t[it] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < cnt) t[it] = *reinterpret_cast<const float4*>(a.table + (int64_t)cc * a.ld_table + 4 * j);
                }
#pragma unroll
                for (int it = 0; it < U; ++it) {
                    acc[0] += vv[it] * t[it].x; acc[1] += vv[it] * t[it].y; acc[2] += vv[it] * t[it].z; acc[3] += vv[it] * t[it].w;
                }
            }
        }
#pragma unroll
        for (int off = LPE; off < kWave; off <<= 1) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] += __shfl_xor(acc[t], off);
        }
        // every lane now holds aggregated features 4j .. 4j+3; contract with W
        float part = 0.f;
#pragma unroll
        for (int i = 0; i < KPL; ++i) part += __shfl(acc[i % 4], kq * (KPL / 4) + i / 4) * wreg[i];
#pragma unroll
        for (int off = FOUT; off < kWave; off <<= 1) part += __shfl_xor(part, off);
        if (kq == 0) {
            float val = part + bias;
            if (a.relu) val = fmaxf(val, 0.f);
            a.out[(int64_t)row * a.ld_out + c] = val;
            if (a.split.planes) write_split(a.split, row, a.split.col_main + c, val);
        }
    }
    for (int64_t t = side_t0; t < side_total; t += (int64_t)n_blocks * blockDim.x) {
        const int64_t i = t / a.side.cols, cc = t - i * a.side.cols;
        const float v = t == side_t0 ? side_first : a.side.src[i * a.side.ld_src + cc];
        const float o = a.side.mode ? fabsf(v) : v;
        a.side.dst[i * a.side.ld_dst + cc] = o;
        if (a.split.planes) write_split(a.split, i, a.split.col_side + (int)cc, o);
    }
}

template <int LPE, int FOUT>
__global__ __launch_bounds__(256) void k_aggregate_transform(AggArgs a, const float* __restrict__ w) {
    aggregate_transform_body<LPE, FOUT>(a, w, (int)blockIdx.x, (int)gridDim.x);
}


// ---- aggregate, then transform, for narrow rows (16 or 32 input features): quads instead of shuffles ----------
// Same job and same wave-per-destination-row mapping as k_aggregate_transform, but a neighbour row is gathered
// by a QUAD (lane j: features 4 j .. 4 j + 3, and 16 + 4 j .. for 32 features), so that the 64 (neighbour,
// coefficient) pairs of a batch are handed out with DPP quad broadcasts (the four pairs a quad works through sit in
// its own four lanes: no LDS-pipe shuffle per group of neighbours) and 16 neighbours are gathered per step.  All
// four steps' gathers are in flight before the first is consumed.  The 16 partial sums of the quads are folded with
// a butterfly (fixed order), after which every quad holds the aggregated row; quad q then computes output column q:
// lane (q, j) multiplies the features it holds by its slice of W and the quad folds with two DPP adds.
template <int CTRL>
__device__ __forceinline__ int agg_dpp(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float agg_dpp_add(float x) {
    return x + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, false));
}

template <int VPL>
__global__ __launch_bounds__(256) void k_aggregate_transform_q(AggArgs a, const float* __restrict__ w) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int FOUT = 16;
    const int lane = threadIdx.x & 63;
    const int q = lane >> 2, j = lane & 3;
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int n_waves = (int)(((int64_t)gridDim.x * blockDim.x) >> 6);
    f32x4 wreg[VPL];                                          // W[16 v + 4 j + c][q]
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int c = 0; c < 4; ++c) wreg[v][c] = w[(16 * v + 4 * j + c) * FOUT + q];
    const float bias = a.bias ? a.bias[q] : 0.f;

    if (a.side.dst) {                                          // concat slot: streamed up front by the whole grid
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / a.side.cols, cc = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + cc];
            a.side.dst[i * a.side.ld_dst + cc] = a.side.mode ? fabsf(v) : v;
        }
    }
    const float* __restrict__ tab = a.table + 4 * j;
    for (int row = wave; row < a.rows; row += n_waves) {
        const int begin = a.rowptr[row], end = a.rowptr[row + 1];
        f32x4 s[VPL];
#pragma unroll
        for (int v = 0; v < VPL; ++v) s[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int base = begin; base < end; base += kWave) {
            const int mine = base + lane;
            const int c = mine < end ? (int)a.col[mine] : 0;
            const float cf = mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f;
            const int left = end - base - 4 * q;                 // pairs of this quad that exist (may be <= 0)
            int cc[4], ff[4];
            cc[0] = agg_dpp<0x00>(c); cc[1] = agg_dpp<0x55>(c); cc[2] = agg_dpp<0xAA>(c); cc[3] = agg_dpp<0xFF>(c);
            const int fi = __float_as_int(cf);
            ff[0] = agg_dpp<0x00>(fi); ff[1] = agg_dpp<0x55>(fi); ff[2] = agg_dpp<0xAA>(fi); ff[3] = agg_dpp<0xFF>(fi);
            f32x4 t[4][VPL];
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int v = 0; v < VPL; ++v) {
                    t[k][v] = (f32x4){0.f, 0.f, 0.f, 0.f};
                    if (left > k) t[k][v] = *reinterpret_cast<const f32x4*>(tab + (int64_t)cc[k] * a.ld_table + 16 * v);
                }
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int v = 0; v < VPL; ++v) s[v] += __int_as_float(ff[k]) * t[k][v];
        }
        // fold the 16 quads (lanes with equal j): butterfly over lane bits 2 .. 5, every lane ends with the total
#pragma unroll
        for (int off = 4; off < kWave; off <<= 1)
#pragma unroll
            for (int v = 0; v < VPL; ++v)
#pragma unroll
                for (int c = 0; c < 4; ++c) s[v][c] += __shfl_xor(s[v][c], off);
        // output column q: this lane's slice of the contraction, then the quad
        float part = 0.f;
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
            for (int c = 0; c < 4; ++c) part += s[v][c] * wreg[v][c];
        part = agg_dpp_add<0xB1>(part);                          // quad_perm [1,0,3,2]
        part = agg_dpp_add<0x4E>(part);                          // quad_perm [2,3,0,1]
        if (j == 0) {
            float val = part + bias;
            if (a.relu) val = fmaxf(val, 0.f);
            a.out[(int64_t)row * a.ld_out + q] = val;
        }
    }
}

// FIN in {16, 32, 64}, FOUT in {16, 32} with at least 4 input features per lane slice
inline bool transform_fusable(int64_t fin, int64_t fout) {
    if (fast_paths_disabled()) return false;
    // exactly the specialisations launch_aggregate_transform has (16 -> 32 has none: its K slice per lane would be two floats)
    return (fout == 16 && (fin == 16 || fin == 32 || fin == 64)) || (fout == 32 && (fin == 32 || fin == 64));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- aggregate, then transform, for wide layers: (A_norm x) W on the matrix cores ------------------------------
// The layers of the node-classification models whose output is at least as wide as their input (64 -> 64, 128 -> 128:
// the second layer of every homogeneous stack): transform-first pays a tall-skinny product (a launch, N x in read, N x out
// written: 14-19 us at 50,000 nodes) before it gathers rows of the SAME width.  Here the input rows are gathered (lane
// groups own rows as in k_aggregate_group), a block of 64 / 32 aggregated rows is staged in LDS and contracted with W on
// v_mfma_f32_16x16x32_bf16: both operands in three bf16 terms, six products, fp32 accumulators (the arithmetic of
// gn_gemm_f32's tall-skinny kernel; GN_GEMM_ARITH_FAST's two terms are not offered here).  W is split once per
// workgroup into LDS fragments; one persistent workgroup of sixteen waves per compute unit walks the row blocks; per block
// the sixteen (row tile, column tile) products are one per wave (two for 64 -> 128).
// 64 input features: W is 24 KB of fragments, so a workgroup per row block (no persistent loop, no 3.05 blocks in 4 rounds)
// with four row gathers in flight per lane measured 169.3 against 171.6 us on the aminer-syn forward (persistent, eight in
// flight; one block per workgroup with eight: 174); 128 features (96 KB of W) stay persistent
#ifndef GN_MFMA_U16
#define GN_MFMA_U16 4
#endif
#ifndef GN_MFMA_PERSIST16
#define GN_MFMA_PERSIST16 0
#endif
template <int LPE>
__global__ __launch_bounds__(1024) void k_aggregate_mfma(AggArgs a, const float* __restrict__ w, int fout, int row_blocks) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    constexpr int FIN = 4 * LPE, S = kWave / LPE, RPI = 16 * S, MT = RPI / 16, CH = FIN / 32, U = LPE == 16 ? GN_MFMA_U16 : 8;
    constexpr int STRIDE = FIN + 4;                            // floats between staged rows: 16 rows of a tile on 16 different bank quads
    extern __shared__ f32x4 lds_mfma[];
    u32x4* wsplit = reinterpret_cast<u32x4*>(lds_mfma);        // [CH][fout / 16][3][64]
    const int nt_all = fout >> 4;
    float* stage0 = reinterpret_cast<float*>(wsplit + (size_t)CH * nt_all * 3 * 64);  // [2][RPI][STRIDE]: blocks alternate, ONE barrier a block
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int slot = lane / LPE, j = lane % LPE;

    auto split3 = [](float x0, float x1, uint32_t (&t)[3]) {  // truncation split: x = hi + mid + lo exactly
        const uint32_t a0 = __builtin_bit_cast(uint32_t, x0), a1 = __builtin_bit_cast(uint32_t, x1);
        t[0] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
        const float r0 = x0 - __builtin_bit_cast(float, a0 & 0xffff0000u), r1 = x1 - __builtin_bit_cast(float, a1 & 0xffff0000u);
        const uint32_t b0 = __builtin_bit_cast(uint32_t, r0), b1 = __builtin_bit_cast(uint32_t, r1);
        t[1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const float s0 = r0 - __builtin_bit_cast(float, b0 & 0xffff0000u), s1 = r1 - __builtin_bit_cast(float, b1 & 0xffff0000u);
        t[2] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, s1), __builtin_bit_cast(uint32_t, s0), 0x07060302u);
    };

    // W as B-operand fragments: lane (n = l & 15, kg = l >> 4) of (chunk, column tile) holds k = 32 chunk + 8 kg .. + 7 of column 16 tile + n
    for (int idx = tid; idx < CH * nt_all * 64; idx += 1024) {
        const int l = idx & 63, t = (idx >> 6) % nt_all, ch = idx / (64 * nt_all);
        const int col = 16 * t + (l & 15), kb = 32 * ch + 8 * (l >> 4);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = w[(int64_t)(kb + q) * fout + col];
        u32x4 tv[3];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            uint32_t t3[3];
            split3(v[2 * h], v[2 * h + 1], t3);
            tv[0][h] = t3[0]; tv[1][h] = t3[1]; tv[2][h] = t3[2];
        }
        u32x4* o = wsplit + ((size_t)(ch * nt_all + t) * 3) * 64 + l;
        o[0] = tv[0]; o[64] = tv[1]; o[128] = tv[2];
    }
    if (a.side.dst) {                                          // concat slot: streamed up front by the whole grid
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + tid; t < total; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / a.side.cols, c = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + c];
            a.side.dst[i * a.side.ld_dst + c] = a.side.mode ? fabsf(v) : v;
        }
    }
    __syncthreads();

    const float* __restrict__ tab = a.table + 4 * j;
    int flip = 0;
    for (int rb = blockIdx.x; rb < row_blocks; rb += gridDim.x, flip ^= 1) {
        // (two staging buffers: a wave that is done with the products of block i gathers block i + 1 at once and writes
        // the other buffer; the buffer of block i - 1 was read by every wave before it arrived at the barrier of block i)
        float* stage = stage0 + (size_t)flip * RPI * STRIDE;
        // ---- gather: this lane group's row of the block (k_aggregate_group) ----
        const int local = wave * S + slot, row = rb * RPI + local;
        const bool live = row < a.rows;
        const int begin = live ? a.rowptr[row] : 0, end = live ? a.rowptr[row + 1] : 0;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = begin; __any(base < end); base += LPE) {
            const int mine = base + j;
            const uint32_t c = mine < end ? a.col[mine] : 0u;
            const float v = mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f;
            const int cnt = min(LPE, end - base);
            for (int t0 = 0; __any(t0 < cnt); t0 += U) {
                float4 r[U];
                float vv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint32_t cc = (uint32_t)__shfl((int)c, t0 + u, LPE);
                    vv[u] = __shfl(v, t0 + u, LPE);
                    r[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (t0 + u < cnt) r[u] = *reinterpret_cast<const float4*>(tab + (int64_t)cc * a.ld_table);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    acc.x += vv[u] * r[u].x; acc.y += vv[u] * r[u].y; acc.z += vv[u] * r[u].z; acc.w += vv[u] * r[u].w;
                }
            }
        }
        *reinterpret_cast<float4*>(stage + (size_t)local * STRIDE + 4 * j) = acc;
        __syncthreads();
        // ---- contract the block with W: (row tile, column tile) products dealt to the waves ----
        const int m = lane & 15, kg = lane >> 4;
        for (int job = wave; job < MT * nt_all; job += 16) {
            const int mt = job % MT, nt = job / MT;
            f32x4 d = (f32x4)(0.f);
            const float* arow = stage + (size_t)(16 * mt + m) * STRIDE + 8 * kg;
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 32 * ch), a1 = *reinterpret_cast<const f32x4*>(arow + 32 * ch + 4);
                u32x4 at[3];
                uint32_t t3[3];
                split3(a0[0], a0[1], t3); at[0][0] = t3[0]; at[1][0] = t3[1]; at[2][0] = t3[2];
                split3(a0[2], a0[3], t3); at[0][1] = t3[0]; at[1][1] = t3[1]; at[2][1] = t3[2];
                split3(a1[0], a1[1], t3); at[0][2] = t3[0]; at[1][2] = t3[1]; at[2][2] = t3[2];
                split3(a1[2], a1[3], t3); at[0][3] = t3[0]; at[1][3] = t3[1]; at[2][3] = t3[2];
                const bf16x8 xh = __builtin_bit_cast(bf16x8, at[0]), xm = __builtin_bit_cast(bf16x8, at[1]), xl = __builtin_bit_cast(bf16x8, at[2]);
                const u32x4* bp = wsplit + ((size_t)(ch * nt_all + nt) * 3) * 64 + lane;
                const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[0]), bm = __builtin_bit_cast(bf16x8, bp[64]), bl = __builtin_bit_cast(bf16x8, bp[128]);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh, d, 0, 0, 0);     // smallest terms first
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, bm, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, bh, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bm, d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh, d, 0, 0, 0);
            }
            const int col = 16 * nt + m;                       // D: column = lane & 15, rows 4 (lane >> 4) + i
            const float bias = a.bias ? a.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int orow = rb * RPI + 16 * mt + 4 * kg + i;
                if (orow < a.rows) {
                    float val = d[i] + bias;
                    if (a.relu) val = fmaxf(val, 0.f);
                    a.out[(int64_t)orow * a.ld_out + col] = val;
                }
            }
        }
    }
}

// in -> out widths the matrix-core form takes (gn_graph_aggregate_f32 with weight != NULL); rows of up to
// GN_AGG_GROUP_MAX_DEG neighbours on average (longer rows belong to a wave each: the product first, then k_aggregate)
inline bool mfma_fusable(int64_t fin, int64_t fout, int64_t rows, int64_t nnz) {
    if (fast_paths_disabled()) return false;
    if (!(fin == 64 || fin == 128) || fout % 16 != 0 || fout < fin || fout > 128) return false;
    return nnz >= 0 && nnz < GN_AGG_GROUP_MAX_DEG * rows && rows >= 4096;
}

inline gn_status launch_aggregate_mfma(const AggArgs& a, const float* w, int fout, hipStream_t st) {
    if (a.rows == 0) return GN_OK;
    const int lpe = a.features / 4, rpi = 16 * (kWave / lpe);
    const int row_blocks = (int)ceil_div(a.rows, rpi);
    const size_t lds = (size_t)a.features * fout * 6 + 2 * (size_t)rpi * (a.features + 4) * sizeof(float);
    int grid = std::min(row_blocks, compute_units());
    if (lpe == 16 && !GN_MFMA_PERSIST16) grid = row_blocks;
    gn_status ls;
    if (lpe == 16) {
        ls = allow_large_lds(reinterpret_cast<const void*>(k_aggregate_mfma<16>), 160 * 1024); if (ls != GN_OK) return ls;
        k_aggregate_mfma<16><<<grid, 1024, lds, st>>>(a, w, fout, row_blocks);
    } else {
        ls = allow_large_lds(reinterpret_cast<const void*>(k_aggregate_mfma<32>), 160 * 1024); if (ls != GN_OK) return ls;
        k_aggregate_mfma<32><<<grid, 1024, lds, st>>>(a, w, fout, row_blocks);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// GN_DISABLE_QUAD=1: the shuffle-based kernel for 16- and 32-wide rows as well (parity tests cover both)
inline bool quad_gather_disabled() {
    const char* e = getenv("GN_DISABLE_QUAD");
    return e && e[0] == '1';
}

// (the quad kernel does not write split planes: its caller follows it with the stand-alone split)
inline bool transform_takes_quad_kernel(int64_t fin, int64_t fout) { return fin == 16 && fout == 16 && !quad_gather_disabled(); }

inline gn_status launch_aggregate_transform(const AggArgs& a, const float* w, int fout, hipStream_t st) {
    if (a.rows == 0) return GN_OK;
    const int grid = (int)std::min<int64_t>(ceil_div(a.rows, 4), GN_AGG_GRID);
    const int key = a.features * 100 + fout;
    // 16-wide rows: quad gathers (14.8 vs 16.4 us on the second gene layer of pose0-syn); 32-wide rows are faster
    // with one 16-byte load per lane and neighbour (k_aggregate_transform<8, 16> 19.2 us, the quad form 21.5)
    if (key == 1616 && a.ld_table % 4 == 0 && aligned16(a.table) && !quad_gather_disabled()) {
        k_aggregate_transform_q<1><<<grid, 256, 0, st>>>(a, w);
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    switch (key) {
        case 1616: k_aggregate_transform<4, 16><<<grid, 256, 0, st>>>(a, w); break;
        case 3216: k_aggregate_transform<8, 16><<<grid, 256, 0, st>>>(a, w); break;
        case 6416: k_aggregate_transform<16, 16><<<grid, 256, 0, st>>>(a, w); break;
        case 3232: k_aggregate_transform<8, 32><<<grid, 256, 0, st>>>(a, w); break;
        case 6432: k_aggregate_transform<16, 32><<<grid, 256, 0, st>>>(a, w); break;
        default: return fail(GN_ERR_UNSUPPORTED, "no fused transform for %d -> %d features", a.features, fout);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}


// Many short rows over a SMALL table (the (relation, source) sums of the relational layer's weight gradient: 6 x 10^5
// rows of ~3 edges gathering from the 645 x 32 gradient rows): the table goes into LDS once per workgroup, and a row's
// neighbours cost LDS reads instead of L2 round trips.  LPE lanes own a row (16 bytes of it each), 64 / LPE rows per
// wave side by side; a wave works on two batches of rows at a time and reads the row bounds of the batches after them
// while it does (the chain bounds -> ids -> table would otherwise be paid per batch).  Unit coefficients only.
template <int LPE>
__global__ __launch_bounds__(1024) void k_aggregate_lds_table(AggArgs a) {
    constexpr int S = kWave / LPE;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ f32x4 tab[];
    const int tid = threadIdx.x, lane = tid & 63, slot = lane / LPE, j = lane % LPE;
    const bool col_live = 4 * j < a.features;
    const int tj = col_live ? j : 0;
    const int units = a.features / 4;                                  // float4 per table row
    const int n_tab = (int)a.table_rows * units;
    // (every workgroup reads the same table at the same time: each starts at its own offset, so that they do not all
    // queue on the same L2 channel)
    const int rot = (int)((blockIdx.x * 977u) % (unsigned)n_tab);
    for (int k = tid; k < n_tab; k += 1024) {
        int i = k + rot;
        i = i < n_tab ? i : i - n_tab;
        const int r = i / units, c = i - r * units;
        tab[i] = *reinterpret_cast<const f32x4*>(a.table + (int64_t)r * a.ld_table + 4 * c);
    }
    if (tid < units) tab[n_tab + tid] = (f32x4){0.f, 0.f, 0.f, 0.f};    // one zero row: what the slots past a row's end read
    __syncthreads();
    const int wave = (int)blockIdx.x * 16 + (tid >> 6), n_waves = (int)gridDim.x * 16;
    const int stride = n_waves * S;
    const uint32_t zero_row = (uint32_t)a.table_rows;
    int rowA = wave * S + slot, rowB = rowA + stride;
    // (every load is unconditional with a clamped index: hipcc waits for conditional loads one by one)
    const int last_row = a.rows - 1, last_id = (int)a.nnz - 1;
    int bA = a.rowptr[min(rowA, last_row)], eA = a.rowptr[min(rowA, last_row) + 1];
    int bB = a.rowptr[min(rowB, last_row)], eB = a.rowptr[min(rowB, last_row) + 1];
    if (rowA >= a.rows) eA = bA;
    if (rowB >= a.rows) eB = bB;
    // software pipeline: the bounds run two pairs of batches ahead of the sums, the first eight ids of every row one
    // pair ahead (the chain bounds -> ids -> table would otherwise be one HBM round trip after the other in every trip).
    // Every lane of a row reads the row's ids itself, eight at a time as two 16-byte loads (4-byte aligned: the plan's
    // id array has eight spare entries): handing them round with ds_bpermute cost more LDS time than the table reads.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    auto ids8 = [&](int first, u32x4& lo, u32x4& hi) {
        const uint32_t* __restrict__ p = a.col + min(first, last_id);
        __builtin_memcpy(&lo, p, 16);
        __builtin_memcpy(&hi, p + 4, 16);
    };
    int nA = rowA + 2 * stride, nB = rowB + 2 * stride;
    int nbA = a.rowptr[min(nA, last_row)], neA = a.rowptr[min(nA, last_row) + 1];
    int nbB = a.rowptr[min(nB, last_row)], neB = a.rowptr[min(nB, last_row) + 1];
    if (nA >= a.rows) neA = nbA;
    if (nB >= a.rows) neB = nbB;
    u32x4 iA0, iA1, iB0, iB1;
    ids8(bA, iA0, iA1);
    ids8(bB, iB0, iB1);
    const uint32_t piece = (uint32_t)tj;
    while (__any(rowA < a.rows)) {
        const int mA = nA + 2 * stride, mB = nB + 2 * stride;
        int mbA = a.rowptr[min(mA, last_row)], meA = a.rowptr[min(mA, last_row) + 1];
        int mbB = a.rowptr[min(mB, last_row)], meB = a.rowptr[min(mB, last_row) + 1];
        if (mA >= a.rows) meA = mbA;
        if (mB >= a.rows) meB = mbB;
        u32x4 jA0, jA1, jB0, jB1;                                      // the next pair's first ids
        ids8(nbA, jA0, jA1);
        ids8(nbB, jB0, jB1);
        f32x4 accA = (f32x4){0.f, 0.f, 0.f, 0.f}, accB = accA;
        // a slot past the row's end names the zero row, so the adds need no predicate
        for (int base = 0; __any(bA + base < eA || bB + base < eB); base += 8) {
            if (base > 0) { ids8(bA + base, iA0, iA1); ids8(bB + base, iB0, iB1); }      // (rows of more than eight edges)
            const int leftA = eA - bA - base, leftB = eB - bB - base;
            f32x4 vA[8], vB[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const uint32_t ia = t < 4 ? iA0[t & 3] : iA1[t & 3], ib = t < 4 ? iB0[t & 3] : iB1[t & 3];
                vA[t] = tab[(t < leftA ? ia : zero_row) * (uint32_t)units + piece];
                vB[t] = tab[(t < leftB ? ib : zero_row) * (uint32_t)units + piece];
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                accA += vA[t];
                accB += vB[t];
            }
        }
        {
        if (rowA < a.rows && col_live) *reinterpret_cast<f32x4*>(a.out + (int64_t)rowA * a.ld_out + 4 * j) = accA;
        if (rowB < a.rows && col_live) *reinterpret_cast<f32x4*>(a.out + (int64_t)rowB * a.ld_out + 4 * j) = accB;
        }
        rowA = nA; rowB = nB; bA = nbA; eA = neA; bB = nbB; eB = neB;
        nA = mA; nB = mB; nbA = mbA; neA = meA; nbB = mbB; neB = meB;
        iA0 = jA0; iA1 = jA1; iB0 = jB0; iB1 = jB1;
    }
}

inline bool lds_table_disabled() {
    const char* e = getenv("GN_DISABLE_LDS_TABLE");
    return e && e[0] == '1';
}

template <int VEC>
inline void launch_aggregate_lpe(const AggArgs& a, int lpe, int grid, hipStream_t st) {
    switch (lpe) {
        case 1: k_aggregate<VEC, 1><<<grid, 256, 0, st>>>(a); break;
        case 2: k_aggregate<VEC, 2><<<grid, 256, 0, st>>>(a); break;
        case 4: k_aggregate<VEC, 4><<<grid, 256, 0, st>>>(a); break;
        case 8: k_aggregate<VEC, 8><<<grid, 256, 0, st>>>(a); break;
        case 16: k_aggregate<VEC, 16><<<grid, 256, 0, st>>>(a); break;
        case 32: k_aggregate<VEC, 32><<<grid, 256, 0, st>>>(a); break;
        default: k_aggregate<VEC, 64><<<grid, 256, 0, st>>>(a); break;
    }
}

inline gn_status check_side(const gn_side_copy* side, int64_t rows, gn_side_copy* out) {
    *out = gn_side_copy{nullptr, 0, nullptr, 0, 0, 0, 0};
    if (!side || side->rows == 0 || side->cols == 0) return GN_OK;
    GN_REQUIRE(side->src && side->dst && side->rows > 0 && side->cols > 0, "side copy has a null pointer or a negative size");
    GN_REQUIRE(side->rows <= rows, "side copy has %lld rows, the launch only %lld", (long long)side->rows, (long long)rows);
    GN_REQUIRE(side->ld_src >= side->cols && side->ld_dst >= side->cols, "side copy leading dimension smaller than its row");
    GN_REQUIRE(side->mode == 0 || side->mode == 1, "unknown side copy mode %d", side->mode);
    *out = *side;
    return GN_OK;
}

inline gn_status launch_aggregate(const AggArgs& a, hipStream_t st) {
    if (a.rows == 0 || a.features == 0) return GN_OK;
    const bool vec = (a.features % 4 == 0) && (a.ld_table % 4 == 0) && (a.ld_out % 4 == 0) && aligned16(a.table) &&
                     aligned16(a.out);
    const int units = vec ? a.features / 4 : a.features;
    int lpe = 1;
    while (lpe < units && lpe < kWave) lpe <<= 1;
    const int grid = (int)std::min<int64_t>(ceil_div(a.rows, 4), GN_AGG_GRID);
    // many short rows over a table that fits the LDS, nothing but a plain sum: the table is gathered from there
    if (vec && lpe >= 4 && lpe <= 16 && a.nnz > 0 && a.nnz < 8 * (int64_t)a.rows && a.rows >= 65536 && !a.coef && !a.rowdiv &&
        !a.addend && !a.bias && !a.relu && !a.side.dst && a.table_rows > 0 &&
        (size_t)a.table_rows * a.features * sizeof(float) <= 128 * 1024 && !fast_paths_disabled() && !lds_table_disabled()) {
        const size_t lds = (size_t)(a.table_rows + 1) * a.features * sizeof(float);
        const int wgs = (int)std::min<int64_t>(256, ceil_div((int64_t)a.rows * lpe, 2 * 1024));
        gn_status ls;
        switch (lpe) {
            case 4: ls = allow_large_lds(reinterpret_cast<const void*>(k_aggregate_lds_table<4>), 160 * 1024); if (ls != GN_OK) return ls;
                    k_aggregate_lds_table<4><<<wgs, 1024, lds, st>>>(a); break;
            case 8: ls = allow_large_lds(reinterpret_cast<const void*>(k_aggregate_lds_table<8>), 160 * 1024); if (ls != GN_OK) return ls;
                    k_aggregate_lds_table<8><<<wgs, 1024, lds, st>>>(a); break;
            default: ls = allow_large_lds(reinterpret_cast<const void*>(k_aggregate_lds_table<16>), 160 * 1024); if (ls != GN_OK) return ls;
                     k_aggregate_lds_table<16><<<wgs, 1024, lds, st>>>(a); break;
        }
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    if (vec && lpe <= 16 && a.nnz >= 0 && a.nnz < GN_AGG_SHORT_MAX_DEG * (int64_t)a.rows && !fast_paths_disabled()) {
        const int sgrid = (int)std::min<int64_t>(ceil_div((int64_t)a.rows * lpe, 256), GN_AGG_GRID);
        switch (lpe) {
            case 1: k_aggregate_short<1><<<sgrid, 256, 0, st>>>(a); break;
            case 2: k_aggregate_short<2><<<sgrid, 256, 0, st>>>(a); break;
            case 4: k_aggregate_short<4><<<sgrid, 256, 0, st>>>(a); break;
            case 8: k_aggregate_short<8><<<sgrid, 256, 0, st>>>(a); break;
            default: k_aggregate_short<16><<<sgrid, 256, 0, st>>>(a); break;
        }
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    if (vec && lpe <= 32 && a.nnz >= 0 && a.nnz < GN_AGG_GROUP_MAX_DEG * (int64_t)a.rows && !fast_paths_disabled()) {
        const int ggrid = (int)ceil_div((int64_t)a.rows * lpe, 256);        // one row per lane group, no grid-stride loop
        switch (lpe) {
            case 1: k_aggregate_group<1, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 2: k_aggregate_group<2, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 4: k_aggregate_group<4, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 8: k_aggregate_group<8, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 16: k_aggregate_group<16, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            default: k_aggregate_group<32, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
        }
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    if (vec) launch_aggregate_lpe<4>(a, lpe, grid, st); else launch_aggregate_lpe<1>(a, lpe, grid, st);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace gn
