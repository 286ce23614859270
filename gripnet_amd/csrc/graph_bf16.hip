// bf16 STORAGE of the gathered table for the GCN-style layers (BASELINE.json configs[4]).
// The reference has no reduced precision anywhere (fp32 throughout); this is the build's own definition:
// the table a layer gathers from - x W, rounded once to bf16 - is read at half the bytes, the sum over the
// neighbours, the bias and the activation stay fp32, and the layer's output is fp32 (the concat buffers, the
// decoders and every parameter are unchanged).  out = act( A_norm . bf16(xw) + b ).
#include "aggregate.cuh"

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));       // round to nearest even, NaN stays NaN
}

// dst[i, c] = bf16(src[i, c]); cols % 8 == 0, 16-byte aligned rows on both sides
__global__ __launch_bounds__(256) void k_cast_bf16(const float* __restrict__ src, int64_t ld_src, uint16_t* __restrict__ dst,
                                                  int64_t ld_dst, int64_t rows, int cols8) {
    const int64_t total = rows * cols8;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / cols8;
        const int c = (int)(t - i * cols8) * 8;
        const float4 a = *reinterpret_cast<const float4*>(src + i * ld_src + c);
        const float4 b = *reinterpret_cast<const float4*>(src + i * ld_src + c + 4);
        *reinterpret_cast<u32x4*>(dst + i * ld_dst + c) =
            (u32x4){pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w)};
    }
}

struct AggBf16Args {
    const int32_t* rowptr; const uint32_t* col; const float* coef;
    const uint16_t* table; int64_t ld_table; int features;
    const float* bias; int relu;
    float* out; int64_t ld_out; int rows;
    gn_side_copy side;
};

// One wave per destination row, 64 (neighbour, coefficient) pairs per coalesced load, S = 64 / LPE neighbour rows
// per group with 16 bytes = 8 bf16 features per lane, two groups requested ahead of their use, fp32 sums, fixed
// fold order (same scheme as k_aggregate).
template <int LPE>
__global__ __launch_bounds__(256) void k_aggregate_bf16(AggBf16Args a) {
    constexpr int S = gn::kWave / LPE, IT = gn::kWave / S, U = IT < 2 ? IT : 2;
    const int lane = threadIdx.x & 63;
    const int slot = lane / LPE, j = lane % LPE;
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
        for (int cb = 0; cb * LPE * 8 < a.features; ++cb) {
            const int fcol = (cb * LPE + j) * 8;
            const bool active = fcol < a.features;
            float acc[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = 0.f;
            for (int base = begin; base < end; base += gn::kWave) {
                const int mine = base + lane;
                const uint32_t c = mine < end ? a.col[mine] : 0u;
                const float v = mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f;
                const int cnt = min(gn::kWave, end - base);
                for (int it0 = 0; it0 * S < cnt; it0 += U) {
                    u32x4 t[U];
                    float vv[U];
#pragma unroll
                    for (int it = 0; it < U; ++it) {
                        const int idx = (it0 + it) * S + slot;
                        const uint32_t cc = (uint32_t)__shfl((int)c, idx);
                        vv[it] = __shfl(v, idx);
                        t[it] = (u32x4){0u, 0u, 0u, 0u};
                        if (idx < cnt && active) t[it] = *reinterpret_cast<const u32x4*>(a.table + (int64_t)cc * a.ld_table + fcol);
                    }
#pragma unroll
                    for (int it = 0; it < U; ++it)
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            acc[2 * k] += vv[it] * __uint_as_float(t[it][k] << 16);
                            acc[2 * k + 1] += vv[it] * __uint_as_float(t[it][k] & 0xffff0000u);
                        }
                }
            }
#pragma unroll
            for (int off = LPE; off < gn::kWave; off <<= 1) {
#pragma unroll
                for (int t = 0; t < 8; ++t) acc[t] += __shfl_xor(acc[t], off);
            }
            if (slot == 0 && active) {
#pragma unroll
                for (int t = 0; t < 8; ++t) {
                    float val = acc[t];
                    if (a.bias) val += a.bias[fcol + t];
                    if (a.relu) val = fmaxf(val, 0.f);
                    acc[t] = val;
                }
                float* dst = a.out + (int64_t)row * a.ld_out + fcol;
                *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
                *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
            }
        }
    }
}

// The many-short-rows form (k_aggregate_group of aggregate.cuh on a bf16 table): LPE lanes own a row - 64 / LPE rows per wave side
// by side, one row per group, no grid-stride loop - with 16 bytes = 8 bf16 features per lane; the group reads its (col, coef)
// pairs LPE at a time with one coalesced load and requests U neighbour rows before it consumes the first.  fp32 sums in
// neighbour order (bitwise reproducible).  Round 6: the wave-per-row kernel above paid the row's latency chain per wave and made
// bf16 storage SLOWER than fp32 (freebase-c-syn 414.7 against 381.7 us per forward).
template <int LPE, int U>
__global__ __launch_bounds__(256) void k_aggregate_group_bf16(AggBf16Args a) {
    constexpr int S = gn::kWave / LPE;
    const int lane = threadIdx.x & 63, slot = lane / LPE, j = lane % LPE;
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int fcol = 8 * j;
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
    const uint16_t* __restrict__ tab = a.table + fcol;
    float acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = 0.f;
    for (int base = begin; __any(base < end); base += LPE) {
        const int mine = base + j;
        const uint32_t c = mine < end ? a.col[mine] : 0u;
        const float v = mine < end ? (a.coef ? a.coef[mine] : 1.0f) : 0.f;
        const int cnt = min(LPE, end - base);                  // of this group (<= 0 once its row is done)
        for (int t0 = 0; __any(t0 < cnt); t0 += U) {
            u32x4 r[U];
            float vv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t cc = (uint32_t)__shfl((int)c, t0 + u, LPE);
                vv[u] = __shfl(v, t0 + u, LPE);
                r[u] = (u32x4){0u, 0u, 0u, 0u};
                if (t0 + u < cnt && active) r[u] = *reinterpret_cast<const u32x4*>(tab + (int64_t)cc * a.ld_table);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    acc[2 * k] += vv[u] * __uint_as_float(r[u][k] << 16);
                    acc[2 * k + 1] += vv[u] * __uint_as_float(r[u][k] & 0xffff0000u);
                }
        }
    }
    if (live && active) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float val = acc[t];
            if (a.bias) val += a.bias[fcol + t];
            if (a.relu) val = fmaxf(val, 0.f);
            acc[t] = val;
        }
        float* dst = a.out + (int64_t)row * a.ld_out + fcol;
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

}  // namespace

extern "C" {

gn_status gn_cast_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int64_t cols, void* stream) {
    GN_REQUIRE(rows >= 0 && cols >= 0, "negative size");
    if (rows == 0 || cols == 0) return GN_OK;
    GN_REQUIRE(src && dst, "operand pointer is null");
    GN_REQUIRE(cols % 8 == 0 && ld_src % 4 == 0 && ld_dst % 8 == 0 && ld_src >= cols && ld_dst >= cols && gn::aligned16(src) &&
               gn::aligned16(dst), "gn_cast_bf16 needs cols %% 8 == 0 and 16-byte aligned rows");
    k_cast_bf16<<<gn::stream_grid(rows * (cols / 8), 256), 256, 0, gn::as_stream(stream)>>>(src, ld_src, dst, ld_dst, rows, (int)(cols / 8));
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_graph_aggregate_bf16(const gn_graph_plan* plan, const uint16_t* table, int64_t ld_table, int64_t num_features,
                                  const float* bias, int relu, float* out, int64_t ld_out, const gn_side_copy* side, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(num_features >= 0 && num_features < (1ll << 31), "bad feature count");
    if (plan->rows == 0 || num_features == 0) return GN_OK;
    GN_REQUIRE(table && out, "feature pointers are null");
    GN_REQUIRE(num_features % 8 == 0 && ld_table % 8 == 0 && ld_out % 4 == 0 && ld_table >= num_features && ld_out >= num_features &&
               gn::aligned16(table) && gn::aligned16(out), "gn_graph_aggregate_bf16 needs features %% 8 == 0 and 16-byte aligned rows");
    AggBf16Args a;
    a.rowptr = plan->rowptr.p; a.col = reinterpret_cast<const uint32_t*>(plan->col.p); a.coef = plan->coef.p;
    a.table = table; a.ld_table = ld_table; a.features = (int)num_features; a.bias = bias; a.relu = relu;
    a.out = out; a.ld_out = ld_out; a.rows = (int)plan->rows;
    gn_status ss = gn::check_side(side, plan->rows, &a.side);
    if (ss != GN_OK) return ss;
    int lpe = 1;
    while (lpe < num_features / 8 && lpe < gn::kWave) lpe <<= 1;
    const int grid = (int)std::min<int64_t>(gn::ceil_div(a.rows, 4), GN_AGG_GRID);
    hipStream_t st = gn::as_stream(stream);
    if (lpe <= 32 && plan->nnz < GN_AGG_GROUP_MAX_DEG * plan->rows && plan->rows >= 4096 && !gn::fast_paths_disabled()) {
        // many short rows (the node-classification graphs): lane groups own rows, eight row gathers in flight per lane
        const int ggrid = (int)gn::ceil_div((int64_t)a.rows * lpe, 256);
        switch (lpe) {
            case 1: k_aggregate_group_bf16<1, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 2: k_aggregate_group_bf16<2, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 4: k_aggregate_group_bf16<4, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 8: k_aggregate_group_bf16<8, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            case 16: k_aggregate_group_bf16<16, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
            default: k_aggregate_group_bf16<32, GN_AGG_GROUP_U><<<ggrid, 256, 0, st>>>(a); break;
        }
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    switch (lpe) {
        case 1: k_aggregate_bf16<1><<<grid, 256, 0, st>>>(a); break;
        case 2: k_aggregate_bf16<2><<<grid, 256, 0, st>>>(a); break;
        case 4: k_aggregate_bf16<4><<<grid, 256, 0, st>>>(a); break;
        case 8: k_aggregate_bf16<8><<<grid, 256, 0, st>>>(a); break;
        case 16: k_aggregate_bf16<16><<<grid, 256, 0, st>>>(a); break;
        case 32: k_aggregate_bf16<32><<<grid, 256, 0, st>>>(a); break;
        default: k_aggregate_bf16<64><<<grid, 256, 0, st>>>(a); break;
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // extern "C"
