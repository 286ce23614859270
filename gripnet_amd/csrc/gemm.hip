// Dense fp32 contractions on the CDNA4 matrix cores, plus the small element-wise merges.
//
// v_mfma_f32_16x16x4_f32 is an exact fp32 FMA chain (no reduced-precision path exists on
// gfx950), so these GEMMs keep the reference's fp32 numerics.  One wave owns a 16-row x
// (up to 64)-column output tile: the A fragment (one float4 per lane per 16-deep K chunk) is
// reused across the column tiles.  The K index inside a chunk is permuted (lane group q,
// element j  <->  k = 4q + j) identically for A and B, which lets A be read with one 16-byte
// load per lane instead of four strided dwords.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kColTiles = 4;  // 16-column tiles per wave (A-fragment reuse)

struct GemmArgs {
    const float* __restrict__ a; int64_t lda, stride_a; const int64_t* __restrict__ a_rows; int64_t a_table_rows;
    const float* __restrict__ b; int64_t ldb, stride_b;
    float* __restrict__ c; int64_t ldc, stride_c;
    int m, n, k;
    const float* bias; int relu; int a_vec_ok;
    int64_t sbk, sbn;                     // element (k, col) of B sits at b[k * sbk + col * sbn]: (ldb, 1), or (1, ldb) for B given transposed
    int64_t sam, sak;                     // element (row, k) of A at a[row * sam + k * sak]: (lda, 1), or (1, lda) for A given transposed
    int accumulate;                       // c += a b instead of c = a b
    const float* addend; int64_t ld_add;  // (may be null) c = a b + addend: a second gradient of the same tensor, added where the product is stored
    int c_vec_ok;                         // c (and bias, addend) take 16-byte accesses at column multiples of four
    int out_bf16 = 0;                     // GN_GEMM_OUT_BF16: c is a bf16 table (ldc in bf16 elements), each value rounded to nearest even once
};

__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs g) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15;   // row inside the tile for A, column inside the tile for B / C
    const int q = lane >> 4;   // K group for A/B, row group for C
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    const int col0 = blockIdx.y * (16 * kColTiles);
    if (row0 >= g.m) return;   // wave-uniform
    const int64_t batch = blockIdx.z;
    const float* __restrict__ A = g.a + batch * g.stride_a;
    const float* __restrict__ B = g.b + batch * g.stride_b;
    float* __restrict__ C = g.c + batch * g.stride_c;

    // Every load below is unconditional (clamped index, zeroed by select afterwards): a load under a
    // condition is waited for on its own, an unconditional batch is issued back to back.
    const int arow = row0 + r;
    int64_t a_src_row = min(arow, g.m - 1);
    bool a_ok = arow < g.m;
    if (g.a_rows) {
        a_src_row = g.a_rows[a_src_row];
        a_ok = a_ok && (uint64_t)a_src_row < (uint64_t)g.a_table_rows;      // out of table -> zeros
        if (!a_ok) a_src_row = 0;
    }
    const float* __restrict__ arow_ptr = A + a_src_row * g.lda;
    const int n_tiles = min(kColTiles, (g.n - col0 + 15) / 16);             // wave-uniform

    f32x4 acc[kColTiles];
#pragma unroll
    for (int t = 0; t < kColTiles; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // K chunks of 16, software-pipelined: the loads of chunk k + 1 are in flight while the 16 MFMAs of chunk k run
    // (a load - wait - MFMA loop pays an L2 round trip per chunk: 25 us instead of ~10 on [50000 x 128] @ [128 x 64]).
    auto load_chunk = [&](int k0, float (&av)[4], float (&bv)[kColTiles][4]) {
        const int kb = k0 + 4 * q;
        if (g.a_vec_ok && k0 + 16 <= g.k) {                                 // wave-uniform
            const float4 t = *reinterpret_cast<const float4*>(arow_ptr + kb);
            av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) av[j] = arow_ptr[min(kb + j, g.k - 1)];
        }
#pragma unroll
        for (int t = 0; t < kColTiles; ++t) {
            const int col = min(col0 + 16 * t + r, g.n - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[t][j] = B[(int64_t)min(kb + j, g.k - 1) * g.sbk + (int64_t)col * g.sbn];
        }
    };
    float av[4], bv[kColTiles][4];
    load_chunk(0, av, bv);
    for (int k0 = 0; k0 < g.k; k0 += 16) {
        const int kb = k0 + 4 * q;
        float an[4], bn[kColTiles][4];
        load_chunk(k0 + 16 < g.k ? k0 + 16 : k0, an, bn);                   // the last trip re-reads its own chunk (unused)
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = (a_ok && kb + j < g.k) ? av[j] : 0.f;
#pragma unroll
        for (int t = 0; t < kColTiles; ++t) {
            if (t >= n_tiles) break;
            const bool col_ok = col0 + 16 * t + r < g.n;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], (col_ok && kb + j < g.k) ? bv[t][j] : 0.f, acc[t], 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) av[j] = an[j];
#pragma unroll
        for (int t = 0; t < kColTiles; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[t][j] = bn[t][j];
    }
#pragma unroll
    for (int t = 0; t < kColTiles; ++t) {
        const int col = col0 + 16 * t + r;
        if (col >= g.n) continue;
        const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = row0 + 4 * q + i;
            if (row < g.m) {
                float v = acc[t][i] + bias;
                if (g.accumulate) v += C[(int64_t)row * g.ldc + col];
                if (g.addend) v += g.addend[(int64_t)row * g.ld_add + col];
                if (g.relu) v = fmaxf(v, 0.f);
                C[(int64_t)row * g.ldc + col] = v;
            }
        }
    }
}

// Deep and narrow: one side of the output is a few tiles and K is hundreds to thousands (dbasis = att^T dW and datt =
// dW basis^T of the relational layer: 32 x 1536 over K = 964 and 964 x 32 over K = 1536).  k_gemm_f32 gives such a shape a
// handful of waves that each walk the whole K.  Here a workgroup owns a (16 MT) x (16 NT) tile, its sixteen waves take the
// 16-deep K chunks round-robin, their accumulators meet in LDS and are added in wave order.  Either operand may be given
// transposed (element strides); a side whose K runs contiguously is read 16 bytes per lane (the K index inside a chunk is
// permuted identically for A and B, as in k_gemm_f32).
constexpr int kDeepWaves = 16;

template <int MT, int NT>
__device__ __forceinline__ void gemm_deep_body(const GemmArgs& g, int bx, int by, f32x4* __restrict__ part) {   // part: [16 waves][MT * NT][64]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int row0 = bx * 16 * MT, col0 = by * 16 * NT;
    const int chunks = (g.k + 15) / 16;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // a wave's chunks (wave, wave + 16, ...), the loads of the next one in flight while the MFMAs of the current one run
    auto load_chunk = [&](int ch, float (&av)[MT][4], float (&bv)[NT][4]) {
        const int kb = 16 * ch + 4 * q;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int row = min(row0 + 16 * t + r, g.m - 1);
            const float* __restrict__ p = g.a + (int64_t)row * g.sam;
            if (g.sak == 1 && g.a_vec_ok && kb + 4 <= g.k) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + kb);
#pragma unroll
                for (int j = 0; j < 4; ++j) av[t][j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) av[t][j] = p[(int64_t)min(kb + j, g.k - 1) * g.sak];
            }
        }
#pragma unroll
        for (int u = 0; u < NT; ++u) {
            const int col = min(col0 + 16 * u + r, g.n - 1);
            const float* __restrict__ p = g.b + (int64_t)col * g.sbn;
            if (g.sbk == 1 && ((g.sbn & 3) == 0) && ((reinterpret_cast<uintptr_t>(g.b) & 15) == 0) && kb + 4 <= g.k) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(p + kb);
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[u][j] = v[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[u][j] = p[(int64_t)min(kb + j, g.k - 1) * g.sbk];
            }
        }
    };
    float av[MT][4], bv[NT][4];
    if (wave < chunks) load_chunk(wave, av, bv);
    for (int ch = wave; ch < chunks; ch += kDeepWaves) {
        const int kb = 16 * ch + 4 * q;
        float an[MT][4], bn[NT][4];
        load_chunk(ch + kDeepWaves < chunks ? ch + kDeepWaves : ch, an, bn);          // (the last trip re-reads its own chunk, unused)
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) av[t][j] = (row0 + 16 * t + r < g.m && kb + j < g.k) ? av[t][j] : 0.f;
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][j] = (col0 + 16 * u + r < g.n && kb + j < g.k) ? bv[u][j] : 0.f;
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int u = 0; u < NT; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][j], bv[u][j], acc[t][u], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) av[t][j] = an[t][j];
#pragma unroll
        for (int u = 0; u < NT; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) bv[u][j] = bn[u][j];
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) part[(wave * MT * NT + t * NT + u) * 64 + lane] = acc[t][u];
    __syncthreads();
    // element i of lane l of a tile: row 4 (l >> 4) + i, column l & 15; thread -> (tile, lane, element), the waves in order
    for (int o = threadIdx.x; o < MT * NT * 256; o += kDeepWaves * 64) {
        const int tile = o >> 8, l = (o & 255) >> 2, i = o & 3;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kDeepWaves; ++w) v += part[(w * MT * NT + tile) * 64 + l][i];
        const int row = row0 + 16 * (tile / NT) + 4 * (l >> 4) + i, col = col0 + 16 * (tile % NT) + (l & 15);
        if (row < g.m && col < g.n) {
            if (g.bias) v += g.bias[col];
            float* c = g.c + (int64_t)row * g.ldc + col;
            if (g.accumulate) v += *c;
            if (g.addend) v += g.addend[(int64_t)row * g.ld_add + col];
            if (g.relu) v = fmaxf(v, 0.f);
            *c = v;
        }
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(kDeepWaves * 64) void k_gemm_deep(GemmArgs g) {
    __shared__ f32x4 part[kDeepWaves * MT * NT * 64];
    gemm_deep_body<MT, NT>(g, blockIdx.x, blockIdx.y, part);
}

// Tall-skinny form (one shared B of at most 64 KB per 64-column block: every layer's x @ W): B is laid out ONCE per
// workgroup in LDS as MFMA B fragments ([16-deep K chunk][column tile][lane] float4), so that a chunk costs one
// 16-byte global load (A) and kColTiles ds_read_b128 per lane instead of 16 four-byte global loads whose issue - not
// the MFMAs - bounded k_gemm_f32 (17 load instructions of 16 cycles per 16 MFMAs of 8 cycles of the CU).  Workgroups
// are persistent over the row tiles; A is read one chunk ahead.
// (bx of nbx workgroups of `waves` waves each walk the row tiles; by: the 64-column block)
__device__ __forceinline__ void gemm_lds_body(const GemmArgs& g, int row_tiles, int bx, int by, int nbx, int waves, f32x4* __restrict__ bfrag) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int col0 = by * (16 * kColTiles);
    const int chunks = (g.k + 15) / 16;
    const int n_tiles = min(kColTiles, (g.n - col0 + 15) / 16);
    for (int idx = threadIdx.x; idx < chunks * kColTiles * 64; idx += 64 * waves) {
        const int l = idx & 63, t = (idx >> 6) % kColTiles, ch = idx / (64 * kColTiles);
        const int col = col0 + 16 * t + (l & 15), kb = 16 * ch + 4 * (l >> 4);
        f32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (col < g.n && kb + j < g.k) ? g.b[(int64_t)(kb + j) * g.sbk + (int64_t)col * g.sbn] : 0.f;
        bfrag[idx] = v;
    }
    __syncthreads();
    for (int tile = bx * waves + wave; tile < row_tiles; tile += nbx * waves) {
        const int row0 = tile * 16;
        const int arow = row0 + r;
        int64_t a_src_row = min(arow, g.m - 1);
        bool a_ok = arow < g.m;
        if (g.a_rows) {
            a_src_row = g.a_rows[a_src_row];
            a_ok = a_ok && (uint64_t)a_src_row < (uint64_t)g.a_table_rows;      // out of table -> zeros
            if (!a_ok) a_src_row = 0;
        }
        const float* __restrict__ arow_ptr = g.a + a_src_row * g.lda;
        auto load_a = [&](int ch) {
            const int kb = 16 * ch + 4 * q;
            f32x4 v;
            if (g.a_vec_ok && 16 * ch + 16 <= g.k) {
                v = *reinterpret_cast<const f32x4*>(arow_ptr + kb);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = arow_ptr[min(kb + j, g.k - 1)];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = kb + j < g.k ? v[j] : 0.f;
            }
            return a_ok ? v : (f32x4){0.f, 0.f, 0.f, 0.f};
        };
        f32x4 acc[kColTiles];
#pragma unroll
        for (int t = 0; t < kColTiles; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        f32x4 av = load_a(0);
        for (int ch = 0; ch < chunks; ++ch) {
            const f32x4 an = load_a(ch + 1 < chunks ? ch + 1 : ch);
            const f32x4* __restrict__ bp = bfrag + (size_t)ch * kColTiles * 64 + lane;
#pragma unroll
            for (int t = 0; t < kColTiles; ++t) {
                if (t >= n_tiles) break;
                const f32x4 bv = bp[t * 64];
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv[j], acc[t], 0, 0, 0);
            }
            av = an;
        }
#pragma unroll
        for (int t = 0; t < kColTiles; ++t) {
            const int col = col0 + 16 * t + r;
            if (col >= g.n) continue;
            const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * q + i;
                if (row < g.m) {
                    float v = acc[t][i] + bias;
                    if (g.accumulate) v += g.c[(int64_t)row * g.ldc + col];
                    if (g.addend) v += g.addend[(int64_t)row * g.ld_add + col];
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.c[(int64_t)row * g.ldc + col] = v;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_gemm_f32_lds(GemmArgs g, int row_tiles) {
    extern __shared__ f32x4 bfrag[];                          // [chunks][kColTiles][64]
    gemm_lds_body(g, row_tiles, blockIdx.x, blockIdx.y, gridDim.x, 4, bfrag);
}


// The same tall-skinny product on the bf16 matrix instruction: every fp32 operand is split into bf16 hi + lo and the
// product is hi.hi + hi.lo + lo.hi in fp32 accumulators (error <= 2^-17 per product, as in k_rgcn_acc; the contract of
// the path is 1e-4).  v_mfma_f32_16x16x4_f32 runs at the fp32 vector rate - 20+ us of matrix time on
// [50000 x 128] @ [128 x 64] - the three v_mfma_f32_16x16x32_bf16 take 3/16 of that and the product becomes a stream
// over A.  B sits in LDS as hi / lo fragments (16 bytes per lane: eight consecutive k of one column); A is read one
// 32-deep chunk ahead (two 16-byte loads per lane) and split in registers.  GN_GEMM_EXACT=1 keeps the fp32 instruction.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// TERMS = 3 (default): v = hi + mid + lo, six products (hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi): what is dropped is
//             below 2^-23 of a product, fp32's own rounding - the fp32-faithful mode.
// TERMS = 2 (GN_GEMM_ARITH_FAST): v = hi + lo, three products, <= 2^-16 per product.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {            // quad_perm exchange inside every quad of lanes
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}

template <int TERMS>
__device__ __forceinline__ void split_terms(float a, float b, uint32_t (&t)[3]) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    t[0] = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {a - __uint_as_float(t[0] << 16), b - __uint_as_float(t[0] & 0xffff0000u)};
    t[1] = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
    if constexpr (TERMS == 3) {
        const f32x2 q = {r[0] - __uint_as_float(t[1] << 16), r[1] - __uint_as_float(t[1] & 0xffff0000u)};
        t[2] = __builtin_bit_cast(uint32_t, __builtin_convertvector(q, bf16x2));
    } else {
        t[2] = 0u;
    }
}

// One workgroup of sixteen waves per compute unit, persistent: B is fetched and split once per compute unit, row tile t
// goes to wave t / grid of workgroup t % grid (a [50000 x K] product is 3125 tiles: at most one per wave, spread over all
// compute units).  The B words are requested first, the wave's first A tile right behind them (requests return in
// order: B is split and laid out in LDS while A is still on its way), and every chunk of the next tile is requested as
// soon as the current tile's chunk has been split into its bf16 terms.
constexpr int kSplitThreads = 1024, kSplitWaves = kSplitThreads / 64;

template <int CH, int TERMS, int CT>   // CT: 16-column tiles per wave (4, or 8: a 128-column block reads A once); CH: chunks of 32 k known at compile time (a whole row tile of A in registers), 0: any number, one chunk ahead
__global__ __launch_bounds__(kSplitThreads) void k_gemm_split_lds(GemmArgs g, int row_tiles, int slab) {
    extern __shared__ f32x4 bfrag[];                          // [chunks][CT][TERMS][64] as 16-byte words
    u32x4* bsplit = reinterpret_cast<u32x4*>(bfrag);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int col0 = blockIdx.y * (16 * CT);
    const int chunks = CH > 0 ? CH : g.k / 32;                // k is a multiple of 32 here
    const int n_tiles = min(CT, (g.n - col0 + 15) / 16);
    const int tile_step = gridDim.x * kSplitWaves;
    int tile = wave * gridDim.x + blockIdx.x;

    auto a_ptr = [&](int t) { return g.a + (int64_t)min(t * 16 + r, g.m - 1) * g.lda + 8 * q; };
    auto put_b = [&](int idx, const float (&v)[8]) {
        const int l = idx & 63, t = (idx >> 6) % CT, ch = idx / (64 * CT);
        u32x4 tv[3];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            uint32_t w[3];
            split_terms<TERMS>(v[2 * h], v[2 * h + 1], w);
            tv[0][h] = w[0]; tv[1][h] = w[1]; tv[2][h] = w[2];
        }
        u32x4* o = bsplit + ((size_t)(ch * CT + t) * TERMS) * 64 + l;
#pragma unroll
        for (int q3 = 0; q3 < TERMS; ++q3) o[64 * q3] = tv[q3];
    };
    auto get_b = [&](int idx, float (&v)[8]) {
        const int l = idx & 63, t = (idx >> 6) % CT, ch = idx / (64 * CT);
        const int col = col0 + 16 * t + (l & 15), kb = 32 * ch + 8 * (l >> 4);
        if (g.sbk == 1 && (g.sbn & 3) == 0 && (reinterpret_cast<uintptr_t>(g.b) & 15) == 0) {
            // B given transposed: the lane's eight k are 32 contiguous bytes (two 16-byte loads, not eight 4-byte ones at the
            // lanes' row stride)
            const f32x4* __restrict__ pb = reinterpret_cast<const f32x4*>(g.b + (int64_t)min(col, g.n - 1) * g.sbn + kb);
            const f32x4 lo = pb[0], hi = pb[1];
            v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = g.b[(int64_t)(kb + j) * g.sbk + (int64_t)min(col, g.n - 1) * g.sbn];      // (B as stored or given transposed)
        }
        if (col >= g.n) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
    };
    f32x4 acc[CT];
    auto step = [&](int ch, const f32x4& a0, const f32x4& a1, bool a_ok) {
        u32x4 at[3];
        {
            uint32_t w[3];
            split_terms<TERMS>(a0[0], a0[1], w); at[0][0] = w[0]; at[1][0] = w[1]; at[2][0] = w[2];
            split_terms<TERMS>(a0[2], a0[3], w); at[0][1] = w[0]; at[1][1] = w[1]; at[2][1] = w[2];
            split_terms<TERMS>(a1[0], a1[1], w); at[0][2] = w[0]; at[1][2] = w[1]; at[2][2] = w[2];
            split_terms<TERMS>(a1[2], a1[3], w); at[0][3] = w[0]; at[1][3] = w[1]; at[2][3] = w[2];
        }
        if (!a_ok) { at[0] = (u32x4){0u, 0u, 0u, 0u}; at[1] = at[0]; at[2] = at[0]; }
        const bf16x8 xh = __builtin_bit_cast(bf16x8, at[0]), xm = __builtin_bit_cast(bf16x8, at[1]), xl = __builtin_bit_cast(bf16x8, at[2]);
        const u32x4* __restrict__ bp = bsplit + (size_t)ch * CT * TERMS * 64 + lane;
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if (t >= n_tiles) break;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, bp[(TERMS * t) * 64]), bm = __builtin_bit_cast(bf16x8, bp[(TERMS * t + 1) * 64]);
            if constexpr (TERMS == 3) {                                     // smallest terms first
                const bf16x8 bl = __builtin_bit_cast(bf16x8, bp[(TERMS * t + 2) * 64]);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xl, bh, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bl, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, bm, acc[t], 0, 0, 0);
            }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xm, bh, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bm, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xh, bh, acc[t], 0, 0, 0);
        }
    };
    auto store_tile = [&](int t0) {
        const int row0 = t0 * 16;
        if (g.c_vec_ok) {
            // A lane holds four ROWS of one column; transposed inside its quad of lanes (two DPP exchanges) it holds four
            // COLUMNS of one row - one 16-byte store (and one 16-byte load of what is accumulated / added) per lane and tile
            // instead of four 4-byte ones: a 50,000 x 256 output is 51 MB, and as 4-byte pieces its stores were the longest part
            // of the launch.
            const int s4 = r & 3, row = row0 + 4 * q + s4;
#pragma unroll
            for (int t = 0; t < CT; ++t) {
                const int col = col0 + 16 * t + (r & ~3);
                const bool odd = (s4 & 1) != 0, hi = (s4 & 2) != 0;
                const float v0 = acc[t][0], v1 = acc[t][1], v2 = acc[t][2], v3 = acc[t][3];
                const float a0 = dpp_f<0xB1>(v1), a1 = dpp_f<0xB1>(v0), a2 = dpp_f<0xB1>(v3), a3 = dpp_f<0xB1>(v2);
                const float w0 = odd ? a0 : v0, w1 = odd ? v1 : a1, w2 = odd ? a2 : v2, w3 = odd ? v3 : a3;
                const float b0 = dpp_f<0x4E>(w2), b1 = dpp_f<0x4E>(w3), b2 = dpp_f<0x4E>(w0), b3 = dpp_f<0x4E>(w1);
                f32x4 v = {hi ? b0 : w0, hi ? b1 : w1, hi ? w2 : b2, hi ? w3 : b3};
                if (col >= g.n || row >= g.m) continue;
                if (g.bias) v += *reinterpret_cast<const f32x4*>(g.bias + col);
                if (g.out_bf16) {                                           // the table a bf16-storage layer gathers from, written here
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));   // (gn_cast_bf16's rounding: nearest even, once)
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                    if (g.relu) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                    const f32x2 lo2 = {v[0], v[1]}, hi2 = {v[2], v[3]};
                    const u32x2 pk = {__builtin_bit_cast(uint32_t, __builtin_convertvector(lo2, bf16x2)),
                                      __builtin_bit_cast(uint32_t, __builtin_convertvector(hi2, bf16x2))};
                    *reinterpret_cast<u32x2*>(reinterpret_cast<uint16_t*>(g.c) + (int64_t)row * g.ldc + col) = pk;
                    continue;
                }
                float* cp = g.c + (int64_t)row * g.ldc + col;
                if (g.accumulate) v += *reinterpret_cast<const f32x4*>(cp);
                if (g.addend) v += *reinterpret_cast<const f32x4*>(g.addend + (int64_t)row * g.ld_add + col);
                if (g.relu) v = (f32x4){fmaxf(v[0], 0.f), fmaxf(v[1], 0.f), fmaxf(v[2], 0.f), fmaxf(v[3], 0.f)};
                *reinterpret_cast<f32x4*>(cp) = v;
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const int col = col0 + 16 * t + r;
            if (col >= g.n) continue;
            const float bias = g.bias ? g.bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * q + i;
                if (row < g.m) {
                    float v = acc[t][i] + bias;
                    if (g.accumulate) v += g.c[(int64_t)row * g.ldc + col];
                    if (g.addend) v += g.addend[(int64_t)row * g.ld_add + col];
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.c[(int64_t)row * g.ldc + col] = v;
                }
            }
        }
    };

    if constexpr (CH > 0) {
        constexpr int NB = (CH * CT * 64 + kSplitThreads - 1) / kSplitThreads;   // B words of 8 k per thread
        float bv[NB][8];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = threadIdx.x + i * kSplitThreads;
            get_b(min(idx, CH * CT * 64 - 1), bv[i]);
        }
        f32x4 av[CH][2];
        if (tile < row_tiles) {                                                 // wave-uniform
            const float* __restrict__ ap = a_ptr(tile);
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                av[ch][0] = *reinterpret_cast<const f32x4*>(ap + 32 * ch);
                av[ch][1] = *reinterpret_cast<const f32x4*>(ap + 32 * ch + 4);
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = threadIdx.x + i * kSplitThreads;
            if (idx < CH * CT * 64) put_b(idx, bv[i]);
        }
        __syncthreads();
        for (; tile < row_tiles; tile += tile_step) {
            const bool a_ok = tile * 16 + r < g.m;
            const int next = tile + tile_step < row_tiles ? tile + tile_step : tile;   // (the last tile re-reads itself, unused)
            const float* __restrict__ np = a_ptr(next);
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ch = 0; ch < CH; ++ch) {
                step(ch, av[ch][0], av[ch][1], a_ok);
                if (next != tile) {
                    av[ch][0] = *reinterpret_cast<const f32x4*>(np + 32 * ch);
                    av[ch][1] = *reinterpret_cast<const f32x4*>(np + 32 * ch + 4);
                }
            }
            store_tile(tile);
        }
    } else {
        // any K (a multiple of 32): B goes through LDS in slabs of `slab` chunks (all of it when it fits), the accumulators
        // stay in registers across the slabs; the workgroup walks its rounds of row tiles together (barriers around a refill)
        const int rounds = (row_tiles + tile_step - 1) / tile_step;
        for (int round = 0; round < rounds; ++round, tile += tile_step) {
            const bool live = tile < row_tiles;                                 // wave-uniform
            const bool a_ok = tile * 16 + r < g.m;
            const float* __restrict__ ap = a_ptr(live ? tile : 0);
#pragma unroll
            for (int t = 0; t < CT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < chunks; s0 += slab) {
                const int s1 = min(chunks, s0 + slab);
                f32x4 a0 = (f32x4)(0.f), a1 = a0;
                if (live) { a0 = *reinterpret_cast<const f32x4*>(ap + 32 * s0); a1 = *reinterpret_cast<const f32x4*>(ap + 32 * s0 + 4); }
                if (slab < chunks || round == 0) {
                    if (round > 0 || s0 > 0) __syncthreads();                   // everyone is done with the slab in LDS
                    for (int idx = threadIdx.x; idx < (s1 - s0) * CT * 64; idx += kSplitThreads) {
                        float v[8];
                        get_b(idx + s0 * CT * 64, v);
                        put_b(idx, v);
                    }
                    __syncthreads();
                }
                if (live)
                    for (int ch = s0; ch < s1; ++ch) {
                        const int nx = (ch + 1 < s1 ? ch + 1 : ch) * 32;
                        const f32x4 n0 = *reinterpret_cast<const f32x4*>(ap + nx), n1 = *reinterpret_cast<const f32x4*>(ap + nx + 4);
                        step(ch - s0, a0, a1, a_ok);
                        a0 = n0; a1 = n1;
                    }
            }
            if (live) store_tile(tile);
        }
    }
}

template <int TERMS, int CT>
gn_status launch_split(const GemmArgs& g, int row_tiles, dim3 sgrid, size_t split_bytes, int slab, hipStream_t st) {
#define GN_SPLIT_CASE(CH)                                                                                              \
    {                                                                                                                  \
        gn_status ls = gn::allow_large_lds(reinterpret_cast<const void*>(k_gemm_split_lds<CH, TERMS, CT>), 160 * 1024); \
        if (ls != GN_OK) return ls;                                                                                    \
        k_gemm_split_lds<CH, TERMS, CT><<<sgrid, kSplitThreads, split_bytes, st>>>(g, row_tiles, slab);                \
    }                                                                                                                  \
    break
    switch (slab < g.k / 32 ? 0 : g.k / 32) {
        case 1: GN_SPLIT_CASE(1);
        case 2: GN_SPLIT_CASE(2);
        case 4: GN_SPLIT_CASE(4);
        case 8: if constexpr (CT == 4) { GN_SPLIT_CASE(8); } else { GN_SPLIT_CASE(0); }
        default: GN_SPLIT_CASE(0);
    }
#undef GN_SPLIT_CASE
    GN_LAUNCH_CHECK();
    return GN_OK;
}

__global__ void k_merge(float* __restrict__ dst, int64_t ld_dst, const float* __restrict__ src, int64_t ld_src,
                        const float* __restrict__ src2, int64_t ld_src2, int64_t rows, int cols, int mode) {
    const int64_t total = rows * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / cols;
        const int c = (int)(t - i * cols);
        const float s = src[i * ld_src + c];
        float* d = dst + i * ld_dst + c;
        switch (mode) {
            case 0: *d = s; break;
            case 1: *d = fabsf(s); break;
            case 2: *d = (*d + fabsf(s)) / 2.0f; break;
            case 3: *d = (*d + fmaxf(s, 0.f)) / 2.0f; break;
            case 4: *d = (*d + s + src2[i * ld_src2 + c]) / 3.0f; break;
            case 5: *d = src2[i * ld_src2 + c] > 0.f ? s : 0.f; break;
            case 6: {
                const float r = src2[i * ld_src2 + c];
                *d = r > 0.f ? s : (r < 0.f ? -s : 0.f);
            } break;
            case 7: *d = s / 3.0f; break;
            case 8: *d = s / 2.0f; break;
            case 9: {
                const float r = src2[i * ld_src2 + c];
                *d = r > 0.f ? s / 2.0f : (r < 0.f ? -s / 2.0f : 0.f);
            } break;
            default: *d = src2[i * ld_src2 + c] > 0.f ? s / 2.0f : 0.f; break;
        }
    }
}

// One wave per row: max, exp, sum, divide (decoder.py:43).
__global__ void k_softmax_rows(float* __restrict__ x, int64_t ld, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < rows; i += n_waves) {
        float* row = x + i * ld;
        float mx = -INFINITY;
        for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, row[c]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int c = lane; c < cols; c += 64) sum += expf(row[c] - mx);
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        for (int c = lane; c < cols; c += 64) row[c] = expf(row[c] - mx) / sum;
    }
}

// Gradient of the row softmax (autograd of decoder.py:43): dx = p (g - sum_c g_c p_c), one wave per row.
__global__ void k_softmax_rows_backward(const float* __restrict__ p, int64_t ld_p, const float* __restrict__ g, int64_t ld_g,
                                        float* __restrict__ dx, int64_t ld_dx, int64_t rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    for (int64_t i = wave; i < rows; i += n_waves) {
        const float* pr = p + i * ld_p;
        const float* gr = g + i * ld_g;
        float dot = 0.f;
        for (int c = lane; c < cols; c += 64) dot += pr[c] * gr[c];
        for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        for (int c = lane; c < cols; c += 64) dx[i * ld_dx + c] = pr[c] * (gr[c] - dot);
    }
}

// softmax?( z[node_list] @ W ) for a handful of classes (multiClassInnerProductDecoder, decoder.py:42-43): sixteen lanes
// per selected row, four rows per wave, one row per lane group and launch-wide no loop on the NC shapes.  W (k x n,
// n <= 16) sits in LDS as [n / 4][k] 16-byte words (the four lane groups read the same words: broadcasts); a lane sums
// its columns j, j + 16, ... of the row, eight loads in flight; the n sums are folded over the sixteen lanes with DPP row
// operations (no LDS traffic), lane c of the group then owns class c: max, exp, sum, divide once per class.  One launch
// and one pass over the selected rows instead of a row-gather GEMM whose 16 x 16 tiles are mostly padding plus a softmax pass.
constexpr int kClassMax = 16;

template <int CTRL>
__device__ __forceinline__ float row_dpp(float x) {            // all 16 lanes of a row active
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row16_sum(float x) {          // quad xor 1, quad xor 2, half-row mirror, row mirror
    x += row_dpp<0xB1>(x); x += row_dpp<0x4E>(x); x += row_dpp<0x141>(x); x += row_dpp<0x140>(x);
    return x;
}
__device__ __forceinline__ float row16_max(float x) {
    x = fmaxf(x, row_dpp<0xB1>(x)); x = fmaxf(x, row_dpp<0x4E>(x)); x = fmaxf(x, row_dpp<0x141>(x)); x = fmaxf(x, row_dpp<0x140>(x));
    return x;
}

__global__ __launch_bounds__(1024) void k_class_scores(const float* __restrict__ z, int64_t ld_z, int64_t table_rows,
                                                       const int64_t* __restrict__ nodes, int64_t m, const float* __restrict__ w,
                                                       int64_t ld_w, int k, int n, int softmax, float* __restrict__ out, int64_t ld_out) {
    extern __shared__ f32x4 wl4[];                             // [kClassMax / 4][k]
    const int n4 = (n + 3) >> 2;
    const int j = threadIdx.x & 15;
    const int64_t group = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4;
    const int64_t n_groups = ((int64_t)gridDim.x * blockDim.x) >> 4;
    // the group's first node id is on its way while W is laid out
    int64_t src_next = group < m ? (nodes ? nodes[group] : group) : 0;
    for (int i = threadIdx.x; i < k * n4; i += 1024) {
        const int c4 = i / k, kk = i - c4 * k;
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) v[t] = 4 * c4 + t < n ? w[(int64_t)kk * ld_w + 4 * c4 + t] : 0.f;
        wl4[i] = v;
    }
    __syncthreads();
    const int64_t trips = (m + n_groups - 1) / n_groups;       // the wave walks together (DPP rows need every lane)
    int64_t i = group;
    for (int64_t trip = 0; trip < trips; ++trip, i += n_groups) {
        const bool live = i < m;
        int64_t src = src_next;
        if (i + n_groups < m) src_next = nodes ? nodes[i + n_groups] : i + n_groups;
        const bool ok = live && (uint64_t)src < (uint64_t)table_rows;   // out of the table -> a row of zeros, as the row-gather GEMM
        if (!ok) src = 0;
        const float* __restrict__ row = z + src * ld_z;
        float acc[kClassMax];
#pragma unroll
        for (int c = 0; c < kClassMax; ++c) acc[c] = 0.f;
        for (int k0 = 0; k0 < k; k0 += 8 * 16) {                // eight loads of the row in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kk = k0 + 16 * u + j;
                v[u] = (ok && kk < k) ? row[kk] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + 16 * u >= k) break;
                const int kk = min(k0 + 16 * u + j, k - 1);     // (beyond k: v is zero)
#pragma unroll
                for (int c4 = 0; c4 < kClassMax / 4; ++c4) {
                    if (c4 >= n4) break;
                    const f32x4 ww = wl4[c4 * k + kk];
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[4 * c4 + t] += v[u] * ww[t];
                }
            }
        }
        float mine = 0.f;                                      // class j of the group's row
#pragma unroll
        for (int c = 0; c < kClassMax; ++c) {
            if (c >= n) break;
            const float s = row16_sum(acc[c]);
            if (c == j) mine = s;
        }
        if (softmax) {
            const float mx = row16_max(j < n ? mine : -INFINITY);
            const float e = j < n ? expf(mine - mx) : 0.f;
            mine = e / row16_sum(e);
        }
        if (live && j < n) out[i * ld_out + j] = mine;
    }
}

}  // namespace


// ---- out[k1, k2] = x^T g over m rows (weight gradients of the GCN-style layers: dW = x^T (A^T g)) -----------------
// Tall and skinny: m is the gene supervertex (~2e4 rows), k1 x k2 a weight matrix (<= 64 x 32).  Every workgroup
// takes a slice of rows into LDS and leaves its k1 x k2 partial; k_xtg_fold adds the slices in slice order
// (bitwise reproducible; a library GEMM without split-K spends 60-100 us on these shapes).
constexpr int kXtgSlices = 256;
constexpr int kXtgMaxRows = 128;       // rows of a slice held in LDS at a time

__global__ __launch_bounds__(256) void k_xtg_partial(const float* __restrict__ x, int64_t ld_x, const float* __restrict__ g,
                                                     int64_t ld_g, int64_t m, int k1, int k2, float* __restrict__ partial) {
    extern __shared__ float xtg_lds[];
    float* xs = xtg_lds;                                  // [rows][k1 + 1]
    float* gs = xtg_lds + (size_t)kXtgMaxRows * (k1 + 1); // [rows][k2]
    const int tid = threadIdx.x;
    const int64_t per = (m + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = blockIdx.x * per, r1 = min(m, r0 + per);
    const int outs = k1 * k2;
    float acc[16];                                        // outputs tid, tid + 256, ... (k1 k2 <= 4096)
#pragma unroll
    for (int o = 0; o < 16; ++o) acc[o] = 0.f;
    for (int64_t base = r0; base < r1; base += kXtgMaxRows) {
        const int rows = (int)min<int64_t>(kXtgMaxRows, r1 - base);
        __syncthreads();
        for (int i = tid; i < rows * k1; i += 256) {
            const int r = i / k1, c = i - r * k1;
            xs[r * (k1 + 1) + c] = x[(base + r) * ld_x + c];
        }
        for (int i = tid; i < rows * k2; i += 256) {
            const int r = i / k2, c = i - r * k2;
            gs[r * k2 + c] = g[(base + r) * ld_g + c];
        }
        __syncthreads();
#pragma unroll
        for (int o = 0; o < 16; ++o) {
            const int idx = tid + o * 256;
            if (idx >= outs) break;
            const int i = idx / k2, j = idx - i * k2;
            float a = acc[o];
            for (int r = 0; r < rows; ++r) a = fmaf(xs[r * (k1 + 1) + i], gs[r * k2 + j], a);
            acc[o] = a;
        }
    }
#pragma unroll
    for (int o = 0; o < 16; ++o) {
        const int idx = tid + o * 256;
        if (idx < outs) partial[(size_t)blockIdx.x * outs + idx] = acc[o];
    }
}

// The same product on the matrix cores in ONE launch, for up to 64 x 32 outputs (every weight gradient of the PoSE model):
// out = x^T g is a GEMM whose K is the row count.  A workgroup takes a slice of the 16-row chunks, its sixteen waves take
// them round-robin (A operand = x read down its columns, B = g; loads of the next chunk in flight), the waves' accumulators
// meet in LDS and are added in wave order, the slice's sums are stored write-through, and the last slice to arrive adds all
// slices in slice order (no fence: MI355X_MICROARCH.md's form for a few KB).  Deterministic; fp32 MFMA = fp32 FMA chains.
constexpr int kXtgMfmaSlices = 32;       // slices the last one to arrive adds alone
constexpr int kXtgMfmaMax = 128;         // slices of a long product (two levels: sets of kXtgSet), <= kXtgSlices - kXtgMfmaMax / kXtgSet
constexpr int kXtgSet = 16;

struct XtgArgs {
    const float* x; int64_t ld_x; const float* g; int64_t ld_g; int64_t m; int k1, k2;
    float* partial; unsigned int* ticket; float* out; int64_t ld_out;
};

template <int MT, int NT>
__device__ __forceinline__ void xtg_mfma_body(const XtgArgs& a, int bx, int nblocks, f32x4* __restrict__ xtg_part, int* last_flag) {
    const float* __restrict__ x = a.x;
    const float* __restrict__ g = a.g;
    const int64_t ld_x = a.ld_x, ld_g = a.ld_g, m = a.m, ld_out = a.ld_out;
    const int k1 = a.k1, k2 = a.k2;
    float* __restrict__ partial = a.partial;
    unsigned int* __restrict__ ticket = a.ticket;
    float* __restrict__ out = a.out;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int chunks = (int)((m + 15) / 16), per = (chunks + nblocks - 1) / nblocks;
    const int c0 = bx * per, c1 = min(chunks, c0 + per);
    f32x4 acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](int ch, float (&av)[MT][4], float (&bv)[NT][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row = min<int64_t>(16 * (int64_t)ch + 4 * q + j, m - 1);
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t][j] = x[row * ld_x + min(16 * t + r, k1 - 1)];
#pragma unroll
            for (int u = 0; u < NT; ++u) bv[u][j] = g[row * ld_g + min(16 * u + r, k2 - 1)];
        }
    };
    float av[MT][4], bv[NT][4];
    if (c0 + wave < c1) load_chunk(c0 + wave, av, bv);
    for (int ch = c0 + wave; ch < c1; ch += 16) {
        float an[MT][4], bn[NT][4];
        load_chunk(ch + 16 < c1 ? ch + 16 : ch, an, bn);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool row_ok = 16 * (int64_t)ch + 4 * q + j < m;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                const float a = (row_ok && 16 * t + r < k1) ? av[t][j] : 0.f;
#pragma unroll
                for (int u = 0; u < NT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, (16 * u + r < k2) ? bv[u][j] : 0.f, acc[t][u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < MT; ++t) av[t][j] = an[t][j];
#pragma unroll
            for (int u = 0; u < NT; ++u) bv[u][j] = bn[u][j];
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int u = 0; u < NT; ++u) xtg_part[(wave * MT * NT + t * NT + u) * 64 + lane] = acc[t][u];
    __syncthreads();
    const bool single = nblocks == 1;
    const int outs = k1 * k2;
    for (int o = tid; o < MT * NT * 256; o += 1024) {
        const int tile = o >> 8, l = (o & 255) >> 2, i = o & 3;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += xtg_part[(w * MT * NT + tile) * 64 + l][i];
        const int row = 16 * (tile / NT) + 4 * (l >> 4) + i, col = 16 * (tile % NT) + (l & 15);
        if (row < k1 && col < k2) {
            if (single) out[(int64_t)row * ld_out + col] = v;
            else __hip_atomic_store(reinterpret_cast<unsigned int*>(partial) + (size_t)bx * outs + row * k2 + col, __float_as_uint(v),
                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (single) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's write-through stores have reached L2 ...
    __syncthreads();                                                       // ... and so have every other wave's when the ticket is drawn
    const unsigned int* __restrict__ parts = reinterpret_cast<const unsigned int*>(partial);
    if (nblocks > kXtgMfmaSlices) {
        // More than 32 slices (the 50,000-row layers of the node-classification models: 32 workgroups were an eighth of the
        // chip): two levels of the same hand-over - the slices in sets of kXtgSet, the last of a set adds its set in slice
        // order and leaves the sum behind the slices (rows kXtgMfmaMax.. of `partial`), the last set to finish adds the sets.
        const int set = bx / kXtgSet, n_sets = (nblocks + kXtgSet - 1) / kXtgSet, set_n = min(kXtgSet, nblocks - set * kXtgSet);
        if (tid == 0) *last_flag = __hip_atomic_fetch_add(ticket + 1 + set, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(set_n - 1);
        __syncthreads();
        if (!*last_flag) return;
        for (int o = tid; o < outs; o += 1024) {
            float p[kXtgSet];
#pragma unroll
            for (int sl = 0; sl < kXtgSet; ++sl)
                p[sl] = __uint_as_float(__hip_atomic_load(parts + (size_t)(set * kXtgSet + min(sl, set_n - 1)) * outs + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            float v = 0.f;
#pragma unroll
            for (int sl = 0; sl < kXtgSet; ++sl) v += sl < set_n ? p[sl] : 0.f;
            __hip_atomic_store(reinterpret_cast<unsigned int*>(partial) + (size_t)(kXtgMfmaMax + set) * outs + o, __float_as_uint(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_store(ticket + 1 + set, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *last_flag = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(n_sets - 1);
        }
        __syncthreads();
        if (!*last_flag) return;
        constexpr int kSetsMax = kXtgMfmaMax / kXtgSet;
        for (int o = tid; o < outs; o += 1024) {
            float p[kSetsMax];
#pragma unroll
            for (int sl = 0; sl < kSetsMax; ++sl)
                p[sl] = __uint_as_float(__hip_atomic_load(parts + (size_t)(kXtgMfmaMax + min(sl, n_sets - 1)) * outs + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            float v = 0.f;
#pragma unroll
            for (int sl = 0; sl < kSetsMax; ++sl) v += sl < n_sets ? p[sl] : 0.f;
            out[(int64_t)(o / k2) * ld_out + o % k2] = v;
        }
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (tid == 0) *last_flag = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nblocks - 1);
    __syncthreads();
    if (!*last_flag) return;
    for (int o = tid; o < outs; o += 1024) {
        // all slices requested before any is added (a load - add chain pays a round trip per slice), added in slice order
        float p[kXtgMfmaSlices];
#pragma unroll
        for (int sl = 0; sl < kXtgMfmaSlices; ++sl)
            p[sl] = __uint_as_float(__hip_atomic_load(parts + (size_t)min(sl, nblocks - 1) * outs + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float v = 0.f;
#pragma unroll
        for (int sl = 0; sl < kXtgMfmaSlices; ++sl) v += sl < nblocks ? p[sl] : 0.f;
        out[(int64_t)(o / k2) * ld_out + o % k2] = v;
    }
    if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
}

// The same product for WIDE outputs (the 128 x 64 ... 256 x 128 weight gradients of the node-classification layers, over
// 20,000-50,000 rows).  In tiles of 64 x 32 outputs, one launch each, x and g were read once per tile (307 MB for the 77 MB
// of a 256 x 128 product) and every tile paid its own fill, LDS reduction and hand-over.  Here a workgroup takes a slice of
// the rows and ALL TI x TJ tiles: a wave per tile (or 16 / tiles waves per tile, which then split the slice's chunks and add
// up through LDS in wave order) walks the slice's 16-row chunks with the A operand x[:, 64 ti ..] and the B operand g[:, 32 tj ..]
// - the waves of a workgroup read the same rows, the L1 serves the repeats - and leaves the slice's sums in `partial`;
// k_xtg_fold adds the slices in slice order behind it.  fp32 matrix instruction: the sums are fp32 FMA chains; bitwise reproducible.
template <int TI, int TJ>
__global__ __launch_bounds__(1024) void k_xtg_wide(XtgArgs a) {
    extern __shared__ f32x4 wide_part[];                                 // [16 waves][8][64] (when several waves share a tile)
    constexpr int TILES = TI * TJ, WPT = 16 / TILES;
    static_assert(TILES >= 2 && 16 % TILES == 0, "2, 4, 8 or 16 tiles");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const int tile = wave / WPT, sub = wave % WPT, ti = tile / TJ, tj = tile % TJ;
    const float* __restrict__ x = a.x + 64 * ti;
    const float* __restrict__ g = a.g + 32 * tj;
    const int64_t ld_x = a.ld_x, ld_g = a.ld_g, m = a.m;
    const int k2 = a.k2;
    const int chunks = (int)((m + 15) / 16), per = (chunks + (int)gridDim.x - 1) / (int)gridDim.x;
    const int c0 = (int)blockIdx.x * per, c1 = min(chunks, c0 + per);
    f32x4 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u) acc[t][u] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto load_chunk = [&](int ch, float (&av)[4][4], float (&bv)[2][4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t row = min<int64_t>(16 * (int64_t)ch + 4 * q + j, m - 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) av[t][j] = x[row * ld_x + 16 * t + r];
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[u][j] = g[row * ld_g + 16 * u + r];
        }
    };
    float av[4][4], bv[2][4];
    if (c0 + sub < c1) load_chunk(c0 + sub, av, bv);
    for (int ch = c0 + sub; ch < c1; ch += WPT) {
        float an[4][4], bn[2][4];
        load_chunk(ch + WPT < c1 ? ch + WPT : ch, an, bn);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool row_ok = 16 * (int64_t)ch + 4 * q + j < m;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float xa = row_ok ? av[t][j] : 0.f;
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa, bv[u][j], acc[t][u], 0, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int t = 0; t < 4; ++t) av[t][j] = an[t][j];
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[u][j] = bn[u][j];
        }
    }
    // element i of lane l of tile (t, u): row 16 t + 4 (l >> 4) + i, column 16 u + (l & 15) of the wave's 64 x 32 outputs
    float* __restrict__ mine = a.partial + (size_t)blockIdx.x * a.k1 * k2;
    if constexpr (WPT == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    mine[(size_t)(64 * ti + 16 * t + 4 * q + i) * k2 + 32 * tj + 16 * u + r] = acc[t][u][i];
    } else {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) wide_part[(wave * 8 + t * 2 + u) * 64 + lane] = acc[t][u];
        __syncthreads();
        for (int o = tid; o < TILES * 8 * 256; o += 1024) {
            const int tl = o / (8 * 256), rem = o - tl * (8 * 256), tu = rem >> 8, l = (rem & 255) >> 2, i = rem & 3;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < WPT; ++w) v += wide_part[((tl * WPT + w) * 8 + tu) * 64 + l][i];
            const int row = 64 * (tl / TJ) + 16 * (tu >> 1) + 4 * (l >> 4) + i, col = 32 * (tl % TJ) + 16 * (tu & 1) + (l & 15);
            mine[(size_t)row * k2 + col] = v;
        }
    }
}

template <int MT, int NT>
__global__ __launch_bounds__(1024) void k_xtg_mfma(XtgArgs a) {
    extern __shared__ f32x4 xtg_part[];                                  // [16 waves][MT * NT][64]
    __shared__ int last;
    xtg_mfma_body<MT, NT>(a, blockIdx.x, gridDim.x, xtg_part, &last);
}

int xtg_slices(int64_t m);
static_assert(kXtgMfmaMax + kXtgMfmaMax / kXtgSet <= kXtgSlices && (1 + kXtgMfmaMax / kXtgSet) * 4 <= 64, "the sets' sums and tickets live in the workspace of gn_xtg_workspace_bytes");

template <int MT, int NT>
gn_status launch_xtg_mfma(const float* x, int64_t ld_x, const float* g, int64_t ld_g, int64_t m, int k1, int k2, float* out, int64_t ld_out,
                          void* workspace, hipStream_t st) {
    const size_t lds = (size_t)16 * MT * NT * 64 * sizeof(f32x4);
    if (lds > 64 * 1024) {
        const gn_status ls = gn::allow_large_lds(reinterpret_cast<const void*>(k_xtg_mfma<MT, NT>), 136 * 1024);   // (+ 4 static bytes)
        if (ls != GN_OK) return ls;
    }
    // about two chunks per wave; the ticket sits behind the slices' sums
    const int slices = xtg_slices(m);
    float* partial = static_cast<float*>(workspace);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(static_cast<char*>(workspace) + (size_t)kXtgSlices * k1 * k2 * sizeof(float));
    const XtgArgs a = {x, ld_x, g, ld_g, m, k1, k2, partial, ticket, out, ld_out};
    k_xtg_mfma<MT, NT><<<slices, 1024, lds, st>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// ---- independent small products in ONE launch ------------------------------------------------------------------------
// The weight gradients of a layer's backward are a handful of products of a few dozen workgroups and ~10 us each (latency
// chains of a few round trips), independent of each other: dbasis = att^T dW, datt = dW basis^T and droot = x^T g of the
// relational layer.  Between gn_dense_batch_begin() and gn_dense_batch_end(stream) the calls of gn_gemm_f32 / gn_xtg_f32 that
// carry GN_GEMM_JOIN_BATCH / GN_XTG_JOIN_BATCH and take the deep-and-narrow, the tall-skinny fp32 or the one-launch x^T g kernel
// are QUEUED (per host thread) and leave as one grid whose workgroups pick their product from a table in the launch's
// arguments; every other call inside the bracket - the library's own products inside other entry points among them -
// launches as usual.  The caller promises that the queued products do not depend on each other.
constexpr int kBatchMax = 4;
struct BatchOp {
    int kind;                        // 0: gemm_deep_body, 1: xtg_mfma_body, 2: gemm_lds_body
    int mt, nt;                      // the body's tile template (kind 2: mt = row tiles)
    int lds;                         // dynamic LDS bytes of the body
    int first, blocks, gx;           // workgroups [first, first + blocks) of the launch; deep: grid (gx, blocks / gx)
    GemmArgs g;
    XtgArgs x;
};
struct BatchTable { BatchOp op[kBatchMax]; int n; };

__global__ __launch_bounds__(1024) void k_dense_batch(BatchTable tab) {
    extern __shared__ f32x4 batch_lds[];
    __shared__ int last;
    int k = 0;
    while (k + 1 < tab.n && (int)blockIdx.x >= tab.op[k + 1].first) ++k;   // (uniform: scalar registers)
    const BatchOp& op = tab.op[k];
    const int vb = (int)blockIdx.x - op.first;
    if (op.kind == 2) {                                          // tall-skinny, B in LDS: sixteen waves per workgroup here
        gemm_lds_body(op.g, op.mt, vb % op.gx, vb / op.gx, op.gx, 16, batch_lds);
        return;
    }
    if (op.kind == 0) {
        const int bx = vb % op.gx, by = vb / op.gx;
        if (op.mt == 1 && op.nt == 1) gemm_deep_body<1, 1>(op.g, bx, by, batch_lds);
        else if (op.mt == 2 && op.nt == 1) gemm_deep_body<2, 1>(op.g, bx, by, batch_lds);
        else if (op.mt == 4 && op.nt == 1) gemm_deep_body<4, 1>(op.g, bx, by, batch_lds);
        else gemm_deep_body<1, 2>(op.g, bx, by, batch_lds);
        return;
    }
#define GN_XTG_BODY(MT, NT) if (op.mt == MT && op.nt == NT) { xtg_mfma_body<MT, NT>(op.x, vb, op.blocks, batch_lds, &last); return; }
    GN_XTG_BODY(1, 1) GN_XTG_BODY(2, 1) GN_XTG_BODY(3, 1) GN_XTG_BODY(4, 1)
    GN_XTG_BODY(1, 2) GN_XTG_BODY(2, 2) GN_XTG_BODY(3, 2) GN_XTG_BODY(4, 2)
#undef GN_XTG_BODY
}

thread_local bool batch_open = false;
thread_local std::vector<BatchOp> batch_queue;

void deep_shape(const GemmArgs& g, BatchOp& op) {
    if (g.m <= 64) {
        op.mt = g.m > 32 ? 4 : g.m > 16 ? 2 : 1; op.nt = 1;
        op.gx = 1; op.blocks = (int)gn::ceil_div(g.n, 16);
    } else {
        op.mt = 1; op.nt = g.n > 16 ? 2 : 1;
        op.gx = (int)gn::ceil_div(g.m, 16); op.blocks = op.gx;
    }
}

gn_status launch_deep(const GemmArgs& g, hipStream_t st) {
    BatchOp op;
    deep_shape(g, op);
    dim3 grid((unsigned)op.gx, (unsigned)(op.blocks / op.gx), 1);
    if (op.mt == 4) k_gemm_deep<4, 1><<<grid, kDeepWaves * 64, 0, st>>>(g);
    else if (op.mt == 2) k_gemm_deep<2, 1><<<grid, kDeepWaves * 64, 0, st>>>(g);
    else if (op.nt == 2) k_gemm_deep<1, 2><<<grid, kDeepWaves * 64, 0, st>>>(g);
    else k_gemm_deep<1, 1><<<grid, kDeepWaves * 64, 0, st>>>(g);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status launch_lds(const GemmArgs& g, hipStream_t st) {
    const size_t lds_bytes = (size_t)gn::ceil_div(g.k, 16) * kColTiles * 64 * sizeof(f32x4);
    const int row_tiles = (int)gn::ceil_div(g.m, 16);
    dim3 lgrid((unsigned)std::min<int64_t>(gn::ceil_div(row_tiles, 4), 1024), (unsigned)gn::ceil_div(g.n, 16 * kColTiles), 1);
    k_gemm_f32_lds<<<lgrid, 256, lds_bytes, st>>>(g, row_tiles);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

XtgArgs xtg_args(const float* x, int64_t ld_x, const float* g, int64_t ld_g, int64_t m, int k1, int k2, float* out, int64_t ld_out, void* workspace) {
    float* partial = static_cast<float*>(workspace);
    unsigned int* ticket = reinterpret_cast<unsigned int*>(static_cast<char*>(workspace) + (size_t)kXtgSlices * k1 * k2 * sizeof(float));
    return XtgArgs{x, ld_x, g, ld_g, m, k1, k2, partial, ticket, out, ld_out};
}
int xtg_slices(int64_t m) {      // about two chunks of sixteen rows per wave; up to 40 of them are 32 (one level), more are up to 128 (two)
    const int64_t want = gn::ceil_div(m, 16 * 16 * 2);
    return (int)std::max<int64_t>(1, want <= 40 ? std::min<int64_t>(kXtgMfmaSlices, want) : std::min<int64_t>(kXtgMfmaMax, want));
}

gn_status launch_xtg_op(const BatchOp& op, hipStream_t st) {
    const XtgArgs& a = op.x;
    void* ws = a.partial;
#define GN_XTG_CASE(MT, NT) if (op.mt == MT && op.nt == NT) return launch_xtg_mfma<MT, NT>(a.x, a.ld_x, a.g, a.ld_g, a.m, a.k1, a.k2, a.out, a.ld_out, ws, st)
    GN_XTG_CASE(1, 1); GN_XTG_CASE(2, 1); GN_XTG_CASE(3, 1); GN_XTG_CASE(4, 1);
    GN_XTG_CASE(1, 2); GN_XTG_CASE(2, 2); GN_XTG_CASE(3, 2); GN_XTG_CASE(4, 2);
#undef GN_XTG_CASE
    return gn::fail(GN_ERR_INVALID_ARG, "x^T g tile shape %d x %d", op.mt, op.nt);
}

gn_status flush_batch(hipStream_t st) {
    std::vector<BatchOp> ops;
    ops.swap(batch_queue);
    for (size_t done = 0; done < ops.size();) {
        const size_t take = std::min<size_t>(kBatchMax, ops.size() - done);
        if (take == 1) {                                      // alone: its own kernel
            const BatchOp& op = ops[done];
            const gn_status rc = op.kind == 0 ? launch_deep(op.g, st) : op.kind == 1 ? launch_xtg_op(op, st) : launch_lds(op.g, st);
            if (rc != GN_OK) return rc;
            done += 1;
            continue;
        }
        BatchTable tab;
        tab.n = (int)take;
        int blocks = 0;
        size_t lds = 0;
        for (size_t i = 0; i < take; ++i) {
            tab.op[i] = ops[done + i];
            tab.op[i].first = blocks;
            blocks += tab.op[i].blocks;
            lds = std::max(lds, (size_t)tab.op[i].lds);
        }
        if (lds > 64 * 1024) {
            const gn_status ls = gn::allow_large_lds(reinterpret_cast<const void*>(k_dense_batch), 136 * 1024);   // (+ 4 static bytes)
            if (ls != GN_OK) return ls;
        }
        k_dense_batch<<<blocks, 1024, lds, st>>>(tab);
        GN_LAUNCH_CHECK();
        done += take;
    }
    return GN_OK;
}

// 16 outputs per workgroup: 16 groups of threads add 16 slices each (independent loads), then the groups are added
// in group order.
__global__ __launch_bounds__(256) void k_xtg_fold(const float* __restrict__ partial, int slices, int outs, int k2,
                                                  float* __restrict__ out, int64_t ld_out) {
    __shared__ float fold[16][17];
    const int o = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int idx = blockIdx.x * 16 + o;
    float v[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int b = grp * 16 + k;
        v[k] = (b < slices && idx < outs) ? partial[(size_t)b * outs + idx] : 0.f;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += v[k];
    fold[grp][o] = s;
    __syncthreads();
    if (grp == 0 && idx < outs) {
        float r = fold[0][o];
#pragma unroll
        for (int k = 1; k < 16; ++k) r += fold[k][o];
        out[(int64_t)(idx / k2) * ld_out + idx % k2] = r;
    }
}

extern "C" {

gn_status gn_gemm_f32(const float* a, int64_t lda, int64_t stride_a, const int64_t* a_rows, int64_t a_table_rows,
                      const float* b, int64_t ldb, int64_t stride_b, float* c, int64_t ldc, int64_t stride_c,
                      int64_t m, int64_t n, int64_t k, int64_t batch, const float* bias, int flags, void* stream) {
    return gn_gemm_addend_f32(a, lda, stride_a, a_rows, a_table_rows, b, ldb, stride_b, c, ldc, stride_c, m, n, k, batch, bias, nullptr, 0, flags, stream);
}

gn_status gn_gemm_addend_f32(const float* a, int64_t lda, int64_t stride_a, const int64_t* a_rows, int64_t a_table_rows,
                             const float* b, int64_t ldb, int64_t stride_b, float* c, int64_t ldc, int64_t stride_c,
                             int64_t m, int64_t n, int64_t k, int64_t batch, const float* bias, const float* addend, int64_t ld_addend,
                             int flags, void* stream) {
    GN_REQUIRE(!addend || (batch == 1 && ld_addend >= n), "an addend goes with a single product and has rows of at least n floats");
    const int relu = flags & GN_GEMM_RELU, fast = flags & GN_GEMM_ARITH_FAST;
    const bool bt = (flags & GN_GEMM_B_TRANSPOSED) != 0, accumulate = (flags & GN_GEMM_ACCUMULATE) != 0, at = (flags & GN_GEMM_A_TRANSPOSED) != 0;
    GN_REQUIRE(!at || !a_rows, "a row gather of a transposed A is not supported");
    GN_REQUIRE(m >= 0 && n >= 0 && k >= 0 && batch >= 0, "negative GEMM size");
    if (m == 0 || n == 0 || batch == 0) return GN_OK;
    GN_REQUIRE(a && b && c, "GEMM operand pointer is null");
    GN_REQUIRE(lda >= (at ? m : k) && ldb >= (bt ? k : n) && ldc >= n, "leading dimension smaller than the row length");
    GN_REQUIRE(m < (1ll << 31) && n < (1ll << 31) && k < (1ll << 31) && batch <= 65535, "GEMM size out of range");
    GemmArgs g;
    g.a = a; g.lda = lda; g.stride_a = stride_a; g.a_rows = a_rows; g.a_table_rows = a_table_rows;
    g.b = b; g.ldb = ldb; g.stride_b = stride_b;
    g.c = c; g.ldc = ldc; g.stride_c = stride_c;
    g.m = (int)m; g.n = (int)n; g.k = (int)k; g.bias = bias; g.relu = relu;
    g.a_vec_ok = ((reinterpret_cast<uintptr_t>(a) & 15) == 0) && (lda % 4 == 0) && (stride_a % 4 == 0);
    g.sbk = bt ? 1 : ldb; g.sbn = bt ? ldb : 1; g.accumulate = accumulate ? 1 : 0;
    g.addend = addend; g.ld_add = ld_addend;
    g.c_vec_ok = (n % 4 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(c) & 15) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0) &&
                 (!addend || ((ld_addend % 4 == 0) && (reinterpret_cast<uintptr_t>(addend) & 15) == 0)) ? 1 : 0;
    g.sam = at ? 1 : lda; g.sak = at ? lda : 1;
    if (flags & GN_GEMM_OUT_BF16) {
        // c is a bf16 table: only the tall-skinny split kernel stores it (what a bf16-storage layer of the node-classification
        // models needs); anything else is GN_ERR_UNSUPPORTED and the caller rounds a fp32 product with gn_cast_bf16
        const bool split_path = batch == 1 && m >= 2048 && !a_rows && k >= 32 && k % 32 == 0 && g.a_vec_ok && !gn::fast_paths_disabled();
        const bool vec = (n % 4 == 0) && (ldc % 4 == 0) && ((reinterpret_cast<uintptr_t>(c) & 7) == 0) && (!bias || (reinterpret_cast<uintptr_t>(bias) & 15) == 0);
        if (!split_path || !vec || accumulate || addend || at)
            return gn::fail(GN_ERR_UNSUPPORTED, "GN_GEMM_OUT_BF16: a tall-skinny product (m >= 2048, k %% 32 == 0) with n %% 4 == 0, 8-byte aligned rows, no accumulate / addend");
        g.out_bf16 = 1; g.c_vec_ok = 1;
    }
    // (a TALL product with at most 32 columns - the general relational path's slab of rows times [basis ; root], 19,726 x 1,088 x 32 -
    // is streamed by the tall-skinny split kernel below, not cut into K slices here: 43 -> 2x us for its 86 MB, round 6)
    const bool want_split = (flags & GN_GEMM_SPLIT_KERNEL) != 0;   // (the caller's products must not change kernel - and bits - with their row count)
    const bool tall_split = !at && (m >= 2048 || want_split) && k >= 32 && k % 32 == 0 && g.a_vec_ok;
    if (batch == 1 && !a_rows && (m <= 64 || n <= 32) && !tall_split && (at || (k >= 256 && !gn::fast_paths_disabled()))) {
        // deep and narrow (and every product with A given transposed): a workgroup per output tile, K over its waves
        if (batch_open && (flags & GN_GEMM_JOIN_BATCH)) {         // between gn_dense_batch_begin / _end: leaves with the others
            BatchOp op;
            op.kind = 0; op.g = g; op.first = 0;
            deep_shape(g, op);
            op.lds = (int)((size_t)16 * op.mt * op.nt * 64 * sizeof(f32x4));
            batch_queue.push_back(op);
            return GN_OK;
        }
        return launch_deep(g, gn::as_stream(stream));
    }
    GN_REQUIRE(!at, "A given transposed: at most 64 rows or 32 columns of output (the deep and narrow kernel)");
    const size_t lds_bytes = (size_t)gn::ceil_div(k, 16) * kColTiles * 64 * sizeof(f32x4);
    const int row_tiles = (int)gn::ceil_div(m, 16);
    if (batch == 1 && (m >= 2048 || want_split) && !a_rows && k >= 32 && k % 32 == 0 && g.a_vec_ok && !gn::fast_paths_disabled()) {
        // tall-skinny, one shared B (as stored, or given transposed: the dx = g W^T of the wide layers' backward): the bf16 matrix
        // instruction on split operands.  (Not queued in a dense batch: on 50,000 x 128 x 128 it takes a third of the fp32 instruction's time.)
        const int terms = fast ? 2 : 3;
        // a wave keeps 64 columns of a row tile, or 128 when the product is wider than 64 (A is then read once per 128)
        const int ct = n > 64 ? 8 : 4;
        // B in LDS: 2 bytes per term and element, at most 160 KB; a deeper K goes through in slabs
        const int slab = (int)std::min<int64_t>(k / 32, std::min<int64_t>(8, (160 * 1024) / ((int64_t)ct * terms * 64 * sizeof(f32x4))));
        const size_t split_bytes = (size_t)slab * ct * terms * 64 * sizeof(f32x4);
        // one persistent workgroup of sixteen waves per compute unit (and column block)
        dim3 sgrid((unsigned)std::min<int64_t>(row_tiles, gn::compute_units()), (unsigned)gn::ceil_div(n, 16 * ct), 1);
        hipStream_t st = gn::as_stream(stream);
        if (ct == 8) return fast ? launch_split<2, 8>(g, row_tiles, sgrid, split_bytes, slab, st) : launch_split<3, 8>(g, row_tiles, sgrid, split_bytes, slab, st);
        return fast ? launch_split<2, 4>(g, row_tiles, sgrid, split_bytes, slab, st) : launch_split<3, 4>(g, row_tiles, sgrid, split_bytes, slab, st);
    }
    if (batch == 1 && m >= 256 && lds_bytes <= 64 * 1024 && !gn::fast_paths_disabled()) {      // tall-skinny on the fp32 instruction
        if (batch_open && (flags & GN_GEMM_JOIN_BATCH) && !a_rows) {   // between gn_dense_batch_begin / _end: leaves with the others
            BatchOp op;
            op.kind = 2; op.g = g; op.first = 0; op.mt = row_tiles; op.nt = 0; op.lds = (int)lds_bytes;
            op.gx = (int)std::min<int64_t>(gn::ceil_div(row_tiles, 16), 256);
            op.blocks = op.gx * (int)gn::ceil_div(n, 16 * kColTiles);
            batch_queue.push_back(op);
            return GN_OK;
        }
        return launch_lds(g, gn::as_stream(stream));
    }
    dim3 grid((unsigned)gn::ceil_div(m, 64), (unsigned)gn::ceil_div(n, 16 * kColTiles), (unsigned)batch);
    k_gemm_f32<<<grid, 256, 0, gn::as_stream(stream)>>>(g);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_merge_f32(float* dst, int64_t ld_dst, const float* src, int64_t ld_src, const float* src2,
                       int64_t ld_src2, int64_t rows, int64_t cols, int mode, void* stream) {
    GN_REQUIRE(rows >= 0 && cols >= 0 && cols < (1ll << 31), "bad merge size");
    GN_REQUIRE(mode >= 0 && mode <= 10, "unknown merge mode %d", mode);
    if (rows == 0 || cols == 0) return GN_OK;
    GN_REQUIRE(dst && src && (mode < 4 || mode == 7 || mode == 8 || src2), "merge operand pointer is null");
    k_merge<<<gn::stream_grid(rows * cols, 256), 256, 0, gn::as_stream(stream)>>>(dst, ld_dst, src, ld_src, src2,
                                                                                ld_src2, rows, (int)cols, mode);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_softmax_rows_f32(float* x, int64_t ld, int64_t rows, int64_t cols, void* stream) {
    GN_REQUIRE(rows >= 0 && cols >= 0 && cols < (1ll << 31), "bad softmax size");
    if (rows == 0 || cols == 0) return GN_OK;
    GN_REQUIRE(x != nullptr, "softmax operand is null");
    k_softmax_rows<<<gn::stream_grid(rows * 64, 256), 256, 0, gn::as_stream(stream)>>>(x, ld, rows, (int)cols);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_softmax_rows_backward_f32(const float* probs, int64_t ld_probs, const float* grad, int64_t ld_grad, float* dx,
                                       int64_t ld_dx, int64_t rows, int64_t cols, void* stream) {
    GN_REQUIRE(rows >= 0 && cols >= 0 && cols < (1ll << 31), "bad softmax size");
    if (rows == 0 || cols == 0) return GN_OK;
    GN_REQUIRE(probs && grad && dx && ld_probs >= cols && ld_grad >= cols && ld_dx >= cols, "softmax operand is null or a leading dimension too small");
    k_softmax_rows_backward<<<gn::stream_grid(rows * 64, 256), 256, 0, gn::as_stream(stream)>>>(probs, ld_probs, grad, ld_grad, dx, ld_dx,
                                                                                                rows, (int)cols);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_class_scores_f32(const float* z, int64_t ld_z, int64_t table_rows, const int64_t* node_list, int64_t m,
                              const float* w, int64_t ld_w, int64_t k, int64_t n, int softmax, float* out, int64_t ld_out,
                              void* stream) {
    GN_REQUIRE(m >= 0 && k >= 0 && n >= 0 && table_rows >= 0, "negative size");
    if (m == 0 || n == 0) return GN_OK;
    GN_REQUIRE(z && w && out, "operand pointer is null");
    GN_REQUIRE(ld_z >= k && ld_w >= n && ld_out >= n, "leading dimension smaller than the row length");
    if (n > kClassMax || (size_t)k * kClassMax * sizeof(float) > 160 * 1024 || gn::fast_paths_disabled()) {
        // many classes: the row-gather GEMM, then the softmax pass
        gn_status s = gn_gemm_f32(z, ld_z, 0, node_list, table_rows, w, ld_w, 0, out, ld_out, 0, m, n, k, 1, nullptr, 0, stream);
        if (s != GN_OK || !softmax) return s;
        return gn_softmax_rows_f32(out, ld_out, m, n, stream);
    }
    const size_t lds = (size_t)k * ((n + 3) / 4) * sizeof(f32x4);
    { gn_status ls = gn::allow_large_lds(reinterpret_cast<const void*>(k_class_scores), 160 * 1024); if (ls != GN_OK) return ls; }
    const int grid = (int)std::min<int64_t>(gn::ceil_div(m, 64), 16 * gn::compute_units());
    k_class_scores<<<grid, 1024, lds, gn::as_stream(stream)>>>(z, ld_z, table_rows, node_list, m, w, ld_w, (int)k, (int)n, softmax,
                                                                out, ld_out);
    GN_LAUNCH_CHECK();
    return GN_OK;
}


int gn_xtg_wide_supported(int64_t m, int64_t k1, int64_t k2) {
    if (gn::fast_paths_disabled() || m < 4096 || k1 < 64 || k2 < 32 || k1 % 64 != 0 || k2 % 32 != 0 || k1 > 256 || k2 > 128) return 0;
    const int64_t tiles = (k1 / 64) * (k2 / 32);
    return tiles >= 2 && tiles <= 16 && 16 % tiles == 0 ? 1 : 0;
}

size_t gn_xtg_workspace_bytes(int64_t k1, int64_t k2) {
    if (k1 <= 0 || k2 <= 0) return 0;
    return (size_t)kXtgSlices * k1 * k2 * sizeof(float) + 64;           // (+ the ticket of the one-launch kernel)
}

gn_status gn_xtg_f32(const float* x, int64_t ld_x, const float* g, int64_t ld_g, int64_t m, int64_t k1, int64_t k2, float* out,
                     int64_t ld_out, void* workspace, size_t workspace_bytes, int flags, void* stream) {
    GN_REQUIRE(m >= 0 && k1 >= 0 && k2 >= 0, "negative size");
    if (k1 == 0 || k2 == 0) return GN_OK;
    const bool wide = gn_xtg_wide_supported(m, k1, k2) != 0;
    if (k1 * k2 > 4096 && !wide)
        return gn::fail(GN_ERR_UNSUPPORTED, "x^T g: %lld x %lld outputs (at most 4096, or a wide product: gn_xtg_wide_supported)", (long long)k1, (long long)k2);
    GN_REQUIRE(out && ld_out >= k2, "output pointer is null or its leading dimension too small");
    GN_REQUIRE(m == 0 || (x && g && ld_x >= k1 && ld_g >= k2), "operand pointer is null or a leading dimension too small");
    GN_REQUIRE(workspace && workspace_bytes >= gn_xtg_workspace_bytes(k1, k2), "workspace too small: need %zu bytes",
               gn_xtg_workspace_bytes(k1, k2));
    hipStream_t st = gn::as_stream(stream);
    if (wide) {
        // all tiles of a row slice in one workgroup, then the fold (two launches whatever the width; never queued in a batch)
        const int ti = (int)(k1 / 64), tj = (int)(k2 / 32), wpt = 16 / (ti * tj);
        const int64_t chunks = gn::ceil_div(m, 16);
        const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(kXtgSlices, gn::compute_units()), chunks / (2 * wpt)));
        const XtgArgs a = {x, ld_x, g, ld_g, m, (int)k1, (int)k2, static_cast<float*>(workspace), nullptr, out, ld_out};
        const size_t lds = wpt > 1 ? (size_t)16 * 8 * 64 * sizeof(f32x4) : 0;
#define GN_WIDE_CASE(TI, TJ)                                                                                              \
        if (ti == TI && tj == TJ) {                                                                                       \
            if (lds > 64 * 1024) { const gn_status ls = gn::allow_large_lds(reinterpret_cast<const void*>(k_xtg_wide<TI, TJ>), 136 * 1024); if (ls != GN_OK) return ls; } \
            k_xtg_wide<TI, TJ><<<slices, 1024, lds, st>>>(a);                                                             \
        }
        GN_WIDE_CASE(1, 2) GN_WIDE_CASE(2, 1) GN_WIDE_CASE(2, 2) GN_WIDE_CASE(1, 4) GN_WIDE_CASE(4, 1) GN_WIDE_CASE(2, 4) GN_WIDE_CASE(4, 2) GN_WIDE_CASE(4, 4)
#undef GN_WIDE_CASE
        GN_LAUNCH_CHECK();
        k_xtg_fold<<<(unsigned)gn::ceil_div(k1 * k2, 16), 256, 0, st>>>(static_cast<const float*>(workspace), slices, (int)(k1 * k2), (int)k2, out, ld_out);
        GN_LAUNCH_CHECK();
        return GN_OK;
    }
    if (m > 0 && k1 <= 64 && k2 <= 32 && (flags & GN_XTG_TICKET_ZEROED) && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0 &&
        !gn::fast_paths_disabled()) {
        const int mt = (int)gn::ceil_div(k1, 16), nt = (int)gn::ceil_div(k2, 16);
        bool ws_free = true;                                  // a queued product owns its workspace until the batch has left
        for (const BatchOp& q : batch_queue) ws_free = ws_free && !(q.kind == 1 && q.x.partial == workspace);
        if (batch_open && (flags & GN_XTG_JOIN_BATCH) && ws_free) {
            BatchOp op;
            op.kind = 1; op.mt = mt; op.nt = nt; op.first = 0; op.gx = 1; op.blocks = xtg_slices(m);
            op.lds = (int)((size_t)16 * mt * nt * 64 * sizeof(f32x4));
            op.x = xtg_args(x, ld_x, g, ld_g, m, (int)k1, (int)k2, out, ld_out, workspace);
            batch_queue.push_back(op);
            return GN_OK;
        }
#define GN_XTG_CASE(MT, NT) if (mt == MT && nt == NT) return launch_xtg_mfma<MT, NT>(x, ld_x, g, ld_g, m, (int)k1, (int)k2, out, ld_out, workspace, st)
        GN_XTG_CASE(1, 1); GN_XTG_CASE(2, 1); GN_XTG_CASE(3, 1); GN_XTG_CASE(4, 1);
        GN_XTG_CASE(1, 2); GN_XTG_CASE(2, 2); GN_XTG_CASE(3, 2); GN_XTG_CASE(4, 2);
#undef GN_XTG_CASE
    }
    const int slices = (int)std::max<int64_t>(1, std::min<int64_t>(kXtgSlices, gn::ceil_div(m, 16)));
    const size_t lds = (size_t)kXtgMaxRows * (k1 + 1 + k2) * sizeof(float);
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_xtg_partial), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    k_xtg_partial<<<slices, 256, lds, st>>>(x, ld_x, g, ld_g, m, (int)k1, (int)k2, static_cast<float*>(workspace));
    GN_LAUNCH_CHECK();
    k_xtg_fold<<<(unsigned)gn::ceil_div(k1 * k2, 16), 256, 0, st>>>(static_cast<const float*>(workspace), slices, (int)(k1 * k2), (int)k2,
                                                                  out, ld_out);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_dense_batch_begin(void) {
    GN_REQUIRE(!batch_open, "gn_dense_batch_begin inside an open batch");
    batch_queue.clear();
    batch_open = true;
    return GN_OK;
}

gn_status gn_dense_batch_end(void* stream) {
    GN_REQUIRE(batch_open, "gn_dense_batch_end without gn_dense_batch_begin");
    batch_open = false;
    return flush_batch(gn::as_stream(stream));
}

}  // extern "C"

