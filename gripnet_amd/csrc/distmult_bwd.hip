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
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>

namespace {

struct Rec { uint32_t a, b; float g; };       // 12 bytes: factor rows and the edge's upstream gradient

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
                            const int64_t* __restrict__ et, const float* __restrict__ gs, int64_t E, int64_t n,
                            int64_t R, uint32_t num_keys, uint32_t* __restrict__ keys, Rec* __restrict__ recs) {
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t uu = u[e], vv = v[e], rr = et[e];
        const bool ok = (uint64_t)uu < (uint64_t)n && (uint64_t)vv < (uint64_t)n && (uint64_t)rr < (uint64_t)R;
        Rec rec;
        uint32_t key;
        if (mode == 0) { key = (uint32_t)uu; rec.a = (uint32_t)vv; rec.b = (uint32_t)rr; }
        else if (mode == 1) { key = (uint32_t)vv; rec.a = (uint32_t)uu; rec.b = (uint32_t)rr; }
        else { key = (uint32_t)rr; rec.a = (uint32_t)uu; rec.b = (uint32_t)vv; }
        rec.g = ok ? gs[e] : 0.f;
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

}  // namespace

extern "C" size_t gn_distmult_backward_workspace_bytes(int64_t n, int64_t f, int64_t r, int64_t e) {
    if (n <= 0 || f <= 0 || r <= 0 || e <= 0) return 0;
    return ws_layout(e, std::max(n, r)).total;
}

extern "C" gn_status gn_distmult_backward_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                              const int64_t* v, const int64_t* et, const float* d, int64_t ld_d,
                                              int64_t r, int64_t e, const float* grad_logit, float* dz, int64_t ld_dz,
                                              float* dd, int64_t ld_dd, void* workspace, size_t workspace_bytes,
                                              void* stream) {
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
    GN_REQUIRE(z && u && v && et && d && grad_logit, "operand pointer is null");
    GN_REQUIRE(ld_z >= f && ld_d >= f, "leading dimension smaller than the row length");
    const WsLayout l = ws_layout(e, std::max(n, r));
    GN_REQUIRE(workspace && workspace_bytes >= l.total, "workspace too small: need %zu bytes", l.total);
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

    for (int mode = 0; mode < 3; ++mode) {
        const int64_t rows = mode == 2 ? r : n;
        const float* A = z;                              // a is always a node id
        const int64_t ld_a = ld_z;
        const float* B = mode == 2 ? z : d;              // b: relation row of D, or the other endpoint's z row
        const int64_t ld_b = mode == 2 ? ld_z : ld_d;
        float* out = mode == 2 ? dd : dz;
        const int64_t ld_out = mode == 2 ? ld_dd : ld_dz;
        const int accumulate = mode == 1;                // dz = u-side pass, then + v-side pass
        k_make_recs<<<gn::stream_grid(e, 256), 256, 0, st>>>(mode, u, v, et, grad_logit, e, n, r, (uint32_t)rows, keys, recs);
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
