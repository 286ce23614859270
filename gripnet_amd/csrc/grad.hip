// Element-wise glue of the layers' backward passes in one launch per layer (autograd of gripnet/layers.py:71-100,165-197):
//
//   gm = (saved_out > 0 ? g : 0)            the ReLU that follows every conv (layers.py:279,305,370), by the saved output
//   gd = gm / rowdiv                        the mean over the incoming edges of the relational layer (layers.py:191)
//   colsum[c] = sum_rows gm[row, c]         the bias gradient
//
// Written with torch ops these are a strided copy, a mask, a division and a column sum (itself a memset and a two-pass
// reduction): four to six launches of ~5 us each per layer inside the replayed training step.  Deterministic: a workgroup
// sums a fixed slice of rows in a fixed order, the last workgroup to arrive adds the slices in workgroup order.
#include "common.h"

namespace {

constexpr int kGradThreads = 256, kGradGroups = 1024;
constexpr int kGradSet = 32;                                 // workgroups whose column sums the last of them adds (then the sets' sums: two levels)

template <int CP>     // columns padded to a power of two: thread (tid / CP, tid % CP) keeps its column
__global__ __launch_bounds__(kGradThreads) void k_grad_prologue(const float* __restrict__ g, int64_t ld_g, const float* __restrict__ saved,
                                                               int64_t ld_saved, const float* __restrict__ rowdiv, int64_t rows, int cols,
                                                               float* __restrict__ gm, int64_t ld_gm, float* __restrict__ gd, int64_t ld_gd,
                                                               float* __restrict__ colsum, float* __restrict__ partial,
                                                               unsigned int* __restrict__ arrived) {
    constexpr int RP = kGradThreads / CP;                                  // rows per pass of a workgroup
    __shared__ float red[kGradThreads];
    __shared__ bool last;
    const int tid = threadIdx.x, c = tid % CP, rl = tid / CP;
    const int64_t per = (rows + gridDim.x - 1) / gridDim.x;
    const int64_t r0 = (int64_t)blockIdx.x * per, r1 = min(rows, r0 + per);
    float sum = 0.f;
    if (c < cols) {
        // four rows per trip, their loads requested together (one row per trip waits out a memory round trip per element)
        for (int64_t r = r0 + rl; r < r1; r += 4 * RP) {
            float v[4], s[4], d[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t rr = min(r + k * RP, r1 - 1);
                v[k] = g[rr * ld_g + c];
                s[k] = saved ? saved[rr * ld_saved + c] : 1.f;
                d[k] = gd ? rowdiv[rr] : 1.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t rr = r + k * RP;
                if (rr < r1) {
                    const float x = s[k] > 0.f ? v[k] : 0.f;
                    if (gm) gm[rr * ld_gm + c] = x;
                    if (gd) gd[rr * ld_gd + c] = x / d[k];
                    sum += x;
                }
            }
        }
    }
    if (!colsum) return;
    red[tid] = sum;
    __syncthreads();
    // Two levels, both "the last to arrive adds in index order" (write-through stores, drained, ticket; agent-scope loads): the
    // workgroups in sets of kGradSet consecutive ones, then the sets.  (One level - the last of up to 1,024 workgroups adding
    // all their sums, RP row lanes per column - was a chain of 64-512 memory round trips in ONE workgroup: 128 of the 145 us of a
    // 50,000 x 128 layer's launch.)
    unsigned int* mine = reinterpret_cast<unsigned int*>(partial) + (size_t)blockIdx.x * CP;
    if (tid < CP) {                                                        // the workgroup's row lanes, in order
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < RP; ++k) s += red[k * CP + tid];
        __hip_atomic_store(mine + tid, __float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this wave's write-through stores have reached L2 ...
    __syncthreads();                                                       // ... and so have every other wave's when the ticket is drawn
    // (a launch whose workgroups' sums are two requests deep for the last one - 16 x RP of them - keeps ONE level: the sets
    // would cost it a hand-over more, ~1 us on the PoSE layers)
    const bool one_level = gridDim.x <= 16u * RP;
    const unsigned set = one_level ? 0u : blockIdx.x / kGradSet, n_sets = one_level ? 1u : (gridDim.x + kGradSet - 1) / kGradSet;
    const unsigned set_lo = set * kGradSet, set_n = one_level ? gridDim.x : min((unsigned)kGradSet, gridDim.x - set_lo);
    if (tid == 0) last = __hip_atomic_fetch_add(arrived + 1 + set, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == set_n - 1;
    __syncthreads();
    if (!last) return;
    // the sums of this set: row lane k takes its workgroups k, k + RP, ... in that order (eight requested before any is added),
    // then the row lanes are added in order - the same association whatever the arrival order was
    auto fold = [&](const unsigned int* base, unsigned count) {
        float s = 0.f;
        for (unsigned b0 = rl; b0 < count; b0 += 8 * RP) {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const unsigned b = min(b0 + k * RP, count - 1);
                v[k] = __uint_as_float(__hip_atomic_load(base + (size_t)b * CP + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) s += (b0 + k * RP < count) ? v[k] : 0.f;
        }
        __syncthreads();                                                   // (red was read above)
        red[tid] = s;
        __syncthreads();
        float t = 0.f;
        if (tid < CP)
#pragma unroll
            for (int k = 0; k < RP; ++k) t += red[k * CP + tid];
        return t;
    };
    const float set_sum = fold(reinterpret_cast<const unsigned int*>(partial) + (size_t)set_lo * CP, set_n);
    if (one_level) {
        if (tid < cols) colsum[tid] = set_sum;
        if (tid == 0) __hip_atomic_store(arrived + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    unsigned int* sets = reinterpret_cast<unsigned int*>(partial) + (size_t)kGradGroups * 256;      // [kGradGroups / kGradSet][256]
    if (tid < CP) __hip_atomic_store(sets + (size_t)set * CP + tid, __float_as_uint(set_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_store(arrived + 1 + set, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
        last = __hip_atomic_fetch_add(arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_sets - 1;
    }
    __syncthreads();
    if (!last) return;
    const float total = fold(sets, n_sets);
    if (tid < cols) colsum[tid] = total;
    if (tid == 0) __hip_atomic_store(arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace

extern "C" {

size_t gn_grad_prologue_workspace_bytes(void) {      // the workgroups' sums, the sets' sums, the tickets (one + one per set)
    return ((size_t)kGradGroups * 256 + (size_t)(kGradGroups / kGradSet) * 256) * sizeof(float) + (size_t)(1 + kGradGroups / kGradSet) * sizeof(unsigned int) + 60;
}

gn_status gn_grad_prologue_f32(const float* g, int64_t ld_g, const float* saved_out, int64_t ld_saved, const float* rowdiv, int64_t rows,
                               int64_t cols, float* gm, int64_t ld_gm, float* gd, int64_t ld_gd, float* colsum, void* workspace,
                               size_t workspace_bytes, void* stream) {
    GN_REQUIRE(rows >= 0 && cols >= 0, "bad size");
    if (rows == 0 || cols == 0) {
        if (colsum && cols > 0) GN_HIP(hipMemsetAsync(colsum, 0, (size_t)cols * sizeof(float), gn::as_stream(stream)));
        return GN_OK;
    }
    GN_REQUIRE(g && ld_g >= cols && (!saved_out || ld_saved >= cols) && (!gm || ld_gm >= cols) && (!gd || (ld_gd >= cols && rowdiv)),
               "operand pointer is null or a leading dimension smaller than the row length");
    GN_REQUIRE(!colsum || (workspace && workspace_bytes >= gn_grad_prologue_workspace_bytes() && (reinterpret_cast<uintptr_t>(workspace) & 3) == 0),
               "column sums need a workspace of %zu bytes, zeroed once", gn_grad_prologue_workspace_bytes());
    float* partial = static_cast<float*>(workspace);
    unsigned int* arrived = workspace ? reinterpret_cast<unsigned int*>(partial + (size_t)kGradGroups * 256 + (size_t)(kGradGroups / kGradSet) * 256) : nullptr;
    hipStream_t st = gn::as_stream(stream);
    // a launch covers up to 256 columns (a thread keeps its column); wider layers take one launch per block of 256 columns, the
    // operands being row-strided already.  The launches share the workspace: they are ordered by the stream, and the last
    // workgroup of a launch leaves the ticket at zero.
    for (int64_t c0 = 0; c0 < cols; c0 += 256) {
        const int cw = (int)std::min<int64_t>(256, cols - c0);
        // (a workgroup covers 256 / CP rows per pass and four passes per trip: about two trips each)
        const int groups = (int)std::min<int64_t>(kGradGroups, gn::ceil_div(rows * std::max<int64_t>(cw, 16), 2048));
#define GN_GRAD_LAUNCH(CP) k_grad_prologue<CP><<<groups, kGradThreads, 0, st>>>(g + c0, ld_g, saved_out ? saved_out + c0 : nullptr, ld_saved, rowdiv, rows, cw, gm ? gm + c0 : nullptr, ld_gm, gd ? gd + c0 : nullptr, ld_gd, colsum ? colsum + c0 : nullptr, partial, arrived)
        if (cw <= 16) GN_GRAD_LAUNCH(16);
        else if (cw <= 32) GN_GRAD_LAUNCH(32);
        else if (cw <= 64) GN_GRAD_LAUNCH(64);
        else if (cw <= 128) GN_GRAD_LAUNCH(128);
        else GN_GRAD_LAUNCH(256);
#undef GN_GRAD_LAUNCH
        GN_LAUNCH_CHECK();
    }
    return GN_OK;
}

}  // extern "C"
