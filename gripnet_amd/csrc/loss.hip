// The link-prediction loss of the training loop (GripNet-pose.py:140-142, EPS = 1e-13 from gripnet/utils.py:10):
//
//   loss = - mean( log(pos + eps) ) - mean( log(1 - neg + eps) )
//
// as ONE launch forward and ONE launch backward.  Written with torch ops it is ~10 element-wise / reduction launches
// forward and as many backward (log, add, neg, mean, rsub, ... over 2 x 2 M scores: ~100 us of the 0.95 ms PoSE step).
// Deterministic: every workgroup sums a fixed slice in a fixed order, the last workgroup to arrive adds the partial sums
// in workgroup order (not in arrival order), in double.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

constexpr int kLossGroups = 512, kLossThreads = 256;

// A thread reads its share with 16-byte loads, four of them in flight: one float at a time it waits out a memory round trip per
// term (60 us for 2 x 2 M scores instead of ~6).
template <bool VEC>
__global__ __launch_bounds__(kLossThreads) void k_link_loss(const float* __restrict__ pos, int64_t n_pos, const float* __restrict__ neg,
                                                           int64_t n_neg, float eps, double* __restrict__ partial,
                                                           unsigned int* __restrict__ counter, float* __restrict__ loss) {
    __shared__ double red[2][kLossThreads / 64];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // log(pos + eps) and log(1 - neg + eps), the additions in the reference's order (GripNet-pose.py:140-141)
    float sp, sn;
    {
        const int64_t t = (int64_t)blockIdx.x * kLossThreads + tid, stride = (int64_t)gridDim.x * kLossThreads;
        float sum = 0.f;
        int64_t done = 0;
        if constexpr (VEC) {
            const f32x4* __restrict__ v = reinterpret_cast<const f32x4*>(pos);
            const int64_t nv = n_pos / 4;
            int64_t i = t;
            for (; i + 3 * stride < nv; i += 4 * stride) {
                const f32x4 a = v[i], b = v[i + stride], c = v[i + 2 * stride], d = v[i + 3 * stride];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    sum += (__logf(a[k] + eps) + __logf(b[k] + eps)) + (__logf(c[k] + eps) + __logf(d[k] + eps));
            }
            for (; i < nv; i += stride) {
                const f32x4 a = v[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) sum += __logf(a[k] + eps);
            }
            done = nv * 4;
        }
        for (int64_t i = done + t; i < n_pos; i += stride) sum += __logf(pos[i] + eps);
        sp = sum;
    }
    {
        const int64_t t = (int64_t)blockIdx.x * kLossThreads + tid, stride = (int64_t)gridDim.x * kLossThreads;
        float sum = 0.f;
        int64_t done = 0;
        if constexpr (VEC) {
            const f32x4* __restrict__ v = reinterpret_cast<const f32x4*>(neg);
            const int64_t nv = n_neg / 4;
            int64_t i = t;
            for (; i + 3 * stride < nv; i += 4 * stride) {
                const f32x4 a = v[i], b = v[i + stride], c = v[i + 2 * stride], d = v[i + 3 * stride];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    sum += (__logf(1.0f - a[k] + eps) + __logf(1.0f - b[k] + eps)) + (__logf(1.0f - c[k] + eps) + __logf(1.0f - d[k] + eps));
            }
            for (; i < nv; i += stride) {
                const f32x4 a = v[i];
#pragma unroll
                for (int k = 0; k < 4; ++k) sum += __logf(1.0f - a[k] + eps);
            }
            done = nv * 4;
        }
        for (int64_t i = done + t; i < n_neg; i += stride) sum += __logf(1.0f - neg[i] + eps);
        sn = sum;
    }
    double dp = sp, dn = sn;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { dp += __shfl_xor(dp, off); dn += __shfl_xor(dn, off); }
    if (lane == 0) { red[0][wave] = dp; red[1][wave] = dn; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kLossThreads / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        // the partial sums are handed over write-through (sc1), the counter is an agent-scope atomic, the last arriver
        // reads them with sc1 loads: the form MI355X_MICROARCH.md lists for a last-arriver hand-over of a few bytes
        __builtin_nontemporal_store(a, partial + 2 * blockIdx.x);
        __builtin_nontemporal_store(b, partial + 2 * blockIdx.x + 1);
        __threadfence();
        const unsigned int arrived = atomicAdd(counter, 1u);
        last = arrived == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // the last workgroup to arrive adds the partial sums: thread t takes workgroups t, t + 256, ... in that order, then a fixed
    // tree over the threads - the same association whatever the arrival order was
    __threadfence();
    double a = 0.0, b = 0.0;
    for (unsigned g = tid; g < gridDim.x; g += kLossThreads) {
        a += __builtin_nontemporal_load(partial + 2 * g);
        b += __builtin_nontemporal_load(partial + 2 * g + 1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
    __syncthreads();                                               // (red was read above by thread 0 only, before the barrier)
    if (lane == 0) { red[0][wave] = a; red[1][wave] = b; }
    __syncthreads();
    if (tid == 0) {
        a = 0.0; b = 0.0;
        for (int w = 0; w < kLossThreads / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        const double lp = n_pos > 0 ? a / (double)n_pos : 0.0, ln = n_neg > 0 ? b / (double)n_neg : 0.0;
        *loss = (float)(-lp - ln);
        *counter = 0u;                                             // ready for the next launch (stream-ordered)
    }
}

__global__ __launch_bounds__(256) void k_link_loss_grad(const float* __restrict__ pos, int64_t n_pos, const float* __restrict__ neg,
                                                        int64_t n_neg, float eps, const float* __restrict__ upstream,
                                                        float* __restrict__ dpos, float* __restrict__ dneg) {
    const float g = upstream ? *upstream : 1.0f;
    const float cp = n_pos > 0 ? -g / (float)n_pos : 0.f, cn = n_neg > 0 ? g / (float)n_neg : 0.f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos; i += stride) dpos[i] = cp / (pos[i] + eps);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_neg; i += stride) dneg[i] = cn / (1.0f - neg[i] + eps);
}

}  // namespace

extern "C" {

size_t gn_link_loss_workspace_bytes(void) { return (size_t)kLossGroups * 2 * sizeof(double) + 64; }

gn_status gn_link_loss_forward_f32(const float* pos_score, int64_t num_pos, const float* neg_score, int64_t num_neg, float eps,
                                   float* loss, void* workspace, size_t workspace_bytes, void* stream) {
    GN_REQUIRE(num_pos >= 0 && num_neg >= 0, "negative score count");
    GN_REQUIRE((num_pos == 0 || pos_score) && (num_neg == 0 || neg_score) && loss, "score / loss pointer is null");
    GN_REQUIRE(workspace && workspace_bytes >= gn_link_loss_workspace_bytes() && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0,
               "workspace too small or unaligned: need %zu bytes, 8-byte aligned, zero-initialised once", gn_link_loss_workspace_bytes());
    double* partial = static_cast<double*>(workspace);
    unsigned int* counter = reinterpret_cast<unsigned int*>(partial + 2 * kLossGroups);
    if (aligned16(pos_score) && aligned16(neg_score))
        k_link_loss<true><<<kLossGroups, kLossThreads, 0, gn::as_stream(stream)>>>(pos_score, num_pos, neg_score, num_neg, eps, partial, counter, loss);
    else
        k_link_loss<false><<<kLossGroups, kLossThreads, 0, gn::as_stream(stream)>>>(pos_score, num_pos, neg_score, num_neg, eps, partial, counter, loss);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_link_loss_backward_f32(const float* pos_score, int64_t num_pos, const float* neg_score, int64_t num_neg, float eps,
                                    const float* upstream_grad, float* dpos, float* dneg, void* stream) {
    GN_REQUIRE(num_pos >= 0 && num_neg >= 0, "negative score count");
    GN_REQUIRE((num_pos == 0 || (pos_score && dpos)) && (num_neg == 0 || (neg_score && dneg)), "score / gradient pointer is null");
    if (num_pos + num_neg == 0) return GN_OK;
    k_link_loss_grad<<<gn::stream_grid(std::max(num_pos, num_neg), 256, 1024), 256, 0, gn::as_stream(stream)>>>(
        pos_score, num_pos, neg_score, num_neg, eps, upstream_grad, dpos, dneg);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // extern "C"
