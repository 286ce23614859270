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

// A thread reads its share with 16-byte loads, eight of them in flight: one float at a time it waits out a memory round trip
// per term (60 us for 2 x 2 M scores).
template <bool VEC>
__global__ __launch_bounds__(kLossThreads) void k_link_loss(const float* __restrict__ pos, int64_t n_pos, const float* __restrict__ neg,
                                                           int64_t n_neg, float eps, double* __restrict__ partial,
                                                           unsigned int* __restrict__ counter, float* __restrict__ loss) {
    __shared__ double red[2][kLossThreads / 64];
    __shared__ bool last;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // log(pos + eps) and log(1 - neg + eps), the additions in the reference's order (GripNet-pose.py:140-141).  Four 16-byte
    // loads of each list are requested before anything is summed, whatever the list's length (clamped index, masked sum): with
    // 2 M scores a thread's share is under four float4, and a loop that only unrolls when four full trips exist never did.
    float sp = 0.f, sn = 0.f;
    {
        const int64_t t = (int64_t)blockIdx.x * kLossThreads + tid, stride = (int64_t)gridDim.x * kLossThreads;
        if constexpr (VEC) {
            const f32x4* __restrict__ vp = reinterpret_cast<const f32x4*>(pos);
            const f32x4* __restrict__ vn = reinterpret_cast<const f32x4*>(neg);
            const int64_t np4 = n_pos / 4, nn4 = n_neg / 4, most = np4 > nn4 ? np4 : nn4;
            for (int64_t i = t; i < most; i += 4 * stride) {
                f32x4 a[4], b[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t j = i + k * stride;
                    a[k] = np4 > 0 ? vp[j < np4 ? j : np4 - 1] : (f32x4){1.f, 1.f, 1.f, 1.f};      // (uniform: an empty list may be null)
                    b[k] = nn4 > 0 ? vn[j < nn4 ? j : nn4 - 1] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int64_t j = i + k * stride;
                    if (j < np4) sp += (__logf(a[k][0] + eps) + __logf(a[k][1] + eps)) + (__logf(a[k][2] + eps) + __logf(a[k][3] + eps));
                    if (j < nn4) sn += (__logf(1.0f - b[k][0] + eps) + __logf(1.0f - b[k][1] + eps)) + (__logf(1.0f - b[k][2] + eps) + __logf(1.0f - b[k][3] + eps));
                }
            }
            for (int64_t i = np4 * 4 + t; i < n_pos; i += stride) sp += __logf(pos[i] + eps);
            for (int64_t i = nn4 * 4 + t; i < n_neg; i += stride) sn += __logf(1.0f - neg[i] + eps);
        } else {
            for (int64_t i = t; i < n_pos; i += stride) sp += __logf(pos[i] + eps);
            for (int64_t i = t; i < n_neg; i += stride) sn += __logf(1.0f - neg[i] + eps);
        }
    }
    double dp = sp, dn = sn;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { dp += __shfl_xor(dp, off); dn += __shfl_xor(dn, off); }
    if (lane == 0) { red[0][wave] = dp; red[1][wave] = dn; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, b = 0.0;
        for (int w = 0; w < kLossThreads / 64; ++w) { a += red[0][w]; b += red[1][w]; }
        // hand-over without fences (a release would write back the XCD's L2 512 times): the two partial sums are stored
        // write-through (agent-scope relaxed atomic stores), drained, then the ticket is drawn; the last arriver reads them
        // with agent-scope loads - the form MI355X_MICROARCH.md lists for a last-arriver hand-over of a few bytes
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(partial) + 2 * blockIdx.x, (unsigned long long)__double_as_longlong(a),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(partial) + 2 * blockIdx.x + 1, (unsigned long long)__double_as_longlong(b),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned int arrived = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = arrived == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    // the last workgroup to arrive adds the partial sums: thread t takes workgroups t, t + 256, ... in that order, then a fixed
    // tree over the threads - the same association whatever the arrival order was
    // (two workgroups' pairs of sums requested before any is added: as a load - add chain the 512 workgroups' sums were four
    // memory round trips in this one workgroup)
    double a = 0.0, b = 0.0;
    const unsigned long long* __restrict__ ps = reinterpret_cast<const unsigned long long*>(partial);
    for (unsigned g0 = tid; g0 < gridDim.x; g0 += 2 * kLossThreads) {
        const unsigned g1 = min(g0 + kLossThreads, gridDim.x - 1);
        const unsigned long long a0 = __hip_atomic_load(ps + 2 * g0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b0 = __hip_atomic_load(ps + 2 * g0 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long a1 = __hip_atomic_load(ps + 2 * g1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b1 = __hip_atomic_load(ps + 2 * g1 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a += __longlong_as_double((long long)a0);
        b += __longlong_as_double((long long)b0);
        if (g0 + kLossThreads < gridDim.x) { a += __longlong_as_double((long long)a1); b += __longlong_as_double((long long)b1); }
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
        __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch (stream-ordered)
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

// The node-classification loss of the training loops (GripNet-aminer.py:133, every freebase driver alike):
// loss = - mean_i log(score[i, class_i] + eps).  A few thousand labelled nodes: one workgroup, a fixed slice per thread summed
// in double, a fixed tree over the threads; logf, not the hardware's fast logarithm (the reference's torch.log: a class
// probability near 1 has a logarithm near 0, where the fast intrinsic's absolute error is the whole value); four nodes per
// thread in flight (the loads are a gather: one score of every labelled row).
__global__ __launch_bounds__(1024) void k_class_loss(const float* __restrict__ score, int64_t ld, const int64_t* __restrict__ cls, int64_t n,
                                                     int classes, float eps, float* __restrict__ loss, int32_t* __restrict__ err) {
    __shared__ double red[16];
    double s = 0.0;
    int64_t i = threadIdx.x;
    for (; i + 3 * 1024 < n; i += 4 * 1024) {
        int64_t c[4];
        float v[4];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) { c[k] = cls[i + k * 1024]; bad = bad || (uint64_t)c[k] >= (uint64_t)classes; }
        if (bad) {                                              // (rare: report, and count only the rows with a class inside the table)
            if (err) atomicOr(err, 1);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if ((uint64_t)c[k] < (uint64_t)classes) s += (double)logf(score[(i + k * 1024) * ld + c[k]] + eps);
            continue;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = score[(i + k * 1024) * ld + c[k]];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += (double)logf(v[k] + eps);
    }
    for (; i < n; i += 1024) {
        const int64_t c = cls[i];
        if ((uint64_t)c < (uint64_t)classes) s += (double)logf(score[i * ld + c] + eps);
        else if (err) atomicOr(err, 1);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        *loss = n > 0 ? (float)(-t / (double)n) : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_class_loss_grad(const float* __restrict__ score, int64_t ld, const int64_t* __restrict__ cls, int64_t n,
                                                         int classes, float eps, const float* __restrict__ upstream,
                                                         float* __restrict__ dscore, int64_t ld_d) {
    const float g = (upstream ? *upstream : 1.0f) / (float)n;
    const int64_t total = n * classes, stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const int64_t i = t / classes;
        const int c = (int)(t - i * classes);
        dscore[i * ld_d + c] = (int64_t)c == cls[i] ? -g / (score[i * ld + c] + eps) : 0.f;
    }
}

}  // namespace

extern "C" {

gn_status gn_class_loss_forward_f32(const float* score, int64_t ld_score, const int64_t* classes, int64_t num_nodes, int64_t num_classes,
                                    float eps, float* loss, int32_t* error_flag, void* stream) {
    GN_REQUIRE(num_nodes >= 0 && num_classes >= 1 && num_classes < (1ll << 31) && ld_score >= num_classes, "bad class-loss size");
    GN_REQUIRE(loss && (num_nodes == 0 || (score && classes)), "score / class / loss pointer is null");
    k_class_loss<<<1, 1024, 0, gn::as_stream(stream)>>>(score, ld_score, classes, num_nodes, (int)num_classes, eps, loss, error_flag);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_class_loss_backward_f32(const float* score, int64_t ld_score, const int64_t* classes, int64_t num_nodes, int64_t num_classes,
                                     float eps, const float* upstream_grad, float* dscore, int64_t ld_dscore, void* stream) {
    GN_REQUIRE(num_nodes >= 0 && num_classes >= 1 && num_classes < (1ll << 31) && ld_score >= num_classes && ld_dscore >= num_classes,
               "bad class-loss size");
    if (num_nodes == 0) return GN_OK;
    GN_REQUIRE(score && classes && dscore, "score / class / gradient pointer is null");
    k_class_loss_grad<<<gn::stream_grid(num_nodes * num_classes, 256, 1024), 256, 0, gn::as_stream(stream)>>>(
        score, ld_score, classes, num_nodes, (int)num_classes, eps, upstream_grad, dscore, ld_dscore);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

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
