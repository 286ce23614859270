// Per-relation link-prediction metrics on the device.
//
// After every epoch the reference computes, for each of the R relations, AUPRC / AUROC / AP of that
// relation's positive and negative scores with scikit-learn on the host: R device -> host copies and R
// sklearn calls per epoch (GripNet-pose.py:148-160,188-199; gripnet/utils.py:28-35).  Here one pass does
// all relations:
//   keys (relation << 32 | descending-order bits of the score) of the 2E scores, labels as values,
//   one radix sort; a global inclusive scan of the labels gives the true-positive count at every rank
//   (false positives = rank - TP); an exclusive max-scan gives, at every rank, the end of the previous
//   group of tied scores; every group end then contributes the trapezoid / step terms of the three
//   curves (the definitions scikit-learn uses: roc_auc_score, average_precision_score,
//   auc(precision_recall_curve) with the extra point (recall 0, precision 1)); a segmented reduction per
//   relation sums them in double precision.
#include "common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_reduce.hpp>

#include <vector>

namespace {

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

__device__ __forceinline__ uint32_t descending_bits(float x) {
    const uint32_t u = __float_as_uint(x);
    const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // monotone in x
    return ~asc;
}

__global__ void k_metric_keys(const float* __restrict__ pos, const float* __restrict__ neg,
                              const int64_t* __restrict__ starts, int R, int64_t E, uint64_t* __restrict__ keys,
                              int32_t* __restrict__ labels) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < 2 * E; i += (int64_t)gridDim.x * blockDim.x) {
        const bool is_pos = i < E;
        const int64_t e = is_pos ? i : i - E;
        int a = 0, b = R;                                    // last r with starts[r] <= e
        while (b - a > 1) {
            int mid = (a + b) >> 1;
            if (starts[mid] <= e) a = mid; else b = mid;
        }
        keys[i] = ((uint64_t)a << 32) | descending_bits(is_pos ? pos[e] : neg[e]);
        labels[i] = is_pos ? 1 : 0;
    }
}

__global__ void k_group_end_index(const uint64_t* __restrict__ keys, int64_t n, int32_t* __restrict__ ge) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        ge[i] = (i == n - 1 || keys[i + 1] != keys[i]) ? (int32_t)i : -1;
}

struct MaxOp {
    __device__ __host__ int32_t operator()(int32_t a, int32_t b) const { return a > b ? a : b; }
};

// terms of the three curves at every group end (0 elsewhere)
__global__ void k_metric_terms(const uint64_t* __restrict__ keys, const int32_t* __restrict__ tp_scan,
                               const int32_t* __restrict__ ge, const int32_t* __restrict__ prev_end,
                               const int64_t* __restrict__ starts, int64_t n, double* __restrict__ t_auprc,
                               double* __restrict__ t_auroc, double* __restrict__ t_ap) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double a = 0.0, b = 0.0, c = 0.0;
        if (ge[i] >= 0) {
            const int r = (int)(keys[i] >> 32);
            const int64_t s = 2 * starts[r], e = 2 * starts[r + 1];      // this relation's slice of the sorted array
            const double P = (double)(e - s) / 2, N = P;                  // E_r positives, E_r negatives
            const int64_t base = s > 0 ? tp_scan[s - 1] : 0;
            const double tp = (double)(tp_scan[i] - base), fp = (double)(i - s + 1) - tp;
            double tp0 = 0.0, fp0 = 0.0, prec0 = 1.0;                     // the curve's start point
            const int64_t j = prev_end[i];
            if (j >= s) {
                tp0 = (double)(tp_scan[j] - base);
                fp0 = (double)(j - s + 1) - tp0;
                prec0 = tp0 / (tp0 + fp0);
            }
            const double prec = tp / (tp + fp);
            a = (tp - tp0) / P * (prec + prec0) * 0.5;                     // trapezoid of the PR curve
            b = (fp - fp0) * (tp + tp0) * 0.5 / (P * N);                   // trapezoid of the ROC curve
            c = (tp - tp0) / P * prec;                                     // average precision step
        }
        t_auprc[i] = a; t_auroc[i] = b; t_ap[i] = c;
    }
}

__global__ void k_seg_offsets(const int64_t* __restrict__ starts, int R, int64_t* __restrict__ off) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= R) off[i] = 2 * starts[i];
}

__global__ void k_nan_empty(const int64_t* __restrict__ starts, int R, double* __restrict__ out) {   // [3][R]
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * R && starts[i % R + 1] == starts[i % R]) out[i] = __builtin_nan("");
}

struct Layout {
    size_t keys, keys_sorted, labels, labels_sorted, tp, ge, prev, t0, t1, t2, starts, offs, tmp, total;
};

Layout layout(int64_t E, int64_t R) {
    const size_t n = 2 * (size_t)E;
    size_t b_sort = 0, b_scan = 0, b_scan2 = 0, b_red = 0;
    (void)rocprim::radix_sort_pairs(nullptr, b_sort, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const int32_t*)nullptr,
                                    (int32_t*)nullptr, n, 0, 64, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b_scan, (const int32_t*)nullptr, (int32_t*)nullptr, n, rocprim::plus<int32_t>(),
                                  (hipStream_t)0);
    (void)rocprim::exclusive_scan(nullptr, b_scan2, (const int32_t*)nullptr, (int32_t*)nullptr, (int32_t)-1, n, MaxOp(),
                                  (hipStream_t)0);
    (void)rocprim::segmented_reduce(nullptr, b_red, (const double*)nullptr, (double*)nullptr, (unsigned)R,
                                    (const int64_t*)nullptr, (const int64_t*)nullptr, rocprim::plus<double>(), 0.0,
                                    (hipStream_t)0);
    Layout l;
    size_t o = 0;
    auto take = [&](size_t bytes) { size_t at = o; o += align_up(bytes); return at; };
    l.keys = take(n * 8); l.keys_sorted = take(n * 8); l.labels = take(n * 4); l.labels_sorted = take(n * 4);
    l.tp = take(n * 4); l.ge = take(n * 4); l.prev = take(n * 4);
    l.t0 = take(n * 8); l.t1 = take(n * 8); l.t2 = take(n * 8);
    l.starts = take((R + 1) * 8); l.offs = take((R + 1) * 8);
    l.tmp = take(std::max(std::max(b_sort, b_scan), std::max(b_scan2, b_red)));
    l.total = o;
    return l;
}

}  // namespace

extern "C" size_t gn_link_metrics_workspace_bytes(int64_t R, int64_t E) {
    if (R <= 0 || E <= 0) return 0;
    return layout(E, R).total;
}

// out: [3, R] float64 (device): AUPRC, AUROC, AP of every relation (NaN for a relation without edges).
extern "C" gn_status gn_link_metrics_f32(const float* pos_score, const float* neg_score, const int64_t* range_list_host,
                                         int64_t R, int64_t E, double* out, void* workspace, size_t workspace_bytes,
                                         void* stream) {
    GN_REQUIRE(R >= 1 && E >= 0 && 2 * E < (1ll << 31), "bad size (R=%lld, E=%lld)", (long long)R, (long long)E);
    GN_REQUIRE(out != nullptr && range_list_host != nullptr, "null pointer");
    hipStream_t st = gn::as_stream(stream);
    std::vector<int64_t> starts(R + 1);
    int64_t cursor = 0;
    for (int64_t r = 0; r < R; ++r) {
        GN_REQUIRE(range_list_host[2 * r] == cursor && range_list_host[2 * r + 1] >= cursor,
                   "range_list must tile [0,E) in relation order (row %lld)", (long long)r);
        starts[r] = cursor;
        cursor = range_list_host[2 * r + 1];
    }
    GN_REQUIRE(cursor == E, "range_list covers %lld edges, %lld scores given", (long long)cursor, (long long)E);
    starts[R] = E;
    if (E == 0) {
        std::vector<double> nan(3 * R, __builtin_nan(""));
        GN_HIP(hipMemcpyAsync(out, nan.data(), nan.size() * sizeof(double), hipMemcpyHostToDevice, st));
        GN_HIP(hipStreamSynchronize(st));
        return GN_OK;
    }
    GN_REQUIRE(pos_score && neg_score, "score pointers are null");
    const Layout l = layout(E, R);
    GN_REQUIRE(workspace && workspace_bytes >= l.total, "workspace too small: need %zu bytes", l.total);
    char* ws = static_cast<char*>(workspace);
    uint64_t *keys = (uint64_t*)(ws + l.keys), *keys_s = (uint64_t*)(ws + l.keys_sorted);
    int32_t *lab = (int32_t*)(ws + l.labels), *lab_s = (int32_t*)(ws + l.labels_sorted);
    int32_t *tp = (int32_t*)(ws + l.tp), *ge = (int32_t*)(ws + l.ge), *prev = (int32_t*)(ws + l.prev);
    double *t0 = (double*)(ws + l.t0), *t1 = (double*)(ws + l.t1), *t2 = (double*)(ws + l.t2);
    int64_t *starts_d = (int64_t*)(ws + l.starts), *offs = (int64_t*)(ws + l.offs);
    size_t tmp_bytes = l.total - l.tmp;
    const size_t n = 2 * (size_t)E;
    GN_HIP(hipMemcpyAsync(starts_d, starts.data(), (R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
    k_metric_keys<<<gn::stream_grid(n, 256), 256, 0, st>>>(pos_score, neg_score, starts_d, (int)R, E, keys, lab);
    GN_LAUNCH_CHECK();
    GN_HIP(rocprim::radix_sort_pairs(ws + l.tmp, tmp_bytes, keys, keys_s, lab, lab_s, n, 0, 32 + bits_for(R), st));
    GN_HIP(rocprim::inclusive_scan(ws + l.tmp, tmp_bytes, lab_s, tp, n, rocprim::plus<int32_t>(), st));
    k_group_end_index<<<gn::stream_grid(n, 256), 256, 0, st>>>(keys_s, (int64_t)n, ge);
    GN_LAUNCH_CHECK();
    GN_HIP(rocprim::exclusive_scan(ws + l.tmp, tmp_bytes, ge, prev, (int32_t)-1, n, MaxOp(), st));
    k_metric_terms<<<gn::stream_grid(n, 256), 256, 0, st>>>(keys_s, tp, ge, prev, starts_d, (int64_t)n, t0, t1, t2);
    GN_LAUNCH_CHECK();
    k_seg_offsets<<<(int)gn::ceil_div(R + 1, 256), 256, 0, st>>>(starts_d, (int)R, offs);
    GN_LAUNCH_CHECK();
    double* terms[3] = {t0, t1, t2};
    for (int k = 0; k < 3; ++k)
        GN_HIP(rocprim::segmented_reduce(ws + l.tmp, tmp_bytes, terms[k], out + k * R, (unsigned)R, offs, offs + 1,
                                         rocprim::plus<double>(), 0.0, st));
    k_nan_empty<<<(int)gn::ceil_div(3 * R, 256), 256, 0, st>>>(starts_d, (int)R, out);
    GN_LAUNCH_CHECK();
    GN_HIP(hipStreamSynchronize(st));        // `starts` (host) was the source of an async copy
    return GN_OK;
}
