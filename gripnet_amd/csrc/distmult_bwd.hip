// Backward of the DistMult decoder (autograd of multiRelaInnerProductDecoder.forward,
// gripnet/decoder.py:19-23, as used by the loss of GripNet-pose.py:140-146):
//
//   s_e = sum_k z[u_e,k] z[v_e,k] D[r_e,k]        gs_e = d loss / d s_e   (the caller folds the sigmoid in)
//   dz[u_e,:] += gs_e * z[v_e,:] * D[r_e,:]       dz[v_e,:] += gs_e * z[u_e,:] * D[r_e,:]
//   dD[r_e,:] += gs_e * z[u_e,:] * z[v_e,:]
//
// The scatter targets are tiny (n x F and R x F) and hit millions of times, so they are privatised:
// per column phase of 16 features a persistent workgroup keeps the z columns and a dz accumulator in
// LDS (2 x n x 64 B), walks its contiguous edge range with 16 lanes per edge (one lane per column,
// four edges per wave step), accumulates dz with LDS float atomics, keeps the dD row of the current
// relation in a register while the relation id does not change (type-sorted edge lists), and adds its
// LDS accumulator to dz with one global atomic per element at the end of the phase.  Float atomics
// make the summation order, hence the last bits of the gradients, vary from run to run.
// Larger node tables take the general kernel (global atomics per edge).
#include "common.h"

namespace {

constexpr int kCw = 16;                       // columns per phase = lanes per edge
constexpr int kThreads = 1024;
constexpr size_t kLdsBudget = 158 * 1024;

struct BwdArgs {
    const float* __restrict__ z; int64_t ld_z; int n; int features;
    const int64_t* __restrict__ u; const int64_t* __restrict__ v; const int64_t* __restrict__ et;
    const float* __restrict__ d; int64_t ld_d; int r;
    const float* __restrict__ gs; int64_t e; int64_t edges_per_wg;
    float* dz; int64_t ld_dz; float* dd; int64_t ld_dd;
    float* slabs;                     // LDS path: [groups][n][features] per-workgroup dz partials
};

template <bool LDS_TABLE>
__global__ __launch_bounds__(kThreads) void k_distmult_bwd(BwdArgs a) {
    extern __shared__ float lds[];
    float* zp = lds;                                   // [n][kCw]   (LDS_TABLE only)
    float* dzp = lds + (size_t)a.n * kCw;              // [n][kCw]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & (kCw - 1), grp = lane >> 4;   // column inside the phase, edge slot inside the step
    const int64_t wg_lo = (int64_t)blockIdx.x * a.edges_per_wg;
    const int64_t wg_hi = min(a.e, wg_lo + a.edges_per_wg);
    constexpr int64_t kStride = (kThreads / 64) * 64;

    for (int c0 = 0; c0 < a.features; c0 += kCw) {
        const bool col_ok = c0 + c < a.features;
        if constexpr (LDS_TABLE) {
            __syncthreads();
            for (int i = tid; i < a.n * kCw; i += kThreads) {
                const int row = i / kCw, cc = i % kCw;
                zp[i] = c0 + cc < a.features ? a.z[(int64_t)row * a.ld_z + c0 + cc] : 0.f;
                dzp[i] = 0.f;
            }
            __syncthreads();
        }
        int cur_r = -1;                                  // relation whose dD partial sits in racc (per 16-lane group)
        float racc = 0.f, dreg = 0.f;
        for (int64_t e0 = wg_lo + wave * 64; e0 < wg_hi; e0 += kStride) {
            const int64_t mine = e0 + lane;
            int iu = 0, iv = 0, ir = 0;
            float g = 0.f;
            if (mine < wg_hi) {
                const int64_t uu = a.u[mine], vv = a.v[mine], rr = a.et[mine];
                const bool ok = (uint64_t)uu < (uint64_t)a.n && (uint64_t)vv < (uint64_t)a.n && (uint64_t)rr < (uint64_t)a.r;
                if (ok) { iu = (int)uu; iv = (int)vv; ir = (int)rr; g = a.gs[mine]; }   // out-of-table edges contribute nothing
            }
            const int cnt = (int)min((int64_t)64, wg_hi - e0);
            for (int t = 0; t * 4 < cnt; ++t) {
                const int srcl = t * 4 + grp;
                const int eu = __shfl(iu, srcl), ev = __shfl(iv, srcl), er = __shfl(ir, srcl);
                const float eg = __shfl(g, srcl);
                if (srcl < cnt && col_ok) {
                    if (er != cur_r) {                   // uniform inside the 16-lane group
                        if (cur_r >= 0) atomicAdd(&a.dd[(int64_t)cur_r * a.ld_dd + c0 + c], racc);
                        cur_r = er;
                        racc = 0.f;
                        dreg = a.d[(int64_t)er * a.ld_d + c0 + c];
                    }
                    float zu, zv;
                    if constexpr (LDS_TABLE) {
                        zu = zp[eu * kCw + c];
                        zv = zp[ev * kCw + c];
                    } else {
                        zu = a.z[(int64_t)eu * a.ld_z + c0 + c];
                        zv = a.z[(int64_t)ev * a.ld_z + c0 + c];
                    }
                    const float w = eg * dreg;
                    if constexpr (LDS_TABLE) {
                        atomicAdd(&dzp[eu * kCw + c], w * zv);
                        atomicAdd(&dzp[ev * kCw + c], w * zu);
                    } else {
                        atomicAdd(&a.dz[(int64_t)eu * a.ld_dz + c0 + c], w * zv);
                        atomicAdd(&a.dz[(int64_t)ev * a.ld_dz + c0 + c], w * zu);
                    }
                    racc += eg * zu * zv;
                }
            }
        }
        if (cur_r >= 0 && col_ok) atomicAdd(&a.dd[(int64_t)cur_r * a.ld_dd + c0 + c], racc);
        if constexpr (LDS_TABLE) {
            // the accumulator leaves as a slab (plain stores): 64-byte pieces in different rows are the slow
            // shape for global float atomics, and a fixed-order reduction kernel follows anyway
            __syncthreads();
            float* slab = a.slabs + (size_t)blockIdx.x * a.n * a.features;
            for (int i = tid; i < a.n * kCw; i += kThreads) {
                const int row = i / kCw, cc = i % kCw;
                if (c0 + cc < a.features) slab[(size_t)row * a.features + c0 + cc] = dzp[i];
            }
        }
    }
}

// dz[row, c] = sum over workgroups of slab[g][row][c], fixed order
__global__ void k_distmult_bwd_reduce(const float* __restrict__ slabs, int groups, int64_t total, int features,
                                      float* __restrict__ dz, int64_t ld_dz) {
    const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (t >= total) return;
    float s = 0.f;
#pragma unroll 8
    for (int g = 0; g < groups; ++g) s += slabs[(size_t)g * total + t];
    const int64_t row = t / features;
    dz[row * ld_dz + (t - row * features)] = s;
}

int64_t bwd_groups(int64_t e, int64_t* edges_per_wg) {
    int64_t groups = std::min<int64_t>(256, gn::ceil_div(e, 64 * (kThreads / 64)));
    if (groups < 1) groups = 1;
    *edges_per_wg = gn::ceil_div(gn::ceil_div(e, groups), 64) * 64;
    return gn::ceil_div(e, *edges_per_wg);
}

bool bwd_lds_path(int64_t n) { return (size_t)n * kCw * 2 * sizeof(float) <= kLdsBudget && !gn::fast_paths_disabled(); }

}  // namespace

extern "C" size_t gn_distmult_backward_workspace_bytes(int64_t n, int64_t f, int64_t e) {
    if (n <= 0 || f <= 0 || e <= 0 || !bwd_lds_path(n)) return 0;
    int64_t per = 0;
    return (size_t)bwd_groups(e, &per) * n * f * sizeof(float);
}

extern "C" gn_status gn_distmult_backward_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                              const int64_t* v, const int64_t* et, const float* d, int64_t ld_d,
                                              int64_t r, int64_t e, const float* grad_logit, float* dz, int64_t ld_dz,
                                              float* dd, int64_t ld_dd, void* workspace, size_t workspace_bytes,
                                              void* stream) {
    GN_REQUIRE(n >= 0 && f >= 0 && r >= 0 && e >= 0, "negative size");
    GN_REQUIRE(f < (1ll << 31) && n < (1ll << 31) && r < (1ll << 31), "table too large");
    GN_REQUIRE((n == 0 || f == 0 || dz) && (r == 0 || f == 0 || dd), "gradient output pointer is null");
    GN_REQUIRE(ld_dz >= f && ld_dd >= f, "leading dimension smaller than the row length");
    hipStream_t st = gn::as_stream(stream);
    if (n > 0 && f > 0) GN_HIP(hipMemset2DAsync(dz, ld_dz * sizeof(float), 0, f * sizeof(float), n, st));
    if (r > 0 && f > 0) GN_HIP(hipMemset2DAsync(dd, ld_dd * sizeof(float), 0, f * sizeof(float), r, st));
    if (e == 0 || f == 0) return GN_OK;
    GN_REQUIRE(n > 0 && r > 0, "edges given but the node or relation table is empty");
    GN_REQUIRE(z && u && v && et && d && grad_logit, "operand pointer is null");
    GN_REQUIRE(ld_z >= f && ld_d >= f, "leading dimension smaller than the row length");
    BwdArgs a;
    a.z = z; a.ld_z = ld_z; a.n = (int)n; a.features = (int)f; a.u = u; a.v = v; a.et = et; a.d = d; a.ld_d = ld_d;
    a.r = (int)r; a.gs = grad_logit; a.e = e; a.dz = dz; a.ld_dz = ld_dz; a.dd = dd; a.ld_dd = ld_dd;
    const int64_t groups = bwd_groups(e, &a.edges_per_wg);
    const size_t lds_bytes = (size_t)n * kCw * 2 * sizeof(float);
    a.slabs = static_cast<float*>(workspace);
    if (bwd_lds_path(n)) {
        GN_REQUIRE(workspace && workspace_bytes >= gn_distmult_backward_workspace_bytes(n, f, e),
                   "workspace too small: need %zu bytes", gn_distmult_backward_workspace_bytes(n, f, e));
        static thread_local bool configured = false;
        if (!configured) {
            GN_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_distmult_bwd<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            configured = true;
        }
        k_distmult_bwd<true><<<(unsigned)groups, kThreads, lds_bytes, st>>>(a);
        GN_LAUNCH_CHECK();
        k_distmult_bwd_reduce<<<(unsigned)gn::ceil_div(n * f, 256), 256, 0, st>>>(a.slabs, (int)groups, n * f, (int)f, dz,
                                                                               ld_dz);
    } else {
        k_distmult_bwd<false><<<(unsigned)groups, kThreads, 0, st>>>(a);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}
