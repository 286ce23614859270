// Multi-relational internal layer (myRGCN, gripnet/layers.py:165-197) for LARGE supervertices: O(E) memory, basis space.
//
//   out[i] * deg_i = sum_{e: dst=i} x[src_e] W_{r(e)},   W_r = sum_b att[r,b] basis[b]                (layers.py:172-189)
//                  = sum_b U_i[b,:] basis[b],             U_i[b,:] = sum_{e: dst=i} att[r(e),b] x[src_e,:]
//
// The reference forms W_r and multiplies per relation slice (O(E) memory); the table form H[r] = X W_r of rgcn.hip's
// first general path needs [R, N, out] floats (2.4 GB and 80 GFLOP per layer for the all-nodes baseline
// baselines/LP_baselines/rgcn_pose.py:53-77 with N = 19,726, R = 964).  Here neither W_r nor H exists: a wave owns a
// destination row; per FOUR incoming edges it issues the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 with
// A = att[r(e), bases] (one float per lane: M = bases) and B = x[src_e, features] (N = features), K = the four edges,
// so that U_i (bases x in) accumulates in registers over the row's edges - every product and sum is an fp32 FMA, the
// same arithmetic as the reference's fp32 matmuls.  The row leaves as one line  [ U_i / max(1, deg_i) | x_i | 0 ]  of a
// slab of rows, and ONE dense product per slab with [basis ; root] (gn_gemm_f32: K = (B + 1) in, fp32-faithful) adds
// the root term, the bias and the activation.  Workspace: the slab (<= 64 MB) + the stacked weights, independent of
// R and N; the edges come from the plan's destination-major list (key = relation * N + source), 4 bytes per edge.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBasisThreads = 256;                 // four waves, a destination row per wave at a time
constexpr int64_t kSlabBytes = 64ll << 20;         // rows of U in flight between the gather and the dense product

struct BasisArgs {
    const float* x; int64_t ld_x; int fin;
    const float* att; int bases;
    const int32_t* rowptr; const uint32_t* key; const float* indeg; const int32_t* order;
    uint32_t n;                                    // nodes (key = relation * n + source)
    int row_lo, row_hi;                            // this launch's slab of destination rows
    float* u; int64_t ld_u;                        // [row_hi - row_lo][ld_u]: U_i (bases outermost) | x_i | zeros up to kp
    int kp;
    int scale, with_x, vec;
    gn_side_copy side;
};

template <int BT, int NT>
__global__ __launch_bounds__(kBasisThreads) void k_rgcn_basis(BasisArgs a) {
    if (a.side.dst) {                                                          // concat slot 0 (layers.py:264-266), by the whole grid
        const int64_t total = a.side.rows * a.side.cols;
        for (int64_t t = (int64_t)blockIdx.x * kBasisThreads + threadIdx.x; t < total; t += (int64_t)gridDim.x * kBasisThreads) {
            const int64_t i = t / a.side.cols, cc = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + cc];
            a.side.dst[i * a.side.ld_dst + cc] = a.side.mode ? fabsf(v) : v;
        }
    }
    const int lane = threadIdx.x & 63, c = lane & 15, kg = lane >> 4;
    const int wave = blockIdx.x * (kBasisThreads / 64) + (threadIdx.x >> 6), W = gridDim.x * (kBasisThreads / 64);
    const int n_rows = (int)a.n;
    // rows in the plan's order (by in-degree, largest first), dealt to the waves in a snake: wave w takes entries
    // w, 2W-1-w, 2W+w, ... - every wave gets one row of every round, the heavy end alternating
    for (int base = 0, round = 0; base < n_rows; base += W, ++round) {
        const int idx = base + ((round & 1) ? W - 1 - wave : wave);
        if (idx >= n_rows) continue;
        const int row = a.order[idx];
        if (row < a.row_lo || row >= a.row_hi) continue;
        const int e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
        f32x4 acc[BT][NT];
#pragma unroll
        for (int jm = 0; jm < BT; ++jm)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[jm][t] = (f32x4)(0.f);
        for (int eb = e0; eb < e1; eb += 64) {
            const int cnt = min(64, e1 - eb);
            const uint32_t k = lane < cnt ? a.key[eb + lane] : 0u;
            const uint32_t rel = k / a.n, src = k - rel * a.n;                 // (one division per lane and 64 edges)
            const int steps = (cnt + 3) >> 2;
            for (int j0 = 0; j0 < steps; j0 += 4) {
                // four steps' operands are requested together (the other waves of the SIMD cover the round trips)
                float av[4][BT], xv[4][NT];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sel = 4 * (j0 + q) + kg;                          // edge of lane group kg in step j0 + q
                    const uint32_t s = (uint32_t)__shfl((int)src, sel & 63), r = (uint32_t)__shfl((int)rel, sel & 63);
                    const bool live = sel < cnt;
                    const float* __restrict__ xr = a.x + (size_t)s * a.ld_x + NT * c;
                    const float* __restrict__ ar = a.att + (size_t)r * a.bases + c;
                    if (NT % 4 == 0 && a.vec) {
#pragma unroll
                        for (int t4 = 0; t4 < NT / 4; ++t4) {
                            const f32x4 v = live ? *reinterpret_cast<const f32x4*>(xr + 4 * t4) : (f32x4)(0.f);
                            xv[q][4 * t4] = v[0]; xv[q][4 * t4 + 1] = v[1]; xv[q][4 * t4 + 2] = v[2]; xv[q][4 * t4 + 3] = v[3];
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < NT; ++t) xv[q][t] = (live && NT * c + t < a.fin) ? xr[t] : 0.f;
                    }
#pragma unroll
                    for (int jm = 0; jm < BT; ++jm) av[q][jm] = (live && 16 * jm + c < a.bases) ? ar[16 * jm] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int jm = 0; jm < BT; ++jm)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[jm][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q][jm], xv[q][t], acc[jm][t], 0, 0, 0);
            }
        }
        // acc[jm][t][i] = U_i[base 16 jm + 4 kg + i][feature NT c + t]
        const float inv = a.scale ? 1.0f / fmaxf(a.indeg[row], 1.0f) : 1.0f;
        float* __restrict__ ur = a.u + (size_t)(row - a.row_lo) * a.ld_u;
#pragma unroll
        for (int jm = 0; jm < BT; ++jm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int b = 16 * jm + 4 * kg + i;
                if (b >= a.bases) continue;
                float* __restrict__ q = ur + (size_t)b * a.fin + NT * c;
                if (NT % 4 == 0 && a.vec) {
#pragma unroll
                    for (int t4 = 0; t4 < NT / 4; ++t4)
                        *reinterpret_cast<f32x4*>(q + 4 * t4) = (f32x4){acc[jm][4 * t4][i] * inv, acc[jm][4 * t4 + 1][i] * inv,
                                                                        acc[jm][4 * t4 + 2][i] * inv, acc[jm][4 * t4 + 3][i] * inv};
                } else {
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        if (NT * c + t < a.fin) q[t] = acc[jm][t][i] * inv;
                }
            }
        const int tail = a.bases * a.fin;
        if (a.with_x)
            for (int col = lane; col < a.fin; col += 64) ur[tail + col] = a.x[(size_t)row * a.ld_x + col];
        for (int col = tail + (a.with_x ? a.fin : 0) + lane; col < a.kp; col += 64) ur[col] = 0.f;
    }
}

// [basis ; root ; 0]: the right-hand side of the slab's dense product
__global__ void k_basis_weights(const float* __restrict__ basis, const float* __restrict__ root, int64_t nb, int64_t nr, int64_t total,
                                float* __restrict__ w) {
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x)
        w[t] = t < nb ? basis[t] : (t < nb + nr ? root[t - nb] : 0.f);
}

struct BasisLayout {
    int kp;
    int64_t slab_rows;
    size_t w_off, u_off, total;
};

BasisLayout basis_layout(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    BasisLayout l;
    l.kp = (int)(gn::ceil_div((bases + 1) * fin, 32) * 32);
    l.slab_rows = std::min<int64_t>(std::max<int64_t>(plan->num_nodes, 1), std::max<int64_t>(256, kSlabBytes / ((int64_t)l.kp * 4)));
    l.w_off = 0;
    l.u_off = ((size_t)l.kp * fout * sizeof(float) + 255) & ~size_t(255);
    l.total = l.u_off + (size_t)l.slab_rows * l.kp * sizeof(float);
    return l;
}

template <int BT>
gn_status launch_bt(int nt, const BasisArgs& a, int grid, hipStream_t st) {
#define GN_BASIS_CASE(N) case N: if constexpr (BT * N <= 16) { k_rgcn_basis<BT, N><<<grid, kBasisThreads, 0, st>>>(a); break; } else return gn::fail(GN_ERR_UNSUPPORTED, "unsupported shape")
    switch (nt) {
        GN_BASIS_CASE(1); GN_BASIS_CASE(2); GN_BASIS_CASE(3); GN_BASIS_CASE(4);
        GN_BASIS_CASE(5); GN_BASIS_CASE(6); GN_BASIS_CASE(7); GN_BASIS_CASE(8);
        default: return gn::fail(GN_ERR_UNSUPPORTED, "unsupported shape");
    }
#undef GN_BASIS_CASE
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

// ---- called from rgcn.hip ----
bool gn_rgcn_basis_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    if (!plan || !plan->row_order.p || fin < 1 || fin > 128 || bases < 1 || bases > 64 || fout < 1) return false;
    const int64_t bt = gn::ceil_div(bases, 16), nt = gn::ceil_div(fin, 16);
    return bt * nt <= 16;
}

size_t gn_rgcn_basis_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    return basis_layout(plan, fin, fout, bases).total;
}

gn_status gn_rgcn_basis_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* basis,
                                const float* att, int64_t bases, const float* root, const float* bias, int64_t fout, int relu,
                                int partial, int fast_arith, float* out, int64_t ld_out, const gn_side_copy& side, void* ws,
                                size_t ws_bytes, hipStream_t st) {
    const BasisLayout l = basis_layout(plan, fin, fout, bases);
    GN_REQUIRE(ws && ws_bytes >= l.total, "workspace too small: the basis-space path needs %zu bytes, got %zu", l.total, ws_bytes);
    const int64_t N = plan->num_nodes;
    float* W = reinterpret_cast<float*>(static_cast<char*>(ws) + l.w_off);
    float* U = reinterpret_cast<float*>(static_cast<char*>(ws) + l.u_off);
    const int64_t nb = bases * fin * fout, nr = partial ? 0 : fin * fout, total = (int64_t)l.kp * fout;
    k_basis_weights<<<gn::stream_grid(total, 256), 256, 0, st>>>(basis, root, nb, nr, total, W);
    GN_LAUNCH_CHECK();
    BasisArgs a;
    a.x = x; a.ld_x = ld_x; a.fin = (int)fin; a.att = att; a.bases = (int)bases;
    a.rowptr = plan->rowptr.p; a.key = plan->key.p; a.indeg = plan->indeg.p; a.order = plan->row_order.p;
    a.n = (uint32_t)N; a.u = U; a.ld_u = l.kp; a.kp = l.kp;
    a.scale = partial ? 0 : 1; a.with_x = partial ? 0 : 1;
    a.vec = (fin % 16 == 0) && (ld_x % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int bt = (int)gn::ceil_div(bases, 16), nt = (int)gn::ceil_div(fin, 16);
    // four waves per SIMD on every compute unit (<= 128 registers), a row per wave at a time
    const int grid = (int)std::min<int64_t>((int64_t)gn::compute_units() * 4, gn::ceil_div(N, kBasisThreads / 64));
    for (int64_t r0 = 0; r0 < N; r0 += l.slab_rows) {
        const int64_t r1 = std::min(N, r0 + l.slab_rows);
        a.row_lo = (int)r0; a.row_hi = (int)r1;
        a.side = r0 == 0 ? side : gn_side_copy{nullptr, 0, nullptr, 0, 0, 0, 0};
        gn_status s;
        switch (bt) {
            case 1: s = launch_bt<1>(nt, a, grid, st); break;
            case 2: s = launch_bt<2>(nt, a, grid, st); break;
            case 3: s = launch_bt<3>(nt, a, grid, st); break;
            default: s = launch_bt<4>(nt, a, grid, st); break;
        }
        if (s != GN_OK) return s;
        s = gn_gemm_f32(U, l.kp, 0, nullptr, 0, W, fout, 0, out + r0 * ld_out, ld_out, 0, r1 - r0, fout, l.kp, 1,
                        partial ? nullptr : bias, ((relu && !partial) ? GN_GEMM_RELU : 0) | (fast_arith ? GN_GEMM_ARITH_FAST : 0), st);
        if (s != GN_OK) return s;
    }
    return GN_OK;
}
