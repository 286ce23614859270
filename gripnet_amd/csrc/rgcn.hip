// Multi-relational internal layer (myRGCN, gripnet/layers.py:165-197): the entry point, its choice of kernel, and the
// TABLE path.
//
// Kernels: destination-major in basis space for small supervertices (rgcn_pair.hip), LDS accumulator rows
// (rgcn_fast.hip), and for everything else the O(E)-memory basis-space path of rgcn_basis.hip (GN_RGCN_PATH_GENERAL).
// The table path below (GN_RGCN_PATH_TABLE) is the transform-then-gather ordering: W_r = att[r,:] .
// basis, H_r = X W_r for every relation (two MFMA GEMMs), then one destination-major gather-reduce over the [R*N, out]
// table with the global mean, the root term and the activation in the epilogue.  It works for any shape but needs
// R * N * out floats of workspace: kept for the shapes the basis-space path does not cover (> 64 bases, > 128 input
// features) and as an independent cross-check in the tests.
#include "aggregate.cuh"

// rgcn_fast.hip
bool gn_rgcn_fast_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases);
size_t gn_rgcn_fast_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases);
gn_status gn_rgcn_fast_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin,
                               const float* basis, const float* att, int64_t bases, const float* root,
                               const float* bias, int64_t fout, int relu, int partial, float* out,
                               int64_t ld_out, const gn_side_copy& side, void* ws, size_t ws_bytes, hipStream_t st);

// rgcn_pair.hip
bool gn_rgcn_pair_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases);
gn_status gn_rgcn_pair_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* basis,
                               const float* att, int64_t bases, const float* root, const float* bias, int64_t fout,
                               int relu, int partial, int fast_arith, float* out, int64_t ld_out, const gn_side_copy& side,
                               const void* x_planes, hipStream_t st, int mode, void* pair_sums, int basis_transposed = 0);
size_t gn_rgcn_pair_sums_bytes(const gn_rgcn_plan* plan, int64_t bases);

bool gn_rgcn_fast_finalize_applicable(int64_t fin, int64_t fout, int64_t ld_summed, const void* summed);
gn_status gn_rgcn_fast_finalize(const gn_rgcn_plan* plan, const float* summed, const float* x, int64_t ld_x, int64_t fin,
                                const float* root, const float* bias, int relu, float* out, int64_t ld_out,
                                const gn_side_copy& side, hipStream_t st);

// rgcn_basis.hip
bool gn_rgcn_basis_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases);
size_t gn_rgcn_basis_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases);
gn_status gn_rgcn_basis_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* basis,
                                const float* att, int64_t bases, const float* root, const float* bias, int64_t fout, int relu,
                                int partial, int fast_arith, float* out, int64_t ld_out, const gn_side_copy& side, void* ws,
                                size_t ws_bytes, hipStream_t st);

namespace {

size_t align_up(size_t v) { return (v + 255) & ~size_t(255); }

// Which kernel serves these shapes under these flags (GN_RGCN_PATH_*): the destination-major kernel (three-term bf16
// splits by default, two-term under GN_RGCN_ARITH_FAST), else the LDS-accumulator / general kernels on the fp32 matrix
// instruction.  A forced kernel that does not cover the shapes is not taken.
int select_path(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases, int flags, const void* basis) {
    const int forced = (flags >> GN_RGCN_PATH_SHIFT) & 7;
    const bool pair_ok = gn_rgcn_pair_applicable(plan, fin, fout, bases) && (reinterpret_cast<uintptr_t>(basis) & 15) == 0;
    const bool lds_ok = gn_rgcn_fast_applicable(plan, fin, fout, bases);
    if (forced == GN_RGCN_PATH_PAIR && pair_ok) return GN_RGCN_PATH_PAIR;
    if (forced == GN_RGCN_PATH_LDS && lds_ok) return GN_RGCN_PATH_LDS;
    const bool basis_ok = gn_rgcn_basis_applicable(plan, fin, fout, bases);
    if (forced == GN_RGCN_PATH_TABLE) return GN_RGCN_PATH_TABLE;
    if (forced == GN_RGCN_PATH_GENERAL) return basis_ok ? GN_RGCN_PATH_GENERAL : GN_RGCN_PATH_TABLE;
    if (pair_ok) return GN_RGCN_PATH_PAIR;
    if (lds_ok) return GN_RGCN_PATH_LDS;
    return basis_ok ? GN_RGCN_PATH_GENERAL : GN_RGCN_PATH_TABLE;
}

struct GeneralWs {
    size_t w_off, h_off, xr_off, total;
};

GeneralWs general_layout(const gn_rgcn_plan* plan, int64_t fin, int64_t fout) {
    GeneralWs l;
    l.w_off = 0;
    l.h_off = align_up((size_t)plan->num_relations * fin * fout * sizeof(float));
    l.xr_off = l.h_off + align_up((size_t)plan->num_relations * plan->num_nodes * fout * sizeof(float));
    l.total = l.xr_off + align_up((size_t)plan->num_nodes * fout * sizeof(float));
    return l;
}

__global__ void k_rgcn_finalize(const float* __restrict__ summed, int64_t ld_s, const float* __restrict__ indeg,
                                int relu, float* __restrict__ out, int64_t ld_o, int64_t rows, int cols,
                                gn_side_copy side) {
    if (side.dst) {                                            // concat slot 0
        const int64_t stotal = side.rows * side.cols;
        for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < stotal; t += (int64_t)gridDim.x * blockDim.x) {
            const int64_t i = t / side.cols, c = t - i * side.cols;
            const float v = side.src[i * side.ld_src + c];
            side.dst[i * side.ld_dst + c] = side.mode ? fabsf(v) : v;
        }
    }
    const int64_t total = rows * cols;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / cols;
        const int c = (int)(t - i * cols);
        float v = summed[i * ld_s + c] / fmaxf(indeg[i], 1.0f) + out[i * ld_o + c];   // out holds x.root + bias
        if (relu) v = fmaxf(v, 0.f);
        out[i * ld_o + c] = v;
    }
}

}  // namespace

extern "C" {

// Scratch of ONE kernel for these shapes: what a call that takes `path` writes.
static size_t path_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases, int path, int flags = 0) {
    if (path == GN_RGCN_PATH_PAIR) return (flags & (GN_RGCN_PAIR_SUMS_ONLY | GN_RGCN_PAIR_SUMS_READY)) ? gn_rgcn_pair_sums_bytes(plan, bases) : 0;
    if (path == GN_RGCN_PATH_LDS) return gn_rgcn_fast_workspace_bytes(plan, fin, fout, bases);
    if (path == GN_RGCN_PATH_GENERAL) return gn_rgcn_basis_workspace_bytes(plan, fin, fout, bases);
    return general_layout(plan, fin, fout).total;
}

size_t gn_rgcn_workspace_bytes(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases, int flags) {
    if (!plan || fin <= 0 || fout <= 0 || bases <= 0) return 0;
    // of the kernel a forward with these flags takes when its `basis` is 16-byte aligned (torch allocations are); a
    // call that ends up on another kernel - an unaligned basis - is refused with the size it needs, never under-served
    return path_workspace_bytes(plan, fin, fout, bases, select_path(plan, fin, fout, bases, flags, nullptr), flags);
}

int gn_rgcn_forward_path(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases, int flags) {
    if (!plan || fin <= 0 || fout <= 0 || bases <= 0) return -1;
    return select_path(plan, fin, fout, bases, flags, nullptr);
}

gn_status gn_rgcn_forward_f32(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin,
                              const float* basis, const float* att, int64_t bases, const float* root,
                              const float* bias, int64_t fout, int relu, int flags, float* out, int64_t ld_out,
                              const gn_side_copy* side, const void* x_planes, void* workspace, size_t workspace_bytes,
                              void* stream) {
    const int partial = flags & GN_RGCN_PARTIAL;
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(fin > 0 && fout > 0 && bases > 0, "feature / basis counts must be positive");
    const int64_t N = plan->num_nodes, R = plan->num_relations;
    if (N == 0) return GN_OK;
    const int split = flags & (GN_RGCN_PAIR_SUMS_ONLY | GN_RGCN_PAIR_SUMS_READY);
    GN_REQUIRE(split != (GN_RGCN_PAIR_SUMS_ONLY | GN_RGCN_PAIR_SUMS_READY), "GN_RGCN_PAIR_SUMS_ONLY and _READY exclude each other");
    if (split == GN_RGCN_PAIR_SUMS_ONLY) {
        // the x-independent half on its own: att and the plan are all it reads (x, basis, root, out may be NULL)
        GN_REQUIRE(att != nullptr, "att is null");
        if (select_path(plan, fin, fout, bases, flags, nullptr) != GN_RGCN_PATH_PAIR)
            return gn::fail(GN_ERR_UNSUPPORTED, "pair sums: these shapes do not take the destination-major kernel");
        const size_t need = gn_rgcn_pair_sums_bytes(plan, bases);
        GN_REQUIRE(workspace && workspace_bytes >= need, "workspace too small: the pair sums need %zu bytes, got %zu", need, workspace_bytes);
        gn_side_copy none = {nullptr, 0, nullptr, 0, 0, 0, 0};
        return gn_rgcn_pair_forward(plan, nullptr, fin, fin, nullptr, att, bases, nullptr, nullptr, fout, 0, 0, 0, nullptr, fout, none,
                                    nullptr, gn::as_stream(stream), 1, workspace);
    }
    GN_REQUIRE(x && basis && att && out && (partial || root), "operand pointer is null");
    GN_REQUIRE(ld_x >= fin && ld_out >= fout, "leading dimension smaller than the row length");
    // the workspace is checked against the kernel THIS call takes (a forced general or table path, or a basis pointer the
    // destination-major kernel cannot use, needs its own scratch whatever gn_rgcn_workspace_bytes said for the
    // default choice)
    const int path = select_path(plan, fin, fout, bases, flags, basis);
    if (split && (path != GN_RGCN_PATH_PAIR || !x_planes || (flags & GN_RGCN_ARITH_FAST)))
        return gn::fail(GN_ERR_UNSUPPORTED, "GN_RGCN_PAIR_SUMS_READY: the destination-major kernel with x as split planes, default arithmetic");
    if ((flags & GN_RGCN_BASIS_TRANSPOSED) && path != GN_RGCN_PATH_PAIR)
        return gn::fail(GN_ERR_UNSUPPORTED, "GN_RGCN_BASIS_TRANSPOSED: only the destination-major kernel reads a transposed basis (pass a transposed copy)");
    const size_t need = path_workspace_bytes(plan, fin, fout, bases, path, flags);
    GN_REQUIRE(need == 0 || (workspace && workspace_bytes >= need), "workspace too small: this call needs %zu bytes, got %zu",
               need, workspace_bytes);
    hipStream_t st = gn::as_stream(stream);
    gn_side_copy sc;
    gn_status ss = gn::check_side(side, N, &sc);
    if (ss != GN_OK) return ss;

    if (path == GN_RGCN_PATH_PAIR)
        return gn_rgcn_pair_forward(plan, x, ld_x, fin, basis, att, bases, root, bias, fout, relu, partial,
                                    flags & GN_RGCN_ARITH_FAST, out, ld_out, sc, x_planes, st, split ? 2 : 0, split ? workspace : nullptr,
                                    flags & GN_RGCN_BASIS_TRANSPOSED);
    if (path == GN_RGCN_PATH_LDS)
        return gn_rgcn_fast_forward(plan, x, ld_x, fin, basis, att, bases, root, bias, fout, relu, partial,
                                    out, ld_out, sc, workspace, workspace_bytes, st);
    if (path == GN_RGCN_PATH_GENERAL)
        return gn_rgcn_basis_forward(plan, x, ld_x, fin, basis, att, bases, root, bias, fout, relu, partial,
                                     flags & GN_RGCN_ARITH_FAST, out, ld_out, sc, workspace, workspace_bytes, st);

    const GeneralWs l = general_layout(plan, fin, fout);
    char* ws = static_cast<char*>(workspace);
    float* W = reinterpret_cast<float*>(ws + l.w_off);
    float* H = reinterpret_cast<float*>(ws + l.h_off);
    float* XR = reinterpret_cast<float*>(ws + l.xr_off);
    gn_status s;
    if (R > 0) {
        // K7: W[R, fin*fout] = att[R,B] @ basis[B, fin*fout]   (layers.py:172-173)
        s = gn_gemm_f32(att, bases, 0, nullptr, 0, basis, fin * fout, 0, W, fin * fout, 0, R, fin * fout, bases, 1, nullptr, 0, stream);
        if (s != GN_OK) return s;
        // H[r] = X @ W[r] for all relations; grid.z carries the relation, in slabs of 32768
        for (int64_t r0 = 0; r0 < R; r0 += 32768) {
            const int64_t nb = std::min<int64_t>(32768, R - r0);
            s = gn_gemm_f32(x, ld_x, 0, nullptr, 0, W + r0 * fin * fout, fout, fin * fout, H + r0 * N * fout, fout,
                            N * fout, N, fout, fin, nb, nullptr, 0, stream);
            if (s != GN_OK) return s;
        }
    }
    if (!partial) {
        // K12: x @ root (+ bias) goes in as the addend of the gather epilogue (layers.py:193-196)
        s = gn_gemm_f32(x, ld_x, 0, nullptr, 0, root, fout, 0, XR, fout, 0, N, fout, fin, 1, bias, 0, stream);
        if (s != GN_OK) return s;
    }
    gn::AggArgs a;
    a.rowptr = plan->rowptr.p;
    a.col = plan->key.p;
    a.coef = nullptr;
    a.table = H;
    a.ld_table = fout;
    a.features = (int)fout;
    a.rowdiv = partial ? nullptr : plan->indeg.p;
    a.addend = partial ? nullptr : XR;
    a.ld_addend = fout;
    a.bias = nullptr;
    a.relu = partial ? 0 : relu;
    a.out = out;
    a.ld_out = ld_out;
    a.rows = (int)N;
    a.side = sc;
    return gn::launch_aggregate(a, st);
}

gn_status gn_rgcn_finalize_f32(const gn_rgcn_plan* plan, const float* summed, int64_t ld_summed, const float* x,
                               int64_t ld_x, int64_t fin, const float* root, const float* bias, int64_t fout,
                               int relu, float* out, int64_t ld_out, const gn_side_copy* side, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(fin > 0 && fout > 0 && fout < (1ll << 31), "feature counts must be positive");
    const int64_t N = plan->num_nodes;
    if (N == 0) return GN_OK;
    GN_REQUIRE(summed && x && root && out, "operand pointer is null");
    GN_REQUIRE(summed != out, "finalize cannot run in place");
    gn_side_copy sc;
    gn_status ss = gn::check_side(side, N, &sc);
    if (ss != GN_OK) return ss;
    if (gn_rgcn_fast_finalize_applicable(fin, fout, ld_summed, summed))
        return gn_rgcn_fast_finalize(plan, summed, x, ld_x, fin, root, bias, relu, out, ld_out, sc, gn::as_stream(stream));
    gn_status s = gn_gemm_f32(x, ld_x, 0, nullptr, 0, root, fout, 0, out, ld_out, 0, N, fout, fin, 1, bias, 0, stream);
    if (s != GN_OK) return s;
    k_rgcn_finalize<<<gn::stream_grid(N * fout, 256), 256, 0, gn::as_stream(stream)>>>(
        summed, ld_summed, plan->indeg.p, relu, out, ld_out, N, (int)fout, sc);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // extern "C"
