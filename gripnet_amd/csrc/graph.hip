// GCN-style aggregation over a cached plan (square graphs with self loops, or the bipartite
// external layer).  Replaces propagate/message/update of myGCN (gripnet/layers.py:92-100).
#include "aggregate.cuh"

// gcn_blocked.hip: the LDS-staged, source-blocked path
bool gn_blocked_applicable(const gn_graph_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* w, int64_t fout);
gn_status gn_blocked_aggregate(const gn_graph_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* w,
                               int64_t fout, const float* bias, int relu, float* out, int64_t ld_out,
                               const gn_side_copy& side, hipStream_t st);

extern "C" gn_status gn_graph_aggregate_f32(const gn_graph_plan* plan, const float* xw, int64_t ld_xw,
                                            int64_t num_features, const float* weight, int64_t out_features,
                                            const float* bias, int relu, float* out, int64_t ld_out,
                                            const gn_side_copy* side, const gn_split_planes* planes, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(num_features >= 0 && num_features < (1ll << 31), "bad feature count");
    if (plan->rows == 0 || num_features == 0) return GN_OK;
    GN_REQUIRE(xw && out, "feature pointers are null");
    const int64_t width = weight ? out_features : num_features;
    GN_REQUIRE(ld_xw >= num_features && ld_out >= width, "leading dimension smaller than the row length");
    gn::AggArgs a;
    a.rowptr = plan->rowptr.p;
    a.col = reinterpret_cast<const uint32_t*>(plan->col.p);
    a.coef = plan->plain_ones ? nullptr : plan->coef.p;       // (null = all ones)
    a.table = xw;
    a.ld_table = ld_xw;
    a.features = (int)num_features;
    a.rowdiv = nullptr;
    a.addend = nullptr;
    a.ld_addend = 0;
    a.bias = bias;
    a.relu = relu;
    a.out = out;
    a.ld_out = ld_out;
    a.rows = (int)plan->rows;
    a.nnz = plan->nnz;
    a.table_rows = plan->table_rows;
    if (plan->ell_ok) { a.ell_col = plan->ell_col.p; a.ell_coef = plan->ell_coef.p; }
    gn_status ss = gn::check_side(side, plan->rows, &a.side);
    if (ss != GN_OK) return ss;
    if (planes && planes->planes) {
        GN_REQUIRE(planes->nt >= 1 && planes->nt <= 4 && planes->rows >= plan->rows && planes->col_main >= 0 &&
                       planes->col_main + width <= 16 * planes->nt &&
                       (!a.side.dst || (planes->col_side >= 0 && planes->col_side + a.side.cols <= 16 * planes->nt)),
                   "split planes do not hold the launch's columns");
    }
    // the kernels that do not write the planes themselves are followed by the stand-alone split of what they wrote
    auto then_split = [&](gn_status s) {
        if (s != GN_OK || !planes || !planes->planes) return s;
        s = gn_split_planes_f32(out, ld_out, plan->rows, width, planes->col_main, planes, stream);
        if (s == GN_OK && a.side.dst)
            s = gn_split_planes_f32(a.side.dst, a.side.ld_dst, a.side.rows, a.side.cols, planes->col_side, planes, stream);
        return s;
    };
    if (gn_blocked_applicable(plan, xw, ld_xw, num_features, weight, width))
        return then_split(gn_blocked_aggregate(plan, xw, ld_xw, num_features, weight, width, bias, relu, out, ld_out, a.side,
                                               gn::as_stream(stream)));
    if (weight && gn::mfma_fusable(num_features, out_features, plan->rows, plan->nnz) && (ld_xw % 4) == 0 && gn::aligned16(xw))
        return then_split(gn::launch_aggregate_mfma(a, weight, (int)out_features, gn::as_stream(stream)));   // wide layers: W on the matrix cores
    if (weight) {            // aggregate the input rows, then contract with W in the epilogue
        if (!gn::transform_fusable(num_features, out_features) || (ld_xw % 4) != 0 || !gn::aligned16(xw))
            return gn::fail(GN_ERR_UNSUPPORTED, "no fused transform for %lld -> %lld features (or unaligned rows)",
                            (long long)num_features, (long long)out_features);
        const bool quad = gn::transform_takes_quad_kernel(num_features, out_features);
        if (planes && planes->planes && !quad) a.split = *planes;               // written by the kernel's own epilogue
        const gn_status s = gn::launch_aggregate_transform(a, weight, (int)out_features, gn::as_stream(stream));
        return quad ? then_split(s) : s;
    }
    return then_split(gn::launch_aggregate(a, gn::as_stream(stream)));
}

extern "C" size_t gn_split_planes_bytes(int64_t rows, int nt) {
    if (rows < 0 || nt < 1) return 0;
    return (size_t)(rows + 1) * 64 * (size_t)((3 * nt + 1) / 2);
}

extern "C" gn_status gn_split_planes_f32(const float* src, int64_t ld_src, int64_t rows, int64_t cols, int64_t col0,
                                         const gn_split_planes* planes, void* stream) {
    GN_REQUIRE(planes && planes->planes, "planes are null");
    GN_REQUIRE(planes->nt >= 1 && planes->nt <= 4 && rows >= 0 && rows <= planes->rows && cols >= 0 && col0 >= 0 &&
                   col0 + cols <= 16 * planes->nt, "split planes do not hold these columns");
    if (rows == 0 || cols == 0) return GN_OK;
    GN_REQUIRE(src && ld_src >= cols, "source matrix is null or its leading dimension too small");
    gn::k_split_planes<0><<<gn::stream_grid(rows * cols, 256), 256, 0, gn::as_stream(stream)>>>(src, ld_src, rows, (int)cols, (int)col0, *planes);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

// Backward of the aggregation with respect to its table: gxw[s, :] = sum_{e: src(e)=s} coef_e * g[dst(e), :]
// (the transpose of the normalised adjacency applied to the output gradient).
extern "C" gn_status gn_graph_aggregate_t_f32(const gn_graph_plan* plan, const float* g, int64_t ld_g,
                                              int64_t num_features, float* out, int64_t ld_out, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(plan->has_transpose, "call gn_graph_plan_build_transpose first");
    GN_REQUIRE(num_features >= 0 && num_features < (1ll << 31), "bad feature count");
    if (plan->table_rows == 0 || num_features == 0) return GN_OK;
    GN_REQUIRE(g && out, "feature pointers are null");
    GN_REQUIRE(ld_g >= num_features && ld_out >= num_features, "leading dimension smaller than the row length");
    gn::AggArgs a;
    a.rowptr = plan->t_rowptr.p;
    a.col = reinterpret_cast<const uint32_t*>(plan->t_col.p);
    a.coef = plan->t_coef.p;
    a.table = g;
    a.ld_table = ld_g;
    a.features = (int)num_features;
    a.rowdiv = nullptr;
    a.addend = nullptr;
    a.ld_addend = 0;
    a.bias = nullptr;
    a.relu = 0;
    a.out = out;
    a.ld_out = ld_out;
    a.rows = (int)plan->table_rows;
    a.nnz = plan->nnz;
    return gn::launch_aggregate(a, gn::as_stream(stream));
}

extern "C" int gn_transform_fusable(int64_t in_features, int64_t out_features) {
    return gn::transform_fusable(in_features, out_features) ? 1 : 0;
}

extern "C" int gn_graph_transform_fusable(const gn_graph_plan* plan, int64_t in_features, int64_t out_features) {
    if (!plan) return 0;
    return (gn::transform_fusable(in_features, out_features) ||
            gn::mfma_fusable(in_features, out_features, plan->rows, plan->nnz)) ? 1 : 0;
}
