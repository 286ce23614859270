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
                                            const gn_side_copy* side, void* stream) {
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
    gn_status ss = gn::check_side(side, plan->rows, &a.side);
    if (ss != GN_OK) return ss;
    if (gn_blocked_applicable(plan, xw, ld_xw, num_features, weight, width))
        return gn_blocked_aggregate(plan, xw, ld_xw, num_features, weight, width, bias, relu, out, ld_out, a.side,
                                    gn::as_stream(stream));
    if (weight && gn::mfma_fusable(num_features, out_features, plan->rows, plan->nnz) && (ld_xw % 4) == 0 && gn::aligned16(xw))
        return gn::launch_aggregate_mfma(a, weight, (int)out_features, gn::as_stream(stream));   // wide layers: W on the matrix cores
    if (weight) {            // aggregate the input rows, then contract with W in the epilogue
        if (!gn::transform_fusable(num_features, out_features) || (ld_xw % 4) != 0 || !gn::aligned16(xw))
            return gn::fail(GN_ERR_UNSUPPORTED, "no fused transform for %lld -> %lld features (or unaligned rows)",
                            (long long)num_features, (long long)out_features);
        return gn::launch_aggregate_transform(a, weight, (int)out_features, gn::as_stream(stream));
    }
    return gn::launch_aggregate(a, gn::as_stream(stream));
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
