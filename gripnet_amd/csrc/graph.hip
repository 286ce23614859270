// GCN-style aggregation over a cached plan (square graphs with self loops, or the bipartite
// external layer).  Replaces propagate/message/update of myGCN (gripnet/layers.py:92-100).
#include "aggregate.cuh"

extern "C" gn_status gn_graph_aggregate_f32(const gn_graph_plan* plan, const float* xw, int64_t ld_xw,
                                            int64_t num_features, const float* bias, int relu, float* out,
                                            int64_t ld_out, const gn_side_copy* side, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(num_features >= 0 && num_features < (1ll << 31), "bad feature count");
    if (plan->rows == 0 || num_features == 0) return GN_OK;
    GN_REQUIRE(xw && out, "feature pointers are null");
    GN_REQUIRE(ld_xw >= num_features && ld_out >= num_features, "leading dimension smaller than the row length");
    gn::AggArgs a;
    a.rowptr = plan->rowptr.p;
    a.col = reinterpret_cast<const uint32_t*>(plan->col.p);
    a.coef = plan->coef.p;
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
    gn_status ss = gn::check_side(side, plan->rows, &a.side);
    if (ss != GN_OK) return ss;
    return gn::launch_aggregate(a, gn::as_stream(stream));
}
