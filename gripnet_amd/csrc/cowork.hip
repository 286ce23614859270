// Two independent, short, latency-bound pieces of the forward in ONE launch: the GCN-style aggregation of the external
// layer (645 destination rows on pose0-syn, 7.7 us as a launch of its own) and the relational layer's weights
// W_r = att . basis (8.9 us), which depend on the parameters only.  The first blocks of the grid run the
// aggregation, the rest the weights; the launch takes about as long as the longer of the two.  A second stream would
// overlap them as well, but the two cross-queue dependencies that needs cost more than either kernel (measured).
#include "aggregate.cuh"
#include "rgcn_weights.cuh"

// rgcn.hip / rgcn_acc.hip
bool gn_rgcn_acc_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases);
int gn_rgcn_acc_weights_args(const gn_rgcn_plan* plan, int64_t fin, const float* basis, const float* att, int64_t bases,
                             int64_t fout, void* ws, int exact, gn_rw::WeightsFragArgs* g);

namespace {

template <int LPE, int FOUT>
__global__ __launch_bounds__(256) void k_aggregate_transform_with_weights(gn::AggArgs a, const float* __restrict__ w,
                                                                         int agg_blocks, gn_rw::WeightsFragArgs g) {
    if ((int)blockIdx.x < agg_blocks) gn::aggregate_transform_body<LPE, FOUT>(a, w, (int)blockIdx.x, agg_blocks);
    else gn_rw::rgcn_weights_frag_body(g, (int)blockIdx.x - agg_blocks);
}

}  // namespace

extern "C" gn_status gn_graph_aggregate_with_rgcn_weights_f32(
    const gn_graph_plan* plan, const float* x, int64_t ld_x, int64_t num_features, const float* weight, int64_t out_features,
    const float* bias, int relu, float* out, int64_t ld_out, const gn_side_copy* side, const gn_rgcn_plan* rgcn_plan,
    int64_t rgcn_in_features, const float* basis, const float* att, int64_t num_bases, int64_t rgcn_out_features,
    void* rgcn_workspace, size_t rgcn_workspace_bytes, void* stream) {
    GN_REQUIRE(plan != nullptr && rgcn_plan != nullptr, "plan is null");
    const int key = (int)(num_features * 100 + out_features);
    const bool fusable = weight && plan->rows > 0 && (key == 6416 || key == 3216 || key == 1616) && (ld_x % 4) == 0 &&
                         gn::aligned16(x) && rgcn_plan->num_relations > 0 && rgcn_plan->num_nodes > 0 &&
                         gn_rgcn_acc_applicable(rgcn_plan, rgcn_in_features, rgcn_out_features, num_bases) &&
                         rgcn_workspace_bytes >= gn_rgcn_workspace_bytes(rgcn_plan, rgcn_in_features, rgcn_out_features, num_bases) &&
                         (reinterpret_cast<uintptr_t>(rgcn_workspace) & 15) == 0;
    if (!fusable) {                                            // any other shape: the two entry points, one after the other
        gn_status s = gn_rgcn_weights_f32(rgcn_plan, rgcn_in_features, basis, att, num_bases, rgcn_out_features,
                                          GN_RGCN_ARITH_FAST, rgcn_workspace, rgcn_workspace_bytes, stream);
        if (s != GN_OK) return s;
        return gn_graph_aggregate_f32(plan, x, ld_x, num_features, weight, out_features, bias, relu, out, ld_out, side, nullptr, stream);
    }
    GN_REQUIRE(x && out && basis && att, "operand pointer is null");
    GN_REQUIRE(ld_x >= num_features && ld_out >= out_features, "leading dimension smaller than the row length");
    gn::AggArgs a;
    a.rowptr = plan->rowptr.p;
    a.col = reinterpret_cast<const uint32_t*>(plan->col.p);
    a.coef = plan->coef.p;
    a.table = x; a.ld_table = ld_x; a.features = (int)num_features;
    a.rowdiv = nullptr; a.addend = nullptr; a.ld_addend = 0;
    a.bias = bias; a.relu = relu; a.out = out; a.ld_out = ld_out; a.rows = (int)plan->rows;
    gn_status ss = gn::check_side(side, plan->rows, &a.side);
    if (ss != GN_OK) return ss;
    gn_rw::WeightsFragArgs g;
    const int w_blocks = gn_rgcn_acc_weights_args(rgcn_plan, rgcn_in_features, basis, att, num_bases, rgcn_out_features,
                                                  rgcn_workspace, /*exact=*/0, &g);   // fragments of the GN_RGCN_ARITH_FAST path
    const int agg_blocks = (int)std::min<int64_t>(gn::ceil_div(a.rows, 4), GN_AGG_GRID);
    hipStream_t st = gn::as_stream(stream);
    switch (key) {
        case 6416: k_aggregate_transform_with_weights<16, 16><<<agg_blocks + w_blocks, 256, 0, st>>>(a, weight, agg_blocks, g); break;
        case 3216: k_aggregate_transform_with_weights<8, 16><<<agg_blocks + w_blocks, 256, 0, st>>>(a, weight, agg_blocks, g); break;
        default: k_aggregate_transform_with_weights<4, 16><<<agg_blocks + w_blocks, 256, 0, st>>>(a, weight, agg_blocks, g); break;
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}
