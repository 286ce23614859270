// LDS-resident specialisation of the multi-relational layer for small supervertices.
// (placeholder until the fast path lands: reports "not applicable", general path is used)
#include <vector>

#include "common.h"

gn_status gn_rgcn_build_fast_segments(gn_rgcn_plan* plan, const int64_t*, const int64_t*,
                                      const std::vector<int64_t>&, hipStream_t) {
    plan->fast_ok = 0;
    return GN_OK;
}

bool gn_rgcn_fast_applicable(const gn_rgcn_plan*, int64_t, int64_t, int64_t) { return false; }
size_t gn_rgcn_fast_workspace_bytes(const gn_rgcn_plan*, int64_t, int64_t, int64_t) { return 0; }
gn_status gn_rgcn_fast_forward(const gn_rgcn_plan*, const float*, int64_t, int64_t, const float*, const float*,
                               int64_t, const float*, const float*, int64_t, int, int, float*, int64_t, void*, size_t,
                               hipStream_t) {
    return gn::fail(GN_ERR_UNSUPPORTED, "fast RGCN path not built");
}
