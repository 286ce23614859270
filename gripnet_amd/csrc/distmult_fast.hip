// LDS-resident specialisation of the DistMult decoder for small node tables.
// (placeholder until the fast path lands: reports "not applicable", general path is used)
#include "common.h"

bool gn_distmult_fast_applicable(int64_t, int64_t, int64_t, int64_t, const void*, const void*) { return false; }
gn_status gn_distmult_fast_forward(const float*, int64_t, int64_t, int64_t, const int64_t*, const int64_t*,
                                   const int64_t*, const float*, int64_t, int64_t, int64_t, int, float*, int32_t*,
                                   hipStream_t) {
    return gn::fail(GN_ERR_UNSUPPORTED, "fast DistMult path not built");
}
