// LDS-resident DistMult decoder for small node tables (the drug supervertex: n_d = 645, F = 80).
//
// The decoder streams 24 B of indices per edge but gathers 2 x F x 4 B of node features per
// edge; served from L2 that gather is ~40x the HBM stream.  Here the node table lives in LDS:
// the feature dimension is cut into P column phases so that n x width x 4 B fits the CU's
// 160 KB, a persistent workgroup (one per CU) keeps one column phase resident and walks its
// contiguous edge range once per phase, carrying the partial sums through `out`.
//
// Per wave: 64 (u, v, r) triples arrive with three coalesced loads.  Each quad of lanes owns 4
// consecutive edges; the edge being worked on is broadcast inside the quad with DPP (quad_perm,
// no LDS traffic), the quad's 4 lanes cover the phase's columns of z[u] and z[v] with
// ds_read_b128 (16 edges in flight per wave step), the relation row D[r] stays in registers
// while r does not change (type-sorted positives), and the quad's partial products are folded
// with two DPP adds.
#include "distmult_quad.cuh"

namespace {

using namespace gn_dm;

struct DmFastArgs {
    const float* z; int64_t ld_z; int n; int features;
    const int64_t* u; const int64_t* v; const int64_t* et;
    const uint32_t* packed; const uint16_t* rel16;   // PACKED: u | v << 16 and the relation of every edge (6 bytes per edge)
    const float* d; int64_t ld_d; int r;
    int64_t e; int64_t edges_per_wg; int sigmoid; float* out; int32_t* err;
    int n_phases; int c0[kMaxPhases]; int width[kMaxPhases];
    int stride4;           // LDS row stride in float4, the same in every phase (see gn_distmult_fast_forward)
    // Raw int64 triples, several column phases: the first phase leaves every edge's triple in LDS behind the table, narrowed to
    // `stage_bytes` (4: u | v << 10 | r << 20 for n <= 1023, r <= 4096; 8: u | v << 16, r; 0: no staging), and the later phases
    // read it there - 24 bytes per edge cross from HBM once instead of once per phase (pose0-syn: 155 MB moved for 56 MB of
    // algorithmic bytes before).  A workgroup whose range does not fit walks it in tiles of `tile_edges`, all phases per tile.
    int stage_bytes; int64_t tile_edges; int stage_off;
};

struct Batch {            // one lane's edge of a 64-edge batch
    int64_t u, v, r;
    float carried;
};

// stage: the tile's narrowed triples in LDS (null: none); first: this is the phase that reads them from memory and leaves them there
template <bool PACKED>
__device__ __forceinline__ Batch load_batch(const DmFastArgs& a, int64_t mine, int64_t lo, int64_t hi, bool carry, char* stage, bool first) {
    Batch b;
    b.u = b.v = b.r = 0;
    b.carried = 0.f;
    if (mine < hi) {
        if constexpr (PACKED) {
            const uint32_t w = a.packed[mine];
            b.u = w & 0xffffu; b.v = w >> 16; b.r = a.rel16[mine];
        } else {
            if (stage != nullptr && !first) {
                if (a.stage_bytes == 4) {
                    const uint32_t w = reinterpret_cast<const uint32_t*>(stage)[mine - lo];
                    b.u = w & 1023u; b.v = (w >> 10) & 1023u; b.r = w >> 20;     // (an edge with an id outside its table: u = 1023 >= n)
                } else {
                    const uint2 w = reinterpret_cast<const uint2*>(stage)[mine - lo];
                    b.u = w.x & 0xffffu; b.v = w.x >> 16; b.r = (int64_t)w.y;     // (... : r = 2^32 - 1)
                }
            } else {
                b.u = a.u[mine]; b.v = a.v[mine]; b.r = a.et[mine];
                if (stage != nullptr) {
                    const bool ok = (uint64_t)b.u < (uint64_t)a.n && (uint64_t)b.v < (uint64_t)a.n && (uint64_t)b.r < (uint64_t)a.r;
                    if (a.stage_bytes == 4)
                        reinterpret_cast<uint32_t*>(stage)[mine - lo] = ok ? ((uint32_t)b.u | (uint32_t)b.v << 10 | (uint32_t)b.r << 20) : 1023u;
                    else
                        reinterpret_cast<uint2*>(stage)[mine - lo] = ok ? make_uint2((uint32_t)b.u | (uint32_t)b.v << 16, (uint32_t)b.r) : make_uint2(0u, 0xffffffffu);
                }
            }
        }
        if (carry) b.carried = a.out[mine];
    }
    return b;
}

template <int W4, int CPL, bool PACKED>
__device__ __forceinline__ void run_phase(const DmFastArgs& a, const char* lds, int stride_bytes, int c0, int w4, bool first, bool last,
                                          int64_t wg_lo, int64_t wg_hi, int wave, int lane, char* stage) {
    const int l4 = lane & 3;
    const float* dcol = a.d + c0 + 4 * l4;
    int cur_r = -1;                                         // wave-uniform: relation whose chunks sit in dreg
    f32x4 dreg[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) dreg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int64_t kStride = (kThreads / 64) * 64;

    int64_t e0 = wg_lo + wave * 64;
    Batch nxt = load_batch<PACKED>(a, e0 + lane, wg_lo, wg_hi, !first, stage, first);
    for (; e0 < wg_hi; e0 += kStride) {
        const int64_t mine = e0 + lane;
        const Batch cur = nxt;
        nxt = load_batch<PACKED>(a, mine + kStride, wg_lo, wg_hi, !first, stage, first);   // in flight while this batch computes
        const bool ok = (uint64_t)cur.u < (uint64_t)a.n && (uint64_t)cur.v < (uint64_t)a.n &&
                        (uint64_t)cur.r < (uint64_t)a.r;
        const int iu = ok ? (int)cur.u : 0, iv = ok ? (int)cur.v : 0;
        int ir = ok ? (int)cur.r : 0;
        // padding lanes of a ragged tail copy lane 0's relation so that they do not break uniformity
        const int r0 = __builtin_amdgcn_readfirstlane(ir);
        if (mine >= wg_hi) ir = r0;
        float result = 0.f;
        if (__all(ir == r0)) {                              // wave-uniform branch
            if (r0 != cur_r) {
                cur_r = r0;
#pragma unroll
                for (int i = 0; i < CPL; ++i)
                    dreg[i] = (l4 + 4 * i < w4) ? *reinterpret_cast<const f32x4*>(dcol + (int64_t)r0 * a.ld_d + 16 * i)
                                                : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            quad_step<0, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<1, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<2, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<3, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
        } else {
            quad_step<0, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<1, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<2, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<3, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
        }
        if (mine < wg_hi) {
            float total = cur.carried + result;
            if (last) {
                if (a.sigmoid) total = sigmoid_f32(total);
                if (!ok) {
                    total = __builtin_nanf("");
                    if (a.err) atomicOr(a.err, 1);
                }
            }
            a.out[mine] = total;
        }
    }
}

template <bool PACKED>
__global__ __launch_bounds__(kThreads) void k_distmult_lds(DmFastArgs a) {
    extern __shared__ float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int64_t wg_lo = (int64_t)blockIdx.x * a.edges_per_wg;
    const int64_t wg_hi = min(a.e, wg_lo + a.edges_per_wg);
    const char* lds = reinterpret_cast<const char*>(lds4);
    char* stage = (!PACKED && a.stage_bytes > 0) ? reinterpret_cast<char*>(lds4) + a.stage_off : nullptr;
    const int64_t tile = stage ? a.tile_edges : max((int64_t)1, wg_hi - wg_lo);
    bool table_in_use = false;

    for (int64_t t_lo = wg_lo; t_lo < wg_hi; t_lo += tile) {
        const int64_t t_hi = min(wg_hi, t_lo + tile);
        for (int ph = 0; ph < a.n_phases; ++ph) {
            const int c0 = a.c0[ph], w4 = a.width[ph] >> 2;
            if (table_in_use) __syncthreads();                  // everyone is done with the previous phase's table (and the previous tile's triples)
            table_in_use = true;
            switch (w4) {                                       // compile-time row width where it is a common one
                case 16: fill_table<16>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
                case 12: fill_table<12>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
                case 8: fill_table<8>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
                default: fill_table<0>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
            }
            __syncthreads();
            const bool first = ph == 0, last = ph == a.n_phases - 1;
            switch (w4) {                                       // common widths get compile-time addressing
                case 16: run_phase<16, 4, PACKED>(a, lds, a.stride4 * 16, c0, w4, first, last, t_lo, t_hi, wave, lane, stage); break;
                case 12: run_phase<12, 3, PACKED>(a, lds, a.stride4 * 16, c0, w4, first, last, t_lo, t_hi, wave, lane, stage); break;
                case 8: run_phase<8, 2, PACKED>(a, lds, a.stride4 * 16, c0, w4, first, last, t_lo, t_hi, wave, lane, stage); break;
                case 4: run_phase<4, 1, PACKED>(a, lds, a.stride4 * 16, c0, w4, first, last, t_lo, t_hi, wave, lane, stage); break;
                default: run_phase<0, 4, PACKED>(a, lds, a.stride4 * 16, c0, w4, first, last, t_lo, t_hi, wave, lane, stage); break;
            }
        }
    }
}

}  // namespace

bool gn_distmult_fast_applicable(int64_t n, int64_t f, int64_t ld_z, int64_t ld_d, const void* z, const void* d);
static gn_status distmult_lds_launch(DmFastArgs& a, int64_t n, int64_t f, int64_t e, bool packed, hipStream_t st);

bool gn_distmult_fast_applicable(int64_t n, int64_t f, int64_t ld_z, int64_t ld_d, const void* z, const void* d) {
    if (gn::fast_paths_disabled()) return false;
    if (n < 1 || n > 65535 || f < 4 || f % 4 != 0 || ld_z % 4 != 0 || ld_d % 4 != 0) return false;
    if (((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d)) & 15) != 0) return false;
    int c0[kMaxPhases], w[kMaxPhases];
    const int phases = plan_phases(n, f, c0, w);
    return phases >= 1 && phases <= 4;   // beyond that the general (L2-gather) kernel is the better choice
}

static gn_status distmult_lds_launch(DmFastArgs& a, int64_t n, int64_t f, int64_t e, bool packed, hipStream_t st) {
    a.n_phases = plan_phases(n, f, a.c0, a.width);
    int max_w = 0;
    for (int k = 0; k < a.n_phases; ++k) max_w = std::max(max_w, a.width[k]);
    const int stride4 = lds_stride4(n, max_w);
    a.stride4 = stride4;
    size_t lds_bytes = (size_t)n * stride4 * 16;
    // one workgroup per CU; every workgroup's range is a multiple of 64 edges
    int64_t groups = std::min<int64_t>(256, gn::ceil_div(e, 64 * (kThreads / 64)));
    if (groups < 1) groups = 1;
    a.edges_per_wg = gn::ceil_div(gn::ceil_div(e, groups), 64) * 64;
    groups = gn::ceil_div(e, a.edges_per_wg);
    // the triples of the raw int64 list staged in LDS behind the table (see DmFastArgs): where a tile of a useful size fits, and
    // the table fills of the extra tiles (every workgroup reads the whole phase table again: ~3 us each at this size) cost less
    // than the passes over the list they save (24 bytes per edge and phase at ~5 TB/s)
    a.stage_bytes = 0; a.tile_edges = 0; a.stage_off = (int)lds_bytes;
    if (!packed && a.n_phases > 1) {
        const int sb = (n <= 1023 && a.r <= 4096) ? 4 : 8;
        const int64_t room = (int64_t)160 * 1024 - (int64_t)lds_bytes;
        const int64_t tile = room / sb / 64 * 64;
        if (tile >= 1024) {
            const int64_t tiles = gn::ceil_div(a.edges_per_wg, tile);
            const double fill_us = (double)lds_bytes * (double)groups / 1.0e7, pass_us = (double)e * 24.0 / 5.0e6;
            if ((double)(tiles - 1) * a.n_phases * fill_us < (double)(a.n_phases - 1) * pass_us) {
                a.stage_bytes = sb;
                a.tile_edges = std::min<int64_t>(tile, a.edges_per_wg);
                lds_bytes += (size_t)a.tile_edges * sb;
            }
        }
    }
    if (packed) {
        { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_distmult_lds<true>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
        k_distmult_lds<true><<<(unsigned)groups, kThreads, lds_bytes, st>>>(a);
    } else {
        { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_distmult_lds<false>), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
        k_distmult_lds<false><<<(unsigned)groups, kThreads, lds_bytes, st>>>(a);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

gn_status gn_distmult_fast_forward(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                   const int64_t* v, const int64_t* et, const float* d, int64_t ld_d, int64_t r,
                                   int64_t e, int sigmoid, float* out, int32_t* err, hipStream_t st) {
    GN_REQUIRE(r >= 1, "no relations");
    DmFastArgs a;
    a.z = z; a.ld_z = ld_z; a.n = (int)n; a.features = (int)f; a.u = u; a.v = v; a.et = et; a.packed = nullptr; a.rel16 = nullptr;
    a.d = d; a.ld_d = ld_d; a.r = (int)r; a.e = e; a.sigmoid = sigmoid; a.out = out; a.err = err;
    return distmult_lds_launch(a, n, f, e, false, st);
}

// The same decoder on PACKED triples: u | v << 16 (uint32) and the relation (uint16) of every edge - six bytes per edge
// and column phase instead of 24.  What gn_negative_sampler_sample_packed writes next to the int64 pairs, with the
// relation ids of the static edge_type tensor (the negatives of GripNet-pose.py:131,138 are scored with the positives'
// train_et).  Ids outside their table give NaN and set bit 0 of *error_flag, as in gn_distmult_forward_f32.
extern "C" gn_status gn_distmult_packed_forward_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const uint32_t* packed_uv,
                                                    const uint16_t* relation, const float* d, int64_t ld_d, int64_t r,
                                                    int64_t e, int apply_sigmoid, float* out, int32_t* err, void* stream) {
    GN_REQUIRE(n >= 0 && f >= 0 && r >= 0 && e >= 0, "negative size");
    if (e == 0) return GN_OK;
    GN_REQUIRE(n > 0 && r > 0, "edges given but the node or relation table is empty");
    GN_REQUIRE(z && packed_uv && relation && d && out, "operand pointer is null");
    GN_REQUIRE(ld_z >= f && ld_d >= f, "leading dimension smaller than the row length");
    if (n > 65535 || r > 65535 || !gn_distmult_fast_applicable(n, f, ld_z, ld_d, z, d))
        return gn::fail(GN_ERR_UNSUPPORTED, "the packed decoder needs node and relation ids of 16 bits and a node table that fits the LDS "
                                            "in four column phases: use gn_distmult_forward_f32");
    DmFastArgs a;
    a.z = z; a.ld_z = ld_z; a.n = (int)n; a.features = (int)f; a.u = nullptr; a.v = nullptr; a.et = nullptr;
    a.packed = packed_uv; a.rel16 = relation;
    a.d = d; a.ld_d = ld_d; a.r = (int)r; a.e = e; a.sigmoid = apply_sigmoid; a.out = out; a.err = err;
    return distmult_lds_launch(a, n, f, e, true, gn::as_stream(stream));
}
