// DistMult decoder (multiRelaInnerProductDecoder.forward, gripnet/decoder.py:19-23):
//   out[e] = sigma?( sum_k z[u_e,k] * z[v_e,k] * D[r_e,k] )
//
// General path: the (u, v, r) int64 triples are streamed with coalesced 512-byte loads (one
// triple per lane), then the wave walks its 64 edges four at a time: 16 lanes cover one edge's
// three feature rows with 16-byte loads (z and D stay L2 / Infinity-Cache resident), fold the
// 16 partial products with cross-lane shuffles, and the 64 results leave in one coalesced store.
#include "common.h"

namespace {

constexpr int kLpe = 16;                 // lanes per edge
constexpr int kSlots = gn::kWave / kLpe; // edges in flight per wave

struct DmArgs {
    const float* z; int64_t ld_z; int64_t n; int features;
    const int64_t* u; const int64_t* v; const int64_t* et;
    const float* d; int64_t ld_d; int64_t r;
    int64_t e; int sigmoid; float* out; int32_t* err;
};

template <int VEC>
__global__ __launch_bounds__(256) void k_distmult(DmArgs a) {
    const int lane = threadIdx.x & 63;
    const int slot = lane / kLpe;
    const int j = lane % kLpe;
    const int64_t wave = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t)gridDim.x * blockDim.x) >> 6;
    const int units = a.features / VEC;

    for (int64_t e0 = wave * 64; e0 < a.e; e0 += n_waves * 64) {
        const int64_t mine = e0 + lane;
        int64_t mu = 0, mv = 0, mr = 0;
        bool ok = true;
        if (mine < a.e) {
            mu = a.u[mine]; mv = a.v[mine]; mr = a.et[mine];
            ok = (uint64_t)mu < (uint64_t)a.n && (uint64_t)mv < (uint64_t)a.n && (uint64_t)mr < (uint64_t)a.r;
            if (!ok) { mu = mv = mr = 0; }
        }
        // node / relation ids fit 32 bits once validated (tables are far smaller than 2^31 rows)
        const int iu = (int)mu, iv = (int)mv, ir = (int)mr;
        const int cnt = (int)min((int64_t)64, a.e - e0);
        float result = 0.f;
        for (int it = 0; it * kSlots < cnt; ++it) {
            const int idx = it * kSlots + slot;
            const int uu = __shfl(iu, idx), vv = __shfl(iv, idx), rr = __shfl(ir, idx);
            float acc = 0.f;
            if (idx < cnt) {
                const float* zu = a.z + (int64_t)uu * a.ld_z;
                const float* zv = a.z + (int64_t)vv * a.ld_z;
                const float* dr = a.d + (int64_t)rr * a.ld_d;
                for (int c = j; c < units; c += kLpe) {
                    if constexpr (VEC == 4) {
                        const float4 p = *reinterpret_cast<const float4*>(zu + 4 * c);
                        const float4 q = *reinterpret_cast<const float4*>(zv + 4 * c);
                        const float4 w = *reinterpret_cast<const float4*>(dr + 4 * c);
                        acc += p.x * q.x * w.x;
                        acc += p.y * q.y * w.y;
                        acc += p.z * q.z * w.z;
                        acc += p.w * q.w * w.w;
                    } else {
                        acc += zu[c] * zv[c] * dr[c];
                    }
                }
            }
#pragma unroll
            for (int off = 1; off < kLpe; off <<= 1) acc += __shfl_xor(acc, off);
            // lane l keeps the score of edge e0 + l: iteration l / kSlots, slot l % kSlots
            const float got = __shfl(acc, (lane % kSlots) * kLpe);
            if (lane / kSlots == it) result = got;
        }
        if (mine < a.e) {
            if (a.sigmoid) result = 1.0f / (1.0f + expf(-result));
            if (!ok) {
                result = __builtin_nanf("");
                if (a.err) atomicOr(a.err, 1);
            }
            a.out[mine] = result;
        }
    }
}

}  // namespace

// distmult_fast.hip
bool gn_distmult_fast_applicable(int64_t n, int64_t f, int64_t ld_z, int64_t ld_d, const void* z, const void* d);
gn_status gn_distmult_fast_forward(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                   const int64_t* v, const int64_t* et, const float* d, int64_t ld_d, int64_t r,
                                   int64_t e, int sigmoid, float* out, int32_t* err, hipStream_t st);

extern "C" gn_status gn_distmult_forward_f32(const float* z, int64_t ld_z, int64_t n, int64_t f, const int64_t* u,
                                             const int64_t* v, const int64_t* et, const float* d, int64_t ld_d,
                                             int64_t r, int64_t e, int apply_sigmoid, float* out, int32_t* err,
                                             void* stream) {
    GN_REQUIRE(n >= 0 && f >= 0 && r >= 0 && e >= 0, "negative size");
    GN_REQUIRE(f < (1ll << 31) && n < (1ll << 31) && r < (1ll << 31), "table too large");
    if (e == 0) return GN_OK;
    GN_REQUIRE(n > 0 && r > 0, "edges given but the node or relation table is empty");
    GN_REQUIRE(z && u && v && et && d && out, "operand pointer is null");
    GN_REQUIRE(ld_z >= f && ld_d >= f, "leading dimension smaller than the row length");
    hipStream_t st = gn::as_stream(stream);
    if (gn_distmult_fast_applicable(n, f, ld_z, ld_d, z, d))
        return gn_distmult_fast_forward(z, ld_z, n, f, u, v, et, d, ld_d, r, e, apply_sigmoid, out, err, st);
    DmArgs a;
    a.z = z; a.ld_z = ld_z; a.n = n; a.features = (int)f; a.u = u; a.v = v; a.et = et;
    a.d = d; a.ld_d = ld_d; a.r = r; a.e = e; a.sigmoid = apply_sigmoid; a.out = out; a.err = err;
    const bool vec = (f % 4 == 0) && (ld_z % 4 == 0) && (ld_d % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d)) & 15) == 0;
    const int grid = gn::stream_grid(gn::ceil_div(e, 64) * 64, 256);
    if (vec) k_distmult<4><<<grid, 256, 0, st>>>(a); else k_distmult<1><<<grid, 256, 0, st>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}
