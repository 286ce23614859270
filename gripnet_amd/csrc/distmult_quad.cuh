// Pieces shared by the LDS-resident DistMult kernels (distmult_fast.hip: raw int64 triples every call;
// distmult_plan.hip: a cached, re-encoded static edge list): DPP helpers, the quad-per-edge step, the column
// phases and the LDS row stride.
#pragma once

#include "common.h"

namespace gn_dm {

constexpr int kMaxPhases = 8;
#ifndef GN_DM_THREADS
#define GN_DM_THREADS 1024
#endif
constexpr int kThreads = GN_DM_THREADS;
constexpr size_t kLdsBudget = 150 * 1024;   // of 160 KB; the rest is left to the runtime

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));


// quad_perm controls write every lane, so no "old" value has to be set up in front of the DPP move
template <int CTRL>
__device__ __forceinline__ int dpp_i(int x) { return __builtin_amdgcn_mov_dpp(x, CTRL, 0xf, 0xf, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false));
}
// sum over the 16 lanes of a DPP row, result in every lane
__device__ __forceinline__ float row_sum16(float x) {
    x = dpp_add<0xB1>(x);    // quad_perm [1,0,3,2]
    x = dpp_add<0x4E>(x);    // quad_perm [2,3,0,1]
    x = dpp_add<0x141>(x);   // row_half_mirror
    x = dpp_add<0x140>(x);   // row_mirror
    return x;
}

// sigma(x) on the hardware exp2 / reciprocal: |error| < 1e-6 against the correctly rounded value (the contract is
// 1e-4), a third of the instructions of expf + an IEEE division
__device__ __forceinline__ float sigmoid_f32(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// One column phase of the node table into LDS: rows of `stride4` float4, the first w4 of them columns c0 .. c0 + 4 w4 of
// z, the rest zero (idle lanes of a ragged chunk group read the padding, x 0).  Every lane carries a float4 (a flat
// index over the n x w4 elements; W4 > 0 makes the division a compile-time one), eight loads in flight per thread
// before the first store: the table was written by the previous kernel, so these are Infinity Cache / HBM round
// trips, and every workgroup reads the same table at the same time - each starts at its own offset so that they do
// not all queue on the same L2 channel.
template <int W4>
__device__ __forceinline__ void fill_table(float4* lds4, const float* __restrict__ z, int64_t ld_z, int n, int c0, int w4rt,
                                           int stride4, int tid) {
    const int w4 = W4 > 0 ? W4 : w4rt;
    const int total = n * w4;
    const int rot = (int)((blockIdx.x * 977u) % (unsigned)total);
    for (int base = 0; base < total; base += 8 * kThreads) {
        float4 v[8];
        int at[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            int i = min(base + k * kThreads + tid, total - 1) + rot;
            i = i < total ? i : i - total;
            const int row = i / w4, c4 = i - row * w4;
            at[k] = row * stride4 + c4;
            v[k] = *reinterpret_cast<const float4*>(z + (int64_t)row * ld_z + c0 + 4 * c4);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (base + k * kThreads + tid < total) lds4[at[k]] = v[k];
    }
    const int pad = stride4 - w4;                                  // zero the row padding
    for (int i = tid; i < n * pad; i += kThreads) {
        const int row = i / pad;
        lds4[row * stride4 + w4 + (i - row * pad)] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// One edge per QUAD (4 lanes), 16 edges per wave step.  Lane l4 of the quad covers the 16-byte
// chunks l4, l4+4, ... of the phase's columns, so a 48-column phase is 3 x (2 ds_read_b128 + a
// few FMAs) per lane and the per-edge fold is two DPP adds.
// UNIFORM_R: the whole 64-edge batch has one relation whose D chunks already sit in `dreg`;
// otherwise each step gathers its D chunks from L2.
// W4 > 0: the phase width (in float4 chunks) is a compile-time constant; W4 == 0: runtime w4.
template <int S, int W4, int CPL, bool UNIFORM_R>
__device__ __forceinline__ void quad_step(const char* __restrict__ lds, int stride_bytes, int w4, int l4, int iu, int iv, int ir,
                                          const float* __restrict__ dcol, int64_t ld_d, const f32x4 (&dreg)[CPL],
                                          float& result) {
    constexpr int kBcast = S * 0x55;     // quad_perm [S,S,S,S]
    const int uu = dpp_i<kBcast>(iu), vv = dpp_i<kBcast>(iv);
    const int width4 = W4 > 0 ? W4 : w4;
    const char* pu = lds + __umul24(uu, stride_bytes) + l4 * 16;
    const char* pv = lds + __umul24(vv, stride_bytes) + l4 * 16;
    const float* dr = dcol;
    if constexpr (!UNIFORM_R) dr = dcol + (int64_t)dpp_i<kBcast>(ir) * ld_d;
    // all LDS reads (and D gathers) of the step are issued before any of them is consumed
    f32x4 P[CPL], Q[CPL], D[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        const bool present = W4 > 0 ? (4 * i < W4) : (4 * i < w4);
        const bool ragged = W4 > 0 ? (4 * i + 3 >= W4) : true;   // only the last chunk group can be partial
        const bool live = present && (!ragged || (l4 + 4 * i < width4));
        const int off = live ? 64 * i : 0;
        if (W4 > 0 && !present) continue;
        P[i] = *reinterpret_cast<const f32x4*>(pu + off);
        Q[i] = *reinterpret_cast<const f32x4*>(pv + off);
        if constexpr (UNIFORM_R) D[i] = dreg[i]; else D[i] = *reinterpret_cast<const f32x4*>(dr + off / 4);
        if (!live) D[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x2 acc2 = {0.f, 0.f};             // two running sums -> v_pk_mul_f32 / v_pk_fma_f32
#pragma unroll
    for (int i = 0; i < CPL; ++i) {
        if (W4 > 0 && 4 * i >= W4) continue;
        const f32x2 lo = P[i].xy * Q[i].xy, hi = P[i].zw * Q[i].zw;
        acc2 = lo * D[i].xy + acc2;
        acc2 = hi * D[i].zw + acc2;
    }
    float acc = acc2.x + acc2.y;
    acc = dpp_add<0xB1>(acc);            // quad_perm [1,0,3,2]
    acc = dpp_add<0x4E>(acc);            // quad_perm [2,3,0,1]
    if (l4 == S) result = acc;
}


// Column phases: as few as possible; each at most 64 columns (4 lanes x 4 chunks x float4) with
// n x width x 4 B inside the LDS budget; widths are multiples of 16 columns except the last.
inline int plan_phases(int64_t n, int64_t f, int* c0, int* width) {
    if (n <= 0 || f <= 0 || f % 4 != 0) return 0;
    int64_t max_w = (int64_t)(kLdsBudget / (n * 4)) / 16 * 16;
    if (max_w > 64) max_w = 64;
    if (max_w < 16) return 0;
    const int64_t phases = gn::ceil_div(f, max_w);
    if (phases > kMaxPhases) return 0;
    int64_t c = 0;
    int k = 0;
    while (c < f) {
        const int64_t w = std::min(max_w, f - c);
        c0[k] = (int)c;
        width[k] = (int)w;
        c += w;
        ++k;
    }
    return k;
}

// LDS rows keep one stride for all phases, an odd number of 64-byte slots where there is room: a quad reads
// 64 contiguous bytes of a row, and with rows 128 bytes apart (a 32-column phase) the four quads of a
// 16-lane access group would share only two of the four 64-byte bank slots.
inline int lds_stride4(int64_t n, int max_w) {
    int stride4 = max_w / 4;
    if ((stride4 / 4) % 2 == 0 && (size_t)n * (stride4 + 4) * 16 <= kLdsBudget + 8 * 1024) stride4 += 4;
    return stride4;
}

}  // namespace gn_dm
