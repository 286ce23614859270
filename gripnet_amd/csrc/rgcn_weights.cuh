// W_r = sum_b att[r, b] basis[b] as MFMA B-operand fragments for k_rgcn_acc (rgcn_acc.hip): the device body, shared
// by the stand-alone kernel and by the launch that runs it next to an aggregation (cowork.hip).
#pragma once

#include "common.h"

namespace gn_rw {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// hi = bf16(v), lo = bf16(v - hi) for two floats at a time; returns the packed pairs (element 0 in the low half).
__device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
    const f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u)};
    lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bf16x2));
}

// W_r = sum_b att[r, b] basis[b]  (layers.py:172-173), written as the B operand of the transform above:
// (k group kg, e-th value of the group) is feature f(kg, e) = 16 (e / 4) + 4 kg + e % 4 (the gather layout of k_rgcn_acc)
//   fp32:  Wfrag[r][nt][p][lane = kg * 16 + col][jj] = W_r[f(kg, 4 p + jj)][16 nt + col]
//   split: fragment (r, nt, m, hi | lo), lane kg * 16 + col, bf16 element j = W_r[f(kg, 8 m + j)][16 nt + col];
//          a quarter that ends half way through its last MFMA stores {hi, hi} and {lo, 0} (see k_rgcn_acc).
// One wave = 16 relations x (up to 8 consecutive k of one quarter) x 16 columns: up to eight fp32 MFMA tiles over
// the bases, and every lane ends up with whole 16-byte fragment elements (16 lanes = one 256-byte run).
struct WeightsFragArgs {
    const float* att; const float* basis; f32x4* wfrag;
    int relations, bases, fin, fout, tasks, split;
};

// `block` = index among the blocks (of 256 threads = four tasks) that run this body
__device__ __forceinline__ void rgcn_weights_frag_body(const WeightsFragArgs& g, int block) {
    const float* __restrict__ att = g.att;
    const float* __restrict__ basis = g.basis;
    f32x4* __restrict__ wfrag = g.wfrag;
    const int relations = g.relations, bases = g.bases, fin = g.fin, fout = g.fout, tasks = g.tasks, split = g.split;
    const int lane = threadIdx.x & 63;
    const int task = block * 4 + (threadIdx.x >> 6);
    if (task >= tasks) return;
    const int n16 = lane & 15, q = lane >> 4;
    const int nts = fout / 16, KQ = fin / 4, KP = KQ / 4, M = (KQ + 7) / 8;
    const int nt = task % nts, m = (task / nts) % M, kg = (task / (nts * M)) % 4, rb = task / (nts * M * 4);
    const int r0 = rb * 16;
    const int nk = min(8, KQ - 8 * m);                                // 8, or 4 when the quarter ends half way (wave-uniform)
    const int arow = min(r0 + n16, relations - 1);
    f32x4 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < bases; b0 += 32) {                          // eight K steps per trip: every load is in flight before the first MFMA
        float av[8], bv[8][8];
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int bc = min(b0 + 4 * h + q, bases - 1);           // unconditional, clamped; zeroed by select below
            av[h] = att[(int64_t)arow * bases + bc];
            const float* __restrict__ bp = basis + (int64_t)bc * fin * fout + nt * 16 + n16;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int e = 8 * m + min(j, nk - 1);
                bv[h][j] = bp[(16 * (e / 4) + 4 * kg + e % 4) * fout];
            }
        }
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const bool live = b0 + 4 * h + q < bases;
            const float a = live ? av[h] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, live ? bv[h][j] : 0.f, acc[j], 0, 0, 0);
            if (nk == 8) {
#pragma unroll
                for (int j = 4; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, live ? bv[h][j] : 0.f, acc[j], 0, 0, 0);
            }
        }
    }
    // lane (col, q), register i: W_r[f(kg, 8 m + j)][16 nt + col] for relation r0 + 4 q + i, j = 0 .. nk - 1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + 4 * q + i;
        if (r >= relations) continue;
        if (!split) {
            f32x4* o = wfrag + (((size_t)r * nts + nt) * KP + 2 * m) * 64 + kg * 16 + n16;
            o[0] = (f32x4){acc[0][i], acc[1][i], acc[2][i], acc[3][i]};
            if (nk == 8) o[64] = (f32x4){acc[4][i], acc[5][i], acc[6][i], acc[7][i]};
        } else {
            uint32_t h0, h1, h2, h3, l0, l1, l2, l3;
            split2(acc[0][i], acc[1][i], h0, l0);
            split2(acc[2][i], acc[3][i], h1, l1);
            if (nk == 8) {
                split2(acc[4][i], acc[5][i], h2, l2);
                split2(acc[6][i], acc[7][i], h3, l3);
            } else {                                                  // packed half MFMA: B = {hi, hi} and {lo, 0}
                h2 = h0; h3 = h1; l2 = 0u; l3 = 0u;
            }
            const u32x4 hi = {h0, h1, h2, h3}, lo = {l0, l1, l2, l3};
            u32x4* o = reinterpret_cast<u32x4*>(wfrag) + ((((size_t)r * nts + nt) * M + m) * 2) * 64 + kg * 16 + n16;
            o[0] = hi;
            o[64] = lo;
        }
    }
}


}  // namespace gn_rw
