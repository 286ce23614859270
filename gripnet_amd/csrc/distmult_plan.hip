// DistMult decoder on a cached, re-encoded STATIC edge list (the positive edges of GripNet-pose.py:137,185:
// the same train_idx / train_et tensors every epoch; negative samples change per epoch and keep going
// through gn_distmult_forward_f32).
//
//   out[e] = sigma?( sum_k z[u_e,k] * z[v_e,k] * D[r_e,k] )                          (decoder.py:19-23)
//
// What the plan buys over streaming the raw int64 triples every call (distmult_fast.hip):
//   - work: triples with the same unordered node pair and relation have the same score, bit for bit (z_u z_v is
//     commutative) - the reference's positive list holds every edge in both directions (utils.py:132-138,168-198) -
//     so the plan pairs them up, scores one of a pair and writes its score to both positions: half the batches;
//   - HBM: 32-bit words per SCORED edge (u : 13 | v : 13; its position; its mirror's position, read in the last phase
//     only) and one relation word per batch, instead of 24 bytes per edge and column phase;
//   - LDS: inside a batch the edges are dealt to the (wave step, access group) cells so that the four edges a
//     16-lane ds_read_b128 access group works on have their u rows - and their v rows - in four different
//     64-byte bank slots wherever the batch allows it (a batch is 64 consecutive scored edges of the caller's list);
//   - VALU: no int64 arithmetic, no range checks (validated once, at plan time), the batch's relation is a
//     scalar.
// Same column phases and quad-per-edge arithmetic as the plan-less kernel: results are bitwise the same.
#include "distmult_quad.cuh"

#include <algorithm>
#include <unordered_map>
#include <vector>

struct gn_distmult_plan {
    int64_t num_edges = 0, num_nodes = 0, num_relations = 0, batches = 0;
    gn::DevBuf<uint32_t> packed;     // [batches * 64]
    gn::DevBuf<int32_t> batch_rel;   // [batches] relation of the batch, or -1 when it holds more than one
    gn::DevBuf<uint16_t> rel16;      // [batches * 64] relation of every slot (read for mixed batches only)
    gn::DevBuf<uint32_t> own;        // [batches * 64] position of the slot's edge in the caller's list
    gn::DevBuf<uint32_t> mirror;     // [batches * 64] second position that takes the slot's score, or kNoMirror
    // Row-class encoding (k_distmult_class): built when the caller names the feature count and a class's rows of ALL
    // its columns fit the LDS.  The nodes are cut into one or three blocks; a class holds the rows of one or two blocks,
    // a scored pair belongs to a class that holds both its endpoints, a workgroup serves one class and keeps that class's
    // rows - whole rows, every column - in LDS for the whole launch: one table fill, no column phases, no partial sums
    // parked between phases.  Steps of 16 pairs share a relation; four steps are a batch (one 32-bit word per lane).
    int cls_ok = 0, cls_features = 0, cls_groups = 0, cls_walks = 1;
    int64_t cls_batches = 0;
    gn::DevBuf<uint32_t> cls_packed;  // [(batches + slack) * 64] local row of u | local row of v << 16
    gn::DevBuf<uint32_t> cls_own;     // [(batches + slack) * 64] position of the slot's edge, or kNoMirror (padding)
    gn::DevBuf<uint32_t> cls_mirror;  // [(batches + slack) * 64]
    gn::DevBuf<uint32_t> cls_rel;     // [(batches + slack) * 2] relation of each of the batch's four steps, 16 bits each
    gn::DevBuf<int32_t> cls_wg;       // [groups][4 + 4 walks] first rows: start, count; second rows: start, count; per walk: batches lo, hi; relations lo, count
};

namespace {

using namespace gn_dm;

constexpr int kNodeBits = 13;
constexpr uint32_t kNodeMask = (1u << kNodeBits) - 1;
using gn_layout::kNoMirror;

struct DmPlanArgs {
    const float* z; int64_t ld_z; int n;
    const uint32_t* packed; const int32_t* batch_rel; const uint16_t* rel16; const uint32_t* own; const uint32_t* mirror;
    const float* d; int64_t ld_d;
    int64_t e; int64_t batches; int64_t batches_per_wg; int sigmoid; float* out;
    int n_phases; int c0[kMaxPhases]; int width[kMaxPhases];
    int stride4;
    int keep_n;                // batches per wave whose partial sums stay in LDS between the phases
    int all_kept;              // no wave has more batches than that
    int first_launch;          // the launch starts the sums (column 0 is its first column)
    int last_launch;           // the launch finishes them (sigmoid, mirror positions); otherwise raw partial sums go to `out`
};

// All indices are 32-bit here (E < 2^31 is a plan invariant) and everything that depends on the batch only is scalar.
template <int W4, int CPL>
__device__ __forceinline__ void run_phase(const DmPlanArgs& a, const char* lds, int stride_bytes, int c0, int w4, bool first,
                                          bool last, uint32_t b_lo, uint32_t b_hi, uint32_t step, int wave, int lane, float* keep,
                                          int keep_rd, int keep_wr) {
    // keep_rd / keep_wr: how many of this wave's batches take their carried-in sum from / leave their sum in the LDS
    // (the rest go through `out`); they differ from keep_n in the first phase of a launch that continues another
    // launch's sums and in the last phase of a launch that does not finish them
    const int l4 = lane & 3;
    const float* __restrict__ dcol = a.d + c0 + 4 * l4;
    const uint32_t* __restrict__ pk = a.packed + lane;
    const int32_t* __restrict__ brel = a.batch_rel;
    const uint32_t e32 = (uint32_t)a.e;
    int cur_r = -1;                                         // relation whose chunks sit in dreg
    f32x4 dreg[CPL];
#pragma unroll
    for (int i = 0; i < CPL; ++i) dreg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const uint32_t kWavesPerWg = step;                      // distance between two batches of this wave

    uint32_t b = b_lo + (uint32_t)wave;
    if (b >= b_hi) return;
    // Between two launches of the forward nothing of the edge stream stays in L2, and a wave has one batch of
    // work (a few hundred cycles) between loads: the packed words run three batches ahead of the arithmetic, the
    // relation word two, the partial sum of the previous phase one.
    auto clampb = [&](uint32_t x) { return x < b_hi ? x : b; };
    uint32_t w0 = pk[b * 64u], w1 = pk[clampb(b + kWavesPerWg) * 64u], w2 = pk[clampb(b + 2 * kWavesPerWg) * 64u];
    int r0 = brel[b], r1 = brel[clampb(b + kWavesPerWg)];
    // a batch = 64 consecutive SCORED edges of the caller's list (the others are mirrors, below): positions are explicit
    const uint32_t* __restrict__ own = a.own + lane;
    // (positions are read where they are used: in the last phase, and in the others by waves that have more batches than
    // partial sums fit the LDS; the mirror positions in the last phase only)
    const bool need_own = last || !a.all_kept, need_mir = last;
    uint32_t o0 = need_own ? own[b * 64u] : 0u, o1 = need_own ? own[clampb(b + kWavesPerWg) * 64u] : 0u;
    const uint32_t* __restrict__ mir = a.mirror + lane;
    uint32_t mnext = need_mir ? mir[b * 64u] : kNoMirror;
    // The partial sums of a wave's first keep_n batches wait for the next phase in LDS (`keep`: 256 bytes per batch,
    // beside the table) - the same wave works on the same batches in every phase; only batches beyond that go
    // through `out`.
    float cnext = (!first && keep_rd < 1 && o0 < e32) ? a.out[o0] : 0.f;
    int nb = 0;                                             // batch number of this wave (wave-uniform)
    for (; b < b_hi; b += kWavesPerWg, ++nb) {
        const uint32_t w = w0;
        const int rel = r0;
        const bool kept_rd = nb < keep_rd, kept_wr = nb < keep_wr;
        const float carried = first ? 0.f : (kept_rd ? keep[nb * 64] : cnext);
        const uint32_t bn = clampb(b + kWavesPerWg);
        const uint32_t mine = o0, mcur = mnext;
        w0 = w1; w1 = w2; r0 = r1; o0 = o1;
        w2 = pk[clampb(b + 3 * kWavesPerWg) * 64u];
        r1 = brel[clampb(b + 2 * kWavesPerWg)];
        if (need_own) o1 = own[clampb(b + 2 * kWavesPerWg) * 64u];
        if (need_mir) mnext = mir[bn * 64u];
        cnext = (!first && nb + 1 >= keep_rd && o0 < e32) ? a.out[o0] : 0.f;
        const int iu = (int)(w & kNodeMask), iv = (int)((w >> kNodeBits) & kNodeMask);
        const bool valid = mine < e32;
        float result = 0.f;
        if (rel >= 0) {                                       // wave-uniform
            if (rel != cur_r) {
                cur_r = rel;
#pragma unroll
                for (int i = 0; i < CPL; ++i)
                    dreg[i] = (l4 + 4 * i < w4) ? *reinterpret_cast<const f32x4*>(dcol + (int64_t)rel * a.ld_d + 16 * i)
                                                : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            quad_step<0, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, 0, dcol, a.ld_d, dreg, result);
            quad_step<1, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, 0, dcol, a.ld_d, dreg, result);
            quad_step<2, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, 0, dcol, a.ld_d, dreg, result);
            quad_step<3, W4, CPL, true>(lds, stride_bytes, w4, l4, iu, iv, 0, dcol, a.ld_d, dreg, result);
        } else {
            const int ir = (int)a.rel16[b * 64u + (uint32_t)lane];
            quad_step<0, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<1, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<2, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
            quad_step<3, W4, CPL, false>(lds, stride_bytes, w4, l4, iu, iv, ir, dcol, a.ld_d, dreg, result);
        }
        float total = carried + result;
        if (last) {
            if (a.sigmoid) total = sigmoid_f32(total);
            if (valid) a.out[mine] = total;
            // the same unordered node pair under the same relation elsewhere in the list (the reversed copy of a
            // bidirectional edge, utils.py:132-138): z_u z_v is commutative, so that score is this one, bit for bit
            if (mcur != kNoMirror) a.out[mcur] = total;
        } else if (kept_wr) {
            keep[nb * 64] = total;
        } else if (valid) {
            a.out[mine] = total;
        }
    }
}

#ifdef GN_STAMPS
// Diagnostic build only (make STAMPS=1): 100 MHz timestamps of wave 0 of every workgroup, never in the product library.
__device__ unsigned long long g_dm_stamps[256][12];
#define GN_DM_STAMP(k) if (tid == 0 && blockIdx.x < 256) g_dm_stamps[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime()
#else
#define GN_DM_STAMP(k)
#endif

__global__ __launch_bounds__(kThreads) void k_distmult_plan(DmPlanArgs a) {
    extern __shared__ float4 lds4[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // batches are dealt to the workgroups round-robin in groups of one per wave: workgroups that own one contiguous
    // range each finish a phase 6-8 us apart
    const uint32_t b_lo = (uint32_t)blockIdx.x * (kThreads / 64), b_hi = (uint32_t)a.batches;
    const uint32_t step = gridDim.x * (kThreads / 64);
    float* keep = reinterpret_cast<float*>(lds4 + (size_t)a.n * a.stride4) + (size_t)wave * a.keep_n * 64 + lane;
    const char* lds = reinterpret_cast<const char*>(lds4);

    GN_DM_STAMP(0);
    for (int ph = 0; ph < a.n_phases; ++ph) {
        const int c0 = a.c0[ph], w4 = a.width[ph] >> 2;
        if (ph > 0) __syncthreads();                        // everyone is done with the previous phase's table
        GN_DM_STAMP(1 + 3 * ph);
        switch (w4) {                                       // compile-time row width where it is a common one
            case 16: fill_table<16>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
            case 12: fill_table<12>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
            case 8: fill_table<8>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
            default: fill_table<0>(lds4, a.z, a.ld_z, a.n, c0, w4, a.stride4, tid); break;
        }
        __syncthreads();
        GN_DM_STAMP(2 + 3 * ph);
        const bool first = a.first_launch && ph == 0, last = a.last_launch && ph == a.n_phases - 1;
        const int keep_rd = ph == 0 ? 0 : a.keep_n;                       // (a first phase carries nothing, or takes it from `out`)
        const int keep_wr = (ph == a.n_phases - 1 && !a.last_launch) ? 0 : a.keep_n;
        switch (w4) {
            case 16: run_phase<16, 4>(a, lds, a.stride4 * 16, c0, w4, first, last, b_lo, b_hi, step, wave, lane, keep, keep_rd, keep_wr); break;
            case 12: run_phase<12, 3>(a, lds, a.stride4 * 16, c0, w4, first, last, b_lo, b_hi, step, wave, lane, keep, keep_rd, keep_wr); break;
            case 8: run_phase<8, 2>(a, lds, a.stride4 * 16, c0, w4, first, last, b_lo, b_hi, step, wave, lane, keep, keep_rd, keep_wr); break;
            case 4: run_phase<4, 1>(a, lds, a.stride4 * 16, c0, w4, first, last, b_lo, b_hi, step, wave, lane, keep, keep_rd, keep_wr); break;
            default: run_phase<0, 4>(a, lds, a.stride4 * 16, c0, w4, first, last, b_lo, b_hi, step, wave, lane, keep, keep_rd, keep_wr); break;
        }
        GN_DM_STAMP(3 + 3 * ph);
    }
}

// ---- row-class kernel --------------------------------------------------------------------------------------------
// One table fill, whole rows.  The column-phase kernel above holds ALL nodes and a slice of the columns, so a launch
// pays two fills, a hand-over between the phases (every wave waits for the slowest) and its per-batch bookkeeping
// twice; PMC passes of round 4 (profiles/r04_decoder_pmc.md) show it bound by VALU issue during the phases (6.0 M
// vector instructions per launch, ~390 per batch of 64 pairs where the arithmetic needs 80) and idle during the
// fills.  Here a workgroup holds a CLASS of rows with every column: a pair is scored in one go, the sum is split into
// the same column parts as the phases (so the bits are those of the plan-less kernel), the relation rows of D the
// workgroup's batches name sit in LDS next to the table, and nothing in the loop needs a clamp or a 64-bit address.
using gn_layout::kClsDCache;          // relation rows of D a workgroup keeps in LDS
using gn_layout::kClsSlack;           // readable batches behind the last one (the prefetches run ahead unclamped)

struct DmClassArgs {
    const float* z; int64_t ld_z;
    const float* d; int64_t ld_d;
    const uint32_t* packed; const uint32_t* own; const uint32_t* mirror; const uint32_t* rel; const int32_t* wg;
    float* out; int c0; int sigmoid, first_launch, last_launch;
    int walks;                 // batch ranges per workgroup: an XCD's workgroups walk its position sub-ranges one after the other
};

// One wave step: quad q scores the pair of slot 4 q + S.  J 16-byte chunks per lane (lane l4 of the quad holds chunks
// l4, l4 + 4, ...: the same columns per lane as the column phases), the first J1 of them are the first part of the sum.
template <int S, int J, int J1, int STRIDE>
__device__ __forceinline__ void class_step(const char* __restrict__ lds, uint32_t lane_off, uint32_t w, int l4,
                                           const f32x4 (&dreg)[J], float& res1, float& res2) {
    constexpr int kBcast = S * 0x55;     // quad_perm [S,S,S,S]
    const uint32_t ws = (uint32_t)dpp_i<kBcast>((int)w);
    const char* pu = lds + (__umul24(ws & 0xffffu, (uint32_t)STRIDE) + lane_off);
    const char* pv = lds + (__umul24(ws >> 16, (uint32_t)STRIDE) + lane_off);
    f32x4 P[J], Q[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
        P[j] = *reinterpret_cast<const f32x4*>(pu + 64 * j);
        Q[j] = *reinterpret_cast<const f32x4*>(pv + 64 * j);
    }
    f32x2 acc2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < J1; ++j) {
        const f32x2 lo = P[j].xy * Q[j].xy, hi = P[j].zw * Q[j].zw;
        acc2 = lo * dreg[j].xy + acc2;
        acc2 = hi * dreg[j].zw + acc2;
    }
    float acc = acc2.x + acc2.y;
    acc = dpp_add<0xB1>(acc);
    acc = dpp_add<0x4E>(acc);
    if (l4 == S) res1 = acc;
    if constexpr (J1 < J) {
        f32x2 bcc2 = {0.f, 0.f};
#pragma unroll
        for (int j = J1; j < J; ++j) {
            const f32x2 lo = P[j].xy * Q[j].xy, hi = P[j].zw * Q[j].zw;
            bcc2 = lo * dreg[j].xy + bcc2;
            bcc2 = hi * dreg[j].zw + bcc2;
        }
        float bcc = bcc2.x + bcc2.y;
        bcc = dpp_add<0xB1>(bcc);
        bcc = dpp_add<0x4E>(bcc);
        if (l4 == S) res2 = bcc;
    }
}

template <int J, int J1>
__global__ __launch_bounds__(kThreads) void k_distmult_class(DmClassArgs a) {
    extern __shared__ float4 lds4[];
    constexpr int ROW4 = 4 * J;                               // float4 per row of the launch's columns
    constexpr int STR4 = (J & 1) ? ROW4 : ROW4 + 4;           // LDS stride: an odd number of 64-byte bank slots
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int32_t* __restrict__ g = a.wg + (size_t)blockIdx.x * (4 + 4 * a.walks);
    const int r0s = g[0], r0c = g[1], r1s = g[2], r1c = g[3];
    // (the first walk's range is read with the rows - one scalar load of the descriptor's first 32 bytes - and every further
    // walk's range one walk ahead: read where it is used, it put a memory round trip in front of the first batches' requests)
    int nb_lo = g[4], nb_hi = g[5], nrel_lo = g[6], nnrel = g[7];
    const int rows = r0c + r1c;
    GN_DM_STAMP(0);
#ifdef GN_STAMPS
    if (tid == 0 && blockIdx.x < 256) g_dm_stamps[blockIdx.x][3] = 0;
#endif
    const bool first = a.first_launch, last = a.last_launch;
    constexpr uint32_t kStep = kThreads / 64;
    const uint32_t* __restrict__ pk = a.packed + lane;
    const uint32_t* __restrict__ own = a.own + lane;
    const uint32_t* __restrict__ mir = a.mirror + lane;
    const uint32_t* __restrict__ rel = a.rel + (lane & 1);
    const int l4 = lane & 3;
    const char* lds = reinterpret_cast<const char*>(lds4);
    const uint32_t lane_off = (uint32_t)l4 * 16u;
    float4* dfill = lds4 + (size_t)rows * STR4;
    const f32x4* __restrict__ dl = reinterpret_cast<const f32x4*>(dfill) + l4;
    // A workgroup walks its batch ranges one after the other (one per position sub-range of its XCD: the half-written
    // score lines of ONE sub-range are what the L2 has to keep); the table is filled once, the relation rows of D a
    // range's batches name are refilled per range.
    for (int walk = 0; walk < a.walks; ++walk) {
        const uint32_t b_lo = (uint32_t)nb_lo, b_hi = (uint32_t)nb_hi;
        const int rel_lo = nrel_lo, nrel = nnrel;
        if (walk + 1 < a.walks) { nb_lo = g[8 + 4 * walk]; nb_hi = g[9 + 4 * walk]; nrel_lo = g[10 + 4 * walk]; nnrel = g[11 + 4 * walk]; }
        // The first batches' words are requested BEFORE the fills (their HBM round trip hides behind them).  Every array
        // has kClsSlack readable batches behind the last one: no prefetch needs a clamp.  The relation words travel as
        // vector loads too (lanes 0 / 1 of `rw`): a scalar load shares its counter with the LDS reads and would be waited
        // for as soon as it is issued.
        uint32_t b = b_lo + (uint32_t)wave;
        uint32_t w0 = pk[b * 64u], w1 = pk[(b + kStep) * 64u], w2 = pk[(b + 2 * kStep) * 64u];
        uint32_t rw0 = rel[2 * b], rw1 = rel[2 * (b + kStep)];
        uint32_t o0 = own[b * 64u], o1 = own[(b + kStep) * 64u];
        uint32_t m0 = last ? mir[b * 64u] : kNoMirror, m1 = last ? mir[(b + kStep) * 64u] : kNoMirror;
        if (walk == 0) {
            // the class's rows, whole: eight 16-byte loads in flight per thread; every workgroup starts at its own offset
            const int total = rows * ROW4;
            const int rot = (int)((blockIdx.x * 977u) % (unsigned)total);
            for (int base = 0; base < total; base += 8 * kThreads) {
                float4 v[8];
                int at[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    int i = min(base + k * kThreads + tid, total - 1) + rot;
                    i = i < total ? i : i - total;
                    const int row = i / ROW4, c4 = i - row * ROW4;
                    const int grow = row < r0c ? r0s + row : r1s + (row - r0c);
                    at[k] = row * STR4 + c4;
                    v[k] = *reinterpret_cast<const float4*>(a.z + (int64_t)grow * a.ld_z + a.c0 + 4 * c4);
                }
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (base + k * kThreads + tid < total) lds4[at[k]] = v[k];
            }
        } else {
            __syncthreads();                                  // every wave is done with the previous range's relation rows
        }
        // the relation rows of D this range's batches name
        for (int i = tid; i < nrel * ROW4; i += kThreads) {
            const int r = i / ROW4, c4 = i - r * ROW4;
            dfill[i] = *reinterpret_cast<const float4*>(a.d + (int64_t)(rel_lo + r) * a.ld_d + a.c0 + 4 * c4);
        }
        __syncthreads();
        if (walk == 0) GN_DM_STAMP(1);
        if (b < b_hi) {
            float cnext = (!first && o0 != kNoMirror) ? a.out[o0] : 0.f;
            int cur = -1;
            f32x4 dreg[J];
#pragma unroll
            for (int j = 0; j < J; ++j) dreg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define GN_CLS_STEP(S, RELV)                                                                                           \
            {                                                                                                          \
                const int rs = (int)(RELV);                                                                            \
                if (rs != cur) {                                                                                       \
                    cur = rs;                                                                                          \
                    const f32x4* dr = dl + (rs - rel_lo) * ROW4;                                                       \
                    _Pragma("unroll") for (int j = 0; j < J; ++j) dreg[j] = dr[4 * j];                                 \
                }                                                                                                      \
                class_step<S, J, J1, STR4 * 16>(lds, lane_off, w, l4, dreg, res1, res2);                               \
            }
#pragma unroll 1
            for (; b < b_hi; b += kStep) {
                const uint32_t w = w0, mine = o0, mcur = m0;
                const uint32_t ra = (uint32_t)__builtin_amdgcn_readlane((int)rw0, 0), rb = (uint32_t)__builtin_amdgcn_readlane((int)rw0, 1);
                const float carried = cnext;
                w0 = w1; w1 = w2; rw0 = rw1; o0 = o1; m0 = m1;
                w2 = pk[(b + 3 * kStep) * 64u];
                rw1 = rel[2 * (b + 2 * kStep)];
                o1 = own[(b + 2 * kStep) * 64u];
                if (last) m1 = mir[(b + 2 * kStep) * 64u];
                cnext = (!first && o0 != kNoMirror) ? a.out[o0] : 0.f;
                float res1 = 0.f, res2 = 0.f;
                GN_CLS_STEP(0, ra & 0xffffu)
                GN_CLS_STEP(1, ra >> 16)
                GN_CLS_STEP(2, rb & 0xffffu)
                GN_CLS_STEP(3, rb >> 16)
                float total = carried + res1;                   // the column parts add up in the phases' order
                if constexpr (J1 < J) total = total + res2;
                if (last) {
                    if (a.sigmoid) total = sigmoid_f32(total);
                    if (mine != kNoMirror) a.out[mine] = total;
                    if (mcur != kNoMirror) a.out[mcur] = total;
                } else if (mine != kNoMirror) {
                    a.out[mine] = total;
                }
            }
#undef GN_CLS_STEP
        }
    }
    GN_DM_STAMP(2);
#ifdef GN_STAMPS
    if (lane == 0 && blockIdx.x < 256) atomicMax(&g_dm_stamps[blockIdx.x][3], __builtin_amdgcn_s_memrealtime());   // the workgroup's last wave
#endif
}

template <int J, int J1>
gn_status launch_class(const gn_distmult_plan* plan, const DmClassArgs& a, int64_t n, hipStream_t st) {
    gn_status s = gn::allow_large_lds(reinterpret_cast<const void*>(k_distmult_class<J, J1>), 160 * 1024);
    if (s != GN_OK) return s;
    const gn::LaunchEvents ev = gn::take_launch_events();
    if (ev.start || ev.stop) hipExtLaunchKernelGGL((k_distmult_class<J, J1>), dim3(plan->cls_groups), dim3(kThreads), 160 * 1024, st, ev.start, ev.stop, 0, a);
    else k_distmult_class<J, J1><<<plan->cls_groups, kThreads, 160 * 1024, st>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

namespace {

// The triples as the builders read them: 16 bits a value (nodes <= 2^13, relations < 2^16 - checked by the caller), narrowed
// and range-checked on the device so that a quarter of the bytes cross to the host and the host never walks the int64 lists.
// bad = the first edge outside the ranges (the smallest index), ~0 when every edge is inside.
__global__ __launch_bounds__(256) void k_narrow_triples(const int64_t* __restrict__ u, const int64_t* __restrict__ v,
                                                        const int64_t* __restrict__ r, int64_t E, int64_t n, int64_t R,
                                                        uint16_t* __restrict__ nu, uint16_t* __restrict__ nv,
                                                        uint16_t* __restrict__ nr, unsigned long long* __restrict__ bad) {
    unsigned long long first = ~0ull;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < E; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t a = u[e], b = v[e], c = r[e];
        if (((uint64_t)a >= (uint64_t)n || (uint64_t)b >= (uint64_t)n || (uint64_t)c >= (uint64_t)R) && (unsigned long long)e < first)
            first = (unsigned long long)e;
        nu[e] = (uint16_t)a; nv[e] = (uint16_t)b; nr[e] = (uint16_t)c;
    }
    if (first != ~0ull) atomicMin(bad, first);
}

template <typename V>
gn_status build_distmult_plan(const V& hu, const V& hv, const V& hr, int64_t E,
                              int64_t num_nodes, int64_t num_relations, int64_t num_features, hipStream_t st,
                              gn_distmult_plan** out) {
    gn::RawVec<int64_t> mirror_of;
    gn::RawVec<char> covered;
    gn_layout::pair_mirrors(hu, hv, hr, kNodeBits, mirror_of, covered);
    const gn::RawVec<int64_t> scored = gn_layout::scored_edges(covered);
    GN_LAP("decoder: scored list");
    gn_distmult_plan* p = new gn_distmult_plan();
    p->num_edges = E; p->num_nodes = num_nodes; p->num_relations = num_relations; p->batches = 0;
    auto bail = [&](hipError_t e) {
        gn_distmult_plan_destroy(p);
        return gn::fail(GN_ERR_HIP, "DistMult plan upload failed: %s", hipGetErrorString(e));
    };
    hipError_t he;
    // The row-class encoding first (round 6): when it exists, the column-phase encoding below is never read by a launch of the
    // decoder's own width (a call with columns the row-class kernel has no instantiation for is refused with GN_ERR_UNSUPPORTED
    // and the caller scores the raw list) - and its dealing was a quarter of the plan's build time.
    if (num_features > 0 && !gn::fast_paths_disabled()) {
        gn_layout::ClassLayout cl = gn_layout::build_class_layout(hu, hv, hr, scored, mirror_of, num_nodes, num_features, gn::compute_units());
        if (cl.ok) {
            GN_LAP(nullptr);
            if ((he = p->cls_packed.alloc(cl.packed.size())) != hipSuccess) return bail(he);
            if ((he = p->cls_own.alloc(cl.own.size())) != hipSuccess) return bail(he);
            if ((he = p->cls_mirror.alloc(cl.mirror.size())) != hipSuccess) return bail(he);
            if ((he = p->cls_rel.alloc(cl.rel32.size())) != hipSuccess) return bail(he);
            if ((he = p->cls_wg.alloc(cl.wg.size())) != hipSuccess) return bail(he);
            if ((he = hipMemcpyAsync(p->cls_packed.p, cl.packed.data(), cl.packed.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
            if ((he = hipMemcpyAsync(p->cls_own.p, cl.own.data(), cl.own.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
            if ((he = hipMemcpyAsync(p->cls_mirror.p, cl.mirror.data(), cl.mirror.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
            if ((he = hipMemcpyAsync(p->cls_rel.p, cl.rel32.data(), cl.rel32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
            if ((he = hipMemcpyAsync(p->cls_wg.p, cl.wg.data(), cl.wg.size() * sizeof(int32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
            if ((he = hipStreamSynchronize(st)) != hipSuccess) return bail(he);
            GN_LAP("decoder: allocations + upload (sync)");
            GN_LAP(nullptr);
            p->cls_features = (int)num_features; p->cls_groups = cl.groups; p->cls_batches = cl.batches; p->cls_walks = cl.walks;
            p->cls_ok = 1;
            *out = p;
            return GN_OK;
        }
    }
    // batches: 64 consecutive scored edges each, in list order; a batch's slots are dealt independently of the others
    const int64_t NBs = gn::ceil_div((int64_t)scored.size(), 64);
    std::vector<uint32_t> packed((size_t)NBs * 64), own((size_t)NBs * 64), mirror((size_t)NBs * 64);
    std::vector<int32_t> batch_rel((size_t)NBs);
    std::vector<uint16_t> rel16((size_t)NBs * 64);
    gn::parallel_for(NBs, 64, [&](int64_t b0, int64_t b1) {
        int slot_of_edge[64];
        int64_t cu[64], cv[64], ce[64];
        for (int64_t bi = b0; bi < b1; ++bi) {
            const int count = (int)std::min<int64_t>(64, (int64_t)scored.size() - bi * 64);
            for (int k = 0; k < count; ++k) { ce[k] = scored[bi * 64 + k]; cu[k] = hu[ce[k]]; cv[k] = hv[ce[k]]; }
            gn_layout::deal_batch(cu, cv, count, slot_of_edge);
            bool uniform = true;
            for (int k = 1; k < count; ++k) uniform = uniform && hr[ce[k]] == hr[ce[0]];
            const size_t s0 = (size_t)bi * 64;
            batch_rel[bi] = uniform ? (int32_t)hr[ce[0]] : -1;
            bool taken[64] = {false};
            auto fill = [&](int s, int k) {
                const int64_t e = ce[k];
                packed[s0 + s] = (uint32_t)hu[e] | ((uint32_t)hv[e] << kNodeBits);
                own[s0 + s] = (uint32_t)e;
                rel16[s0 + s] = (uint16_t)hr[e];
                mirror[s0 + s] = mirror_of[e] >= 0 ? (uint32_t)mirror_of[e] : kNoMirror;
            };
            for (int k = 0; k < count; ++k) { taken[slot_of_edge[k]] = true; fill(slot_of_edge[k], k); }
            // a slot without an edge repeats the batch's first one: the same score into the same positions
            for (int s = 0; s < 64; ++s)
                if (!taken[s]) fill(s, 0);
        }
    });
    const int64_t NB = (int64_t)batch_rel.size();
    p->batches = NB;
    if ((he = p->packed.alloc((size_t)NB * 64)) != hipSuccess) return bail(he);
    if ((he = p->batch_rel.alloc((size_t)NB)) != hipSuccess) return bail(he);
    if ((he = p->rel16.alloc((size_t)NB * 64)) != hipSuccess) return bail(he);
    if ((he = p->mirror.alloc((size_t)NB * 64)) != hipSuccess) return bail(he);
    if ((he = p->own.alloc((size_t)NB * 64)) != hipSuccess) return bail(he);
    if (NB > 0) {
        if ((he = hipMemcpyAsync(p->packed.p, packed.data(), (size_t)NB * 64 * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
        if ((he = hipMemcpyAsync(p->batch_rel.p, batch_rel.data(), (size_t)NB * sizeof(int32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
        if ((he = hipMemcpyAsync(p->rel16.p, rel16.data(), (size_t)NB * 64 * sizeof(uint16_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
        if ((he = hipMemcpyAsync(p->mirror.p, mirror.data(), (size_t)NB * 64 * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
        if ((he = hipMemcpyAsync(p->own.p, own.data(), (size_t)NB * 64 * sizeof(uint32_t), hipMemcpyHostToDevice, st)) != hipSuccess) return bail(he);
        if ((he = hipStreamSynchronize(st)) != hipSuccess) return bail(he);     // host vectors go out of scope after this
    }
    *out = p;
    return GN_OK;
}

}  // namespace

extern "C" {

gn_status gn_distmult_plan_create(const int64_t* u, const int64_t* v, const int64_t* edge_type, int64_t num_edges,
                                  int64_t num_nodes, int64_t num_relations, int64_t num_features, void* stream,
                                  gn_distmult_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(num_edges >= 0 && num_nodes >= 0 && num_relations >= 0, "negative size");
    GN_REQUIRE(num_edges == 0 || (u && v && edge_type), "edge pointers are null");
    if (num_nodes > (int64_t)kNodeMask + 1 || num_relations > 65535 || num_edges >= ((int64_t)1 << 31))
        return gn::fail(GN_ERR_UNSUPPORTED, "edge list too large for the packed plan encoding (nodes <= %u, relations <= 65535)",
                        kNodeMask + 1);
    hipStream_t st = gn::as_stream(stream);
    const int64_t E = num_edges;
    GN_LAP(nullptr);
#ifdef GN_LAYOUT_TIMES
    struct ExitLap { ~ExitLap() { GN_LAP("decoder: host vectors freed"); } } exit_lap;
#endif
    gn::ArenaHold arena;                                       // (before every host array of this build: host_layout.hpp)
    gn::RawVec<uint16_t> hu(E), hv(E), hr(E);
    GN_LAP("decoder: host vectors");
    if (E > 0) {
        gn::DevBuf<uint16_t> narrow;
        gn::DevBuf<unsigned long long> bad;
        auto give_up = [&](hipError_t e) {
            narrow.release(); bad.release();
            return gn::fail(GN_ERR_HIP, "DistMult plan: reading the edge list failed: %s", hipGetErrorString(e));
        };
        hipError_t he;
        if ((he = narrow.alloc((size_t)3 * E)) != hipSuccess) return give_up(he);
        if ((he = bad.alloc(1)) != hipSuccess) return give_up(he);
        unsigned long long first = ~0ull;
        if ((he = hipMemsetAsync(bad.p, 0xff, sizeof(unsigned long long), st)) != hipSuccess) return give_up(he);
        k_narrow_triples<<<gn::stream_grid(E, 256), 256, 0, st>>>(u, v, edge_type, E, num_nodes, num_relations, narrow.p, narrow.p + E,
                                                                  narrow.p + 2 * E, bad.p);
        if ((he = hipGetLastError()) != hipSuccess) return give_up(he);
        if ((he = hipMemcpyAsync(&first, bad.p, sizeof(first), hipMemcpyDeviceToHost, st)) != hipSuccess) return give_up(he);
        if ((he = hipMemcpyAsync(hu.data(), narrow.p, E * sizeof(uint16_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return give_up(he);
        if ((he = hipMemcpyAsync(hv.data(), narrow.p + E, E * sizeof(uint16_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return give_up(he);
        if ((he = hipMemcpyAsync(hr.data(), narrow.p + 2 * E, E * sizeof(uint16_t), hipMemcpyDeviceToHost, st)) != hipSuccess) return give_up(he);
        if ((he = hipStreamSynchronize(st)) != hipSuccess) return give_up(he);
        GN_LAP("decoder: narrow + D2H (sync)");
        narrow.release(); bad.release();
        GN_LAP("decoder: frees");
        if (first != ~0ull) {
            int64_t t[3] = {0, 0, 0};
            GN_HIP(hipMemcpy(&t[0], u + first, sizeof(int64_t), hipMemcpyDeviceToHost));
            GN_HIP(hipMemcpy(&t[1], v + first, sizeof(int64_t), hipMemcpyDeviceToHost));
            GN_HIP(hipMemcpy(&t[2], edge_type + first, sizeof(int64_t), hipMemcpyDeviceToHost));
            return gn::fail(GN_ERR_INDEX_RANGE, "edge %lld = (%lld, %lld, type %lld) is outside [0,%lld) x [0,%lld) x [0,%lld)",
                            (long long)first, (long long)t[0], (long long)t[1], (long long)t[2], (long long)num_nodes,
                            (long long)num_nodes, (long long)num_relations);
        }
    }
    return build_distmult_plan(hu, hv, hr, E, num_nodes, num_relations, num_features, st, out);
}

void gn_distmult_plan_destroy(gn_distmult_plan* p) {
    if (!p) return;
    p->packed.release();
    p->batch_rel.release();
    p->rel16.release();
    p->mirror.release();
    p->own.release();
    p->cls_packed.release();
    p->cls_own.release();
    p->cls_mirror.release();
    p->cls_rel.release();
    p->cls_wg.release();
    delete p;
}

int64_t gn_distmult_plan_edges(const gn_distmult_plan* plan) { return plan ? plan->num_edges : -1; }

}  // extern "C"

// Columns [col_lo, col_hi) of the feature dimension in one launch; the launch with col_lo = 0 starts the sums, the one
// with col_hi = num_features finishes them.
static gn_status plan_forward_cols(const gn_distmult_plan* plan, const float* z, int64_t ld_z, int64_t num_features, int64_t col_lo,
                                   int64_t col_hi, const float* d, int64_t ld_d, int apply_sigmoid, float* out, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    if (plan->num_edges == 0) return GN_OK;
    GN_REQUIRE(z && d && out, "operand pointer is null");
    GN_REQUIRE(num_features > 0 && ld_z >= col_hi && ld_d >= num_features, "feature count / leading dimension mismatch");
    GN_REQUIRE(0 <= col_lo && col_lo < col_hi && col_hi <= num_features && col_lo % 4 == 0, "column range outside the features");
    const int64_t n = plan->num_nodes, f = col_hi - col_lo;
    if (plan->cls_ok && f % 16 == 0 && f <= plan->cls_features && ld_z % 4 == 0 && ld_d % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d)) & 15) == 0 && !gn::fast_paths_disabled()) {
        // the sum keeps the column parts of the phase kernels (plan_phases), so the bits do not depend on the kernel
        int pc0[kMaxPhases], pw[kMaxPhases];
        const int parts = plan_phases(n, f, pc0, pw);
        const int J = (int)(f / 16), J1 = parts >= 1 ? pw[0] / 16 : 0;
        if ((parts == 1 || (parts == 2 && pw[0] % 16 == 0)) && J1 >= 1) {
            DmClassArgs c;
            c.z = z; c.ld_z = ld_z; c.d = d; c.ld_d = ld_d;
            c.packed = plan->cls_packed.p; c.own = plan->cls_own.p; c.mirror = plan->cls_mirror.p; c.rel = plan->cls_rel.p;
            c.wg = plan->cls_wg.p; c.out = out; c.c0 = (int)col_lo; c.sigmoid = apply_sigmoid;
            c.first_launch = col_lo == 0; c.last_launch = col_hi == num_features; c.walks = plan->cls_walks;
            hipStream_t st = gn::as_stream(stream);
            switch (J * 8 + J1) {
                case 5 * 8 + 3: return launch_class<5, 3>(plan, c, n, st);
                case 5 * 8 + 4: return launch_class<5, 4>(plan, c, n, st);
                case 4 * 8 + 4: return launch_class<4, 4>(plan, c, n, st);
                case 3 * 8 + 3: return launch_class<3, 3>(plan, c, n, st);
                case 2 * 8 + 2: return launch_class<2, 2>(plan, c, n, st);
                case 1 * 8 + 1: return launch_class<1, 1>(plan, c, n, st);
                default: break;                                               // other widths: the column-phase kernel
            }
        }
    }
    if (plan->cls_ok && plan->batches == 0)
        return gn::fail(GN_ERR_UNSUPPORTED, "the plan holds the row-class encoding only (built for %d features): no kernel of it covers columns [%lld, %lld): "
                                            "use gn_distmult_forward_f32", plan->cls_features, (long long)col_lo, (long long)col_hi);
    DmPlanArgs a;
    a.n_phases = (f % 4 == 0 && ld_z % 4 == 0 && ld_d % 4 == 0 &&
                  ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(d)) & 15) == 0)
                     ? plan_phases(n, f, a.c0, a.width) : 0;
    if (gn::fast_paths_disabled() || a.n_phases < 1 || a.n_phases > 4)
        return gn::fail(GN_ERR_UNSUPPORTED, "the planned decoder needs a node table that fits the LDS in at most four column "
                                            "phases (n = %lld, features = %lld): use gn_distmult_forward_f32", (long long)n, (long long)f);
    for (int k = 0; k < a.n_phases; ++k) a.c0[k] += (int)col_lo;
    a.first_launch = col_lo == 0;
    a.last_launch = col_hi == num_features;
    a.z = z; a.ld_z = ld_z; a.n = (int)n; a.packed = plan->packed.p; a.batch_rel = plan->batch_rel.p; a.rel16 = plan->rel16.p; a.own = plan->own.p; a.mirror = plan->mirror.p;
    a.d = d; a.ld_d = ld_d; a.e = plan->num_edges; a.batches = plan->batches; a.sigmoid = apply_sigmoid; a.out = out;
    int max_w = 0;
    for (int k = 0; k < a.n_phases; ++k) max_w = std::max(max_w, a.width[k]);
    a.stride4 = lds_stride4(n, max_w);
    const size_t table_bytes = (size_t)n * a.stride4 * 16;
    int64_t groups = std::min<int64_t>(256, gn::ceil_div(plan->batches, kThreads / 64));
    if (groups < 1) groups = 1;
    a.batches_per_wg = gn::ceil_div(plan->batches, groups);
    groups = gn::ceil_div(plan->batches, a.batches_per_wg);
    // partial sums between phases: as many batches per wave as the LDS left over by the table holds (4 KB each per workgroup)
    const int64_t per_wave = gn::ceil_div(plan->batches, groups * (kThreads / 64));
    const int64_t room = ((int64_t)160 * 1024 - 1024 - (int64_t)table_bytes) / ((kThreads / 64) * 256);
    a.keep_n = a.n_phases > 1 ? (int)std::max<int64_t>(0, std::min(per_wave, room)) : 0;
    // (a launch that takes sums from, or leaves them to, another launch needs the edge positions in those phases too)
    a.all_kept = a.n_phases > 1 && a.keep_n >= per_wave && a.first_launch && a.last_launch;
    const size_t lds_bytes = table_bytes + (size_t)a.keep_n * (kThreads / 64) * 256;
    { gn_status lds_status = gn::allow_large_lds(reinterpret_cast<const void*>(k_distmult_plan), 160 * 1024); if (lds_status != GN_OK) return lds_status; }
    k_distmult_plan<<<(unsigned)groups, kThreads, lds_bytes, gn::as_stream(stream)>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

extern "C" {

gn_status gn_distmult_plan_forward_f32(const gn_distmult_plan* plan, const float* z, int64_t ld_z, int64_t num_features,
                                       const float* d, int64_t ld_d, int apply_sigmoid, float* out, void* stream) {
    return plan_forward_cols(plan, z, ld_z, num_features, 0, num_features, d, ld_d, apply_sigmoid, out, stream);
}

gn_status gn_distmult_plan_forward_cols_f32(const gn_distmult_plan* plan, const float* z, int64_t ld_z, int64_t num_features,
                                            int64_t col_lo, int64_t col_hi, const float* d, int64_t ld_d, int apply_sigmoid,
                                            float* out, void* stream) {
    return plan_forward_cols(plan, z, ld_z, num_features, col_lo, col_hi, d, ld_d, apply_sigmoid, out, stream);
}

}  // extern "C"

#ifdef GN_STAMPS
extern "C" __attribute__((visibility("default"))) int gn_debug_read_dm_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dm_stamps), sizeof(unsigned long long) * 256 * 12);
}
#endif
