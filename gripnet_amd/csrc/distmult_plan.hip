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
};

namespace {

using namespace gn_dm;

constexpr int kNodeBits = 13;
constexpr uint32_t kNodeMask = (1u << kNodeBits) - 1;
constexpr uint32_t kNoMirror = 0xffffffffu;

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

// Deals the (up to) 64 edges of a batch to its slots.  Lane l of the wave holds slot l; wave step S works on the
// slots 4 q + S of the 16 quads q, and ds_read_b128 serves the quads in four access groups.  A cell = (step,
// access group) = four slots that hit the LDS together: its edges should have four different u % 4 and four
// different v % 4 (the bank slot of a row is (row * odd stride) % 4).
void deal_batch(const int64_t* u, const int64_t* v, int count, int* slot_of_edge) {
    static const int kGroupQuads[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    static const int kPerms[24][4] = {{0, 1, 2, 3}, {0, 1, 3, 2}, {0, 2, 1, 3}, {0, 2, 3, 1}, {0, 3, 1, 2}, {0, 3, 2, 1},
                                      {1, 0, 2, 3}, {1, 0, 3, 2}, {1, 2, 0, 3}, {1, 2, 3, 0}, {1, 3, 0, 2}, {1, 3, 2, 0},
                                      {2, 0, 1, 3}, {2, 0, 3, 1}, {2, 1, 0, 3}, {2, 1, 3, 0}, {2, 3, 0, 1}, {2, 3, 1, 0},
                                      {3, 0, 1, 2}, {3, 0, 2, 1}, {3, 1, 0, 2}, {3, 1, 2, 0}, {3, 2, 0, 1}, {3, 2, 1, 0}};
    std::vector<int> bucket[4][4];                         // edges by (u % 4, v % 4)
    for (int e = 0; e < count; ++e) bucket[u[e] & 3][v[e] & 3].push_back(e);
    int left = count;
    for (int cell = 0; cell < 16; ++cell) {
        const int S = cell & 3, g = cell >> 2;
        int chosen[4] = {-1, -1, -1, -1};
        if (left > 0) {
            // a full cell: one edge from each (c, sigma(c)) for the permutation whose scarcest bucket is fullest
            int best = -1, best_min = 0;
            for (int p = 0; p < 24; ++p) {
                int mn = 1 << 30;
                for (int c = 0; c < 4; ++c) mn = std::min(mn, (int)bucket[c][kPerms[p][c]].size());
                if (mn > best_min) { best_min = mn; best = p; }
            }
            if (best >= 0) {
                for (int c = 0; c < 4; ++c) { auto& bk = bucket[c][kPerms[best][c]]; chosen[c] = bk.back(); bk.pop_back(); }
            } else {
                // no conflict-free quadruple left: take edges one by one, preferring unused u and v classes
                unsigned used_u = 0, used_v = 0;
                for (int k = 0; k < 4; ++k) {
                    int bc = -1, bd = -1, bscore = -1;
                    for (int c = 0; c < 4; ++c)
                        for (int dd = 0; dd < 4; ++dd) {
                            if (bucket[c][dd].empty()) continue;
                            const int score = 2 * (!((used_u >> c) & 1) + !((used_v >> dd) & 1)) * 64 + (int)bucket[c][dd].size();
                            if (score > bscore) { bscore = score; bc = c; bd = dd; }
                        }
                    if (bc < 0) break;
                    chosen[k] = bucket[bc][bd].back();
                    bucket[bc][bd].pop_back();
                    used_u |= 1u << bc; used_v |= 1u << bd;
                }
            }
        }
        for (int k = 0; k < 4; ++k)
            if (chosen[k] >= 0) { slot_of_edge[chosen[k]] = 4 * kGroupQuads[g][k] + S; --left; }
    }
}

}  // namespace

extern "C" {

gn_status gn_distmult_plan_create(const int64_t* u, const int64_t* v, const int64_t* edge_type, int64_t num_edges,
                                  int64_t num_nodes, int64_t num_relations, void* stream, gn_distmult_plan** out) {
    GN_REQUIRE(out != nullptr, "plan output pointer is null");
    *out = nullptr;
    GN_REQUIRE(num_edges >= 0 && num_nodes >= 0 && num_relations >= 0, "negative size");
    GN_REQUIRE(num_edges == 0 || (u && v && edge_type), "edge pointers are null");
    if (num_nodes > (int64_t)kNodeMask + 1 || num_relations > 65535 || num_edges >= ((int64_t)1 << 31))
        return gn::fail(GN_ERR_UNSUPPORTED, "edge list too large for the packed plan encoding (nodes <= %u, relations <= 65535)",
                        kNodeMask + 1);
    hipStream_t st = gn::as_stream(stream);
    const int64_t E = num_edges;
    std::vector<int64_t> hu(E), hv(E), hr(E);
    if (E > 0) {
        GN_HIP(hipMemcpyAsync(hu.data(), u, E * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipMemcpyAsync(hv.data(), v, E * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipMemcpyAsync(hr.data(), edge_type, E * sizeof(int64_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipStreamSynchronize(st));
    }
    for (int64_t e = 0; e < E; ++e)
        if ((uint64_t)hu[e] >= (uint64_t)num_nodes || (uint64_t)hv[e] >= (uint64_t)num_nodes ||
            (uint64_t)hr[e] >= (uint64_t)num_relations)
            return gn::fail(GN_ERR_INDEX_RANGE, "edge %lld = (%lld, %lld, type %lld) is outside [0,%lld) x [0,%lld) x [0,%lld)",
                            (long long)e, (long long)hu[e], (long long)hv[e], (long long)hr[e], (long long)num_nodes,
                            (long long)num_nodes, (long long)num_relations);
    // Triples with the same unordered node pair and relation have the same score (the reference's positive list holds
    // every edge in both directions): they are paired up, the first of a pair is scored and writes both positions.
    std::vector<int64_t> mirror_of(E, -1);
    std::vector<char> covered(E, 0);
    {
        // open addressing on a power-of-two table (keys are unique per open triple; an erased slot keeps its key with
        // value -1 so that probe chains stay intact)
        size_t cap = 1;
        while (cap < (size_t)E * 2 + 16) cap <<= 1;
        std::vector<uint64_t> keys(cap, ~(uint64_t)0);
        std::vector<int64_t> vals(cap, -1);
        for (int64_t e = 0; e < E; ++e) {
            const uint64_t lo = (uint64_t)std::min(hu[e], hv[e]), hi = (uint64_t)std::max(hu[e], hv[e]);
            const uint64_t key = ((uint64_t)hr[e] << (2 * kNodeBits)) | (lo << kNodeBits) | hi;
            size_t h = (size_t)((key * 0x9E3779B97F4A7C15ull) >> 20) & (cap - 1);
            while (keys[h] != ~(uint64_t)0 && keys[h] != key) h = (h + 1) & (cap - 1);
            if (keys[h] == key && vals[h] >= 0) {                // the open copy of this triple: pair up
                mirror_of[vals[h]] = e;
                covered[e] = 1;
                vals[h] = -1;
            } else {                                            // first (or third, fifth, ...) copy: stays open
                keys[h] = key;
                vals[h] = e;
            }
        }
    }
    // batches: 64 consecutive scored edges each, in list order; a batch's slots are dealt independently of the others
    std::vector<int64_t> scored;
    scored.reserve((size_t)E);
    for (int64_t e = 0; e < E; ++e)
        if (!covered[e]) scored.push_back(e);
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
            deal_batch(cu, cv, count, slot_of_edge);
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
    gn_distmult_plan* p = new gn_distmult_plan();
    p->num_edges = E; p->num_nodes = num_nodes; p->num_relations = num_relations; p->batches = NB;
    auto bail = [&](hipError_t e) {
        gn_distmult_plan_destroy(p);
        return gn::fail(GN_ERR_HIP, "DistMult plan upload failed: %s", hipGetErrorString(e));
    };
    hipError_t he;
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

void gn_distmult_plan_destroy(gn_distmult_plan* p) {
    if (!p) return;
    p->packed.release();
    p->batch_rel.release();
    p->rel16.release();
    p->mirror.release();
    p->own.release();
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
