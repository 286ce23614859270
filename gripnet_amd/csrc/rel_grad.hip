// Weight gradient of the relational layer (autograd of myRGCN.forward, gripnet/layers.py:165-197, as the training loop
// of GripNet-pose.py:140-146 runs it): for P = sum_e x[src_e] W_{r(e)} and gm = dL/dP,
//
//   dW_r = sum_{e in r} x[src_e]^T gm[dst_e] = X^T Q_r,      Q_r[s,:] = sum_{e in r, src_e = s} gm[dst_e,:]
//
// ONE launch instead of the (relation, source) sums written out ([R n, out] = 80 MB on PoSE: 54 us) and a batched library
// GEMM over them (40 us): a workgroup keeps gm in LDS (n x out floats: 82 KB), a wave takes a UNIT = four chunks of up to
// eight edges of one (relation, source) row each - lane (kg = lane >> 4, c = lane & 15) sums columns c and 16 + c of the
// gm rows that chunk kg's ids name, which IS the B operand of v_mfma_f32_16x16x4_f32 (k = chunk, n = column) - and
// multiplies by the A operand x[source of chunk kg][16 t + c].  Q never exists in memory.  Every unit costs the same and
// its addresses do not depend on anything loaded before (host_layout.hpp), so ids and x rows are requested four units
// ahead into four fixed register sets (no register is moved while a load into it is in flight).  fp32 MFMA: every product
// is an fp32 FMA, as in the library GEMM.  The sixteen waves of a workgroup split a relation's units; their accumulators
// meet in LDS and are added in wave order; a relation too large for one workgroup is cut into parts whose sums the last
// part to arrive adds in part order (write-through stores, atomic ticket, sc1 loads): no float atomics, the same bits every
// run.
#include "common.h"

#include <type_traits>

struct gn_rel_grad_plan {
    int64_t num_nodes = 0, num_relations = 0, edges = 0;
    int groups = 0, scratch_slots = 0, entries = 0;
    int64_t units = 0;
    gn::DevBuf<uint16_t> src, ids;
    gn::DevBuf<int32_t> entry, wave_cnt, wg_off, wave_u0;
    gn::DevBuf<float> scratch;             // [scratch_slots][kRelMaxOutputs] partial dW of the parts of split relations
    gn::DevBuf<uint32_t> ticket;           // [num_relations] parts arrived (left at zero by every launch)
};

namespace {

constexpr int kRelThreads = 1024;
constexpr int kRelMaxOutputs = 64 * 32;    // in x out of the largest supported layer
constexpr int kLdsBytes = 160 * 1024;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct RelArgs {
    const float* x; uint32_t ld_x;
    const float* gm; int64_t ld_gm;
    int n;
    const uint16_t* src;                    // [units][4]
    const uint16_t* ids;                    // [units][4][8]
    const int32_t* entry; const int32_t* wave_cnt; const int32_t* wg_off; const int32_t* wave_u0;
    float* dw;                              // [R][16 MT][16 NT]
    float* scratch; uint32_t* ticket;
};

#ifdef GN_STAMPS
// Diagnostic build only (make STAMPS=1): 100 MHz timestamps per workgroup (entry, table filled, first / last wave out of its
// units of the workgroup's LAST entry, exit) and per wave (loop time of every entry summed), never in the product library.
__device__ unsigned long long g_rel_stamps[256][8];
__device__ unsigned long long g_rel_wave[256][16];
#define GN_REL_STAMP(k) if (tid == 0 && blockIdx.x < 256) g_rel_stamps[blockIdx.x][k] = __builtin_amdgcn_s_memrealtime()
#else
#define GN_REL_STAMP(k)
#endif

template <int MT, int NT>
__global__ __launch_bounds__(kRelThreads) void k_rel_weight_grad(RelArgs a) {
    typedef float vec_t __attribute__((ext_vector_type(NT)));
    constexpr int TW = 16 * NT, FF = 256 * MT * NT;
    extern __shared__ float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t kg = lane >> 4, c = lane & 15;
    GN_REL_STAMP(0);
#ifdef GN_STAMPS
    if (tid == 0 && blockIdx.x < 256) { g_rel_stamps[blockIdx.x][2] = ~0ull; g_rel_stamps[blockIdx.x][3] = 0; }
    unsigned long long loop_time = 0;
#endif
    // ---- gm -> LDS, columns (c, 16 + c) side by side: a lane's two sums come from one ds_read_b64; row n is zero ----
    const int n_tab = a.n * TW;
    if ((a.ld_gm & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gm) & 15) == 0) {
        // 16-byte loads, six per thread in flight (a thread that fetches one float per trip waits out a round trip per float)
        constexpr int units = TW / 4;
        const int total = a.n * units;
        for (int base = 0; base < total; base += 6 * kRelThreads) {
            f32x4 v[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i = min(base + k * kRelThreads + tid, total - 1), d = i / units, j = i - d * units;
                v[k] = *reinterpret_cast<const f32x4*>(a.gm + (int64_t)d * a.ld_gm + 4 * j);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int i = base + k * kRelThreads + tid, d = i / units, j = i - d * units;
                if (i < total) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) lds[d * TW + ((4 * j + e) & 15) * NT + ((4 * j + e) >> 4)] = v[k][e];
                }
            }
        }
    } else {
        for (int i = tid; i < n_tab; i += kRelThreads) {
            const int d = i / TW, k = i - d * TW, cc = k / NT, ct = k - cc * NT;
            lds[i] = a.gm[(int64_t)d * a.ld_gm + 16 * ct + cc];
        }
    }
    if (tid < TW) lds[n_tab + tid] = 0.f;
    float* __restrict__ part = lds + (size_t)(a.n + 1) * TW;            // [16 waves][MT][256]: one column tile of every wave
    float* __restrict__ whole = part + gn_layout::kRelWaves * MT * 256;  // [16 MT][16 NT] the relation's sums, row-major
    int* last_part = reinterpret_cast<int*>(whole + FF);                 // (16 spare bytes behind them)
    __syncthreads();
    GN_REL_STAMP(1);

    // ---- four units in flight per wave, in four fixed register sets: ids and the A operand of unit u + 4 are requested
    //      when unit u has been consumed.  `idq` / `swq` point at the unit of set 0 of the current trip: every request is
    //      that pointer plus a constant.  The four sources of a unit are wave-uniform: they come through the scalar cache,
    //      one step ahead of the request that needs them (a vector load per lane group was a fifth load per unit, and the
    //      copy of its loop-carried result made the compiler wait for every outstanding load in every trip) ----
    const uint32_t u0 = (uint32_t)a.wave_u0[blockIdx.x * gn_layout::kRelWaves + wave];
    // (constant address space + a wave-uniform index: the compiler fetches through the scalar cache and keeps the lgkmcnt books)
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    typedef const __attribute__((address_space(4))) u32x2* scalar_words_t;
    scalar_words_t swq = (scalar_words_t)(uintptr_t)a.src + u0;
    const u32x4* __restrict__ idq = reinterpret_cast<const u32x4*>(a.ids) + (size_t)u0 * 4 + kg;
    const float* __restrict__ xc = a.x + c;
    const uint32_t c8 = c * (uint32_t)sizeof(vec_t);
    uint32_t src0, src1, src2, src3;
    u32x4 ids0, ids1, ids2, ids3;
    float av0[MT], av1[MT], av2[MT], av3[MT];
    // (W: the unit's four sources; K: unit of the request relative to the trip's first: P in the prologue, P + 4 in a trip)
#define GN_REL_REQUEST(P, K, W)                                                                                \
    do {                                                                                                       \
        const uint32_t w_ = kg >= 2u ? (W).y : (W).x;                                                          \
        src##P = (kg & 1u) ? w_ >> 16 : w_ & 0xffffu;                                                          \
        const float* __restrict__ xr_ = xc + (src##P == gn_layout::kRelNoSource ? 0u : src##P) * a.ld_x;      \
        _Pragma("unroll") for (int t = 0; t < MT; ++t) av##P[t] = xr_[16 * t];                                 \
        ids##P = idq[4 * (K)];                                                                                 \
    } while (0)
    {
        const u32x2 w0 = swq[0], w1 = swq[1], w2 = swq[2], w3 = swq[3];
        GN_REL_REQUEST(0, 0, w0); GN_REL_REQUEST(1, 1, w1); GN_REL_REQUEST(2, 2, w2); GN_REL_REQUEST(3, 3, w3);
    }
    u32x2 sw_next = swq[4];
    // Q of a unit: the lane's columns of the eight gm rows its chunk's ids name.  LDS byte address of row id = id * (16
    // sizeof(vec_t)) + c sizeof(vec_t): one v_mad_u32_u16 per id (op_sel picks the half of the word that holds the id).
    // Reads and adds are written out (left to the compiler the sum became a horizontal reduction over re-packed register
    // pairs, 8 moves and 2 extra adds per unit, behind address adds of the LDS base - which is 0: the kernel has no static
    // LDS) and in two halves: the reads of the NEXT unit are issued in front of the current unit's MFMAs, their sum is
    // taken behind them - the matrix pipe and the LDS round trip were the two long waits of a unit, and with everything
    // in program order a wave sat through them one after the other.  `gather_finish` counts on lgkmcnt: a scalar load the
    // compiler has in flight at that point can only make its waits longer (LDS reads return in order, so "at most k
    // outstanding" still means the first 8 - k reads are back).  Association: ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7)).
    typedef typename std::conditional<NT == 2, vec_t, float>::type reg_t;
    const uint32_t row_bytes = 16u * (uint32_t)sizeof(vec_t);
    reg_t p0, p1, p2, p3, p4, p5, p6, p7;
    auto gather_issue = [&](const u32x4& w) {
        uint32_t a0, a1, a2, a3, a4, a5, a6, a7;
        if constexpr (NT == 2) {
            asm volatile("v_mad_u32_u16 %8, %16, %20, %21\n\tv_mad_u32_u16 %9, %16, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "v_mad_u32_u16 %10, %17, %20, %21\n\tv_mad_u32_u16 %11, %17, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "v_mad_u32_u16 %12, %18, %20, %21\n\tv_mad_u32_u16 %13, %18, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "v_mad_u32_u16 %14, %19, %20, %21\n\tv_mad_u32_u16 %15, %19, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "ds_read_b64 %0, %8\n\tds_read_b64 %1, %9\n\tds_read_b64 %2, %10\n\tds_read_b64 %3, %11\n\t"
                         "ds_read_b64 %4, %12\n\tds_read_b64 %5, %13\n\tds_read_b64 %6, %14\n\tds_read_b64 %7, %15"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                           "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
                         : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w), "v"(row_bytes), "v"(c8)
                         : "memory");
        } else {
            asm volatile("v_mad_u32_u16 %8, %16, %20, %21\n\tv_mad_u32_u16 %9, %16, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "v_mad_u32_u16 %10, %17, %20, %21\n\tv_mad_u32_u16 %11, %17, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "v_mad_u32_u16 %12, %18, %20, %21\n\tv_mad_u32_u16 %13, %18, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "v_mad_u32_u16 %14, %19, %20, %21\n\tv_mad_u32_u16 %15, %19, %20, %21 op_sel:[1,0,0,0]\n\t"
                         "ds_read_b32 %0, %8\n\tds_read_b32 %1, %9\n\tds_read_b32 %2, %10\n\tds_read_b32 %3, %11\n\t"
                         "ds_read_b32 %4, %12\n\tds_read_b32 %5, %13\n\tds_read_b32 %6, %14\n\tds_read_b32 %7, %15"
                         : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7),
                           "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7)
                         : "v"(w.x), "v"(w.y), "v"(w.z), "v"(w.w), "v"(row_bytes), "v"(c8)
                         : "memory");
        }
    };
    auto gather_finish = [&]() -> vec_t {
        if constexpr (NT == 2) {
            asm volatile("s_waitcnt lgkmcnt(6)\n\tv_pk_add_f32 %0, %0, %1\n\t"
                         "s_waitcnt lgkmcnt(4)\n\tv_pk_add_f32 %2, %2, %3\n\t"
                         "s_waitcnt lgkmcnt(2)\n\tv_pk_add_f32 %4, %4, %5\n\tv_pk_add_f32 %0, %0, %2\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_pk_add_f32 %6, %6, %7\n\tv_pk_add_f32 %4, %4, %6\n\t"
                         "v_pk_add_f32 %0, %0, %4\n\ts_nop 1"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : : "memory");
            return p0;
        } else {
            asm volatile("s_waitcnt lgkmcnt(6)\n\tv_add_f32 %0, %0, %1\n\t"
                         "s_waitcnt lgkmcnt(4)\n\tv_add_f32 %2, %2, %3\n\t"
                         "s_waitcnt lgkmcnt(2)\n\tv_add_f32 %4, %4, %5\n\tv_add_f32 %0, %0, %2\n\t"
                         "s_waitcnt lgkmcnt(0)\n\tv_add_f32 %6, %6, %7\n\tv_add_f32 %4, %4, %6\n\t"
                         "v_add_f32 %0, %0, %4\n\ts_nop 1"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : : "memory");
            return (vec_t)(p0);
        }
    };
    gather_issue(ids0);
    vec_t s_cur = gather_finish();                                         // Q of the wave's first unit
#define GN_REL_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, C, 0, 0, 0)
    // the unit of set P (its Q is in s_cur): the next unit's reads go out, the MFMAs, the set is refilled with the unit four
    // further on, the next unit's Q is summed
#define GN_REL_STEP(P, N)                                                                                      \
    do {                                                                                                       \
        gather_issue(ids##N);                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        const bool real_ = src##P != gn_layout::kRelNoSource;                                                  \
        _Pragma("unroll") for (int t = 0; t < MT; ++t) {                                                       \
            const float av_ = real_ ? av##P[t] : 0.f;                                                          \
            _Pragma("unroll") for (int ct = 0; ct < NT; ++ct)                                                  \
                acc[t][ct] = GN_REL_MFMA(av_, s_cur[ct], acc[t][ct]);                                          \
        }                                                                                                      \
        {                                                                                                      \
            const u32x2 sw_ = sw_next;                                                                         \
            sw_next = swq[(P) + 5];                                                                            \
            GN_REL_REQUEST(P, P + 4, sw_);                                                                     \
        }                                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                     \
        s_cur = gather_finish();                                                                               \
        __builtin_amdgcn_sched_barrier(0);      /* (the steps stay apart: merged, every load is consumed one step after its request) */ \
    } while (0)

    const int e_end = a.wg_off[blockIdx.x + 1];
    for (int e = a.wg_off[blockIdx.x]; e < e_end; ++e) {
        const int rel = a.entry[4 * e], parts = a.entry[4 * e + 1], part_index = a.entry[4 * e + 2], slot0 = a.entry[4 * e + 3];
        int left = a.wave_cnt[e * gn_layout::kRelWaves + wave];
        f32x4 acc[MT][NT];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int ct = 0; ct < NT; ++ct) acc[t][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifdef GN_STAMPS
        const unsigned long long t_in = __builtin_amdgcn_s_memrealtime();
#endif
        for (; left > 0; left -= gn_layout::kRelRing) {                    // (a multiple of four: the plan pads with empty units)
            GN_REL_STEP(0, 1); GN_REL_STEP(1, 2); GN_REL_STEP(2, 3); GN_REL_STEP(3, 0);
            idq += 4 * gn_layout::kRelRing; swq += gn_layout::kRelRing;
        }
#ifdef GN_STAMPS
        {
            const unsigned long long t_out = __builtin_amdgcn_s_memrealtime();
            loop_time += t_out - t_in;
            if (e == e_end - 1 && lane == 0 && blockIdx.x < 256) {
                atomicMin(&g_rel_stamps[blockIdx.x][2], t_out);
                atomicMax(&g_rel_stamps[blockIdx.x][3], t_out);
            }
        }
#endif
        // ---- the sixteen waves' accumulators meet in LDS, one column tile at a time, and are added in wave order; the
        //      relation's 16 MT x 16 NT sums are collected in LDS and leave as whole rows (a thread's own sums are a column of
        //      a tile: written straight out they would be 4-byte stores 128 bytes apart) ----
        float* __restrict__ dst = parts == 1 ? a.dw + (size_t)rel * FF : a.scratch + (size_t)(slot0 + part_index) * kRelMaxOutputs;
#pragma unroll
        for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
            for (int t = 0; t < MT; ++t) *reinterpret_cast<f32x4*>(part + ((size_t)(wave * MT + t) * 256 + lane * 4)) = acc[t][ct];
            __syncthreads();
#ifdef GN_STAMPS
            if (e == e_end - 1 && ct == 0) GN_REL_STAMP(6);
#endif
            if (tid < MT * 256) {
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < gn_layout::kRelWaves; ++w) v += part[w * MT * 256 + tid];
                // element i of lane l of tile t is row 16 t + 4 (l >> 4) + i, column l & 15 of the tile
                const int t = tid >> 8, l = (tid & 255) >> 2, i = tid & 3;
                whole[(16 * t + 4 * (l >> 4) + i) * TW + 16 * ct + (l & 15)] = v;
            }
            __syncthreads();
        }
#ifdef GN_STAMPS
        if (e == e_end - 1) GN_REL_STAMP(7);
#endif
        if (parts == 1) {
            if (tid < FF / 4) reinterpret_cast<f32x4*>(dst)[tid] = reinterpret_cast<const f32x4*>(whole)[tid];
        } else {
            // hand-over (the form of MI355X_MICROARCH.md for a few KB): every handed-over dword is stored write-through
            // (agent-scope relaxed atomic store = sc1), every storing wave drains its stores (an EXPLICIT s_waitcnt vmcnt(0): the
            // barrier alone waits for nothing that is in flight to memory) in front of the barrier, ONE lane draws the ticket; the part that draws the last one reads all parts with sc1 loads and adds them in part order.
            // No fence: a release here writes back the XCD's whole L2 (all the dW rows its workgroups have stored so far).
            uint32_t* __restrict__ mine = reinterpret_cast<uint32_t*>(dst);
            for (int o = tid; o < FF; o += kRelThreads) __hip_atomic_store(mine + o, __float_as_uint(whole[o]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // this wave's write-through stores have reached L2
            __syncthreads();
            if (tid == 0) {
                const uint32_t drawn = __hip_atomic_fetch_add(a.ticket + rel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *last_part = drawn == (uint32_t)parts - 1;
            }
            __syncthreads();
            if (*last_part) {
                const uint32_t* __restrict__ all = reinterpret_cast<const uint32_t*>(a.scratch + (size_t)slot0 * kRelMaxOutputs);
                for (int o = tid; o < FF; o += kRelThreads) {
                    // (eight parts requested before any is added - a load - add chain paid a memory round trip per part, and a
                    // hub relation has dozens; the order of the additions is unchanged)
                    float v = 0.f;
                    for (int p0 = 0; p0 < parts; p0 += 8) {
                        float q[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            q[k] = __uint_as_float(__hip_atomic_load(all + (size_t)min(p0 + k, parts - 1) * kRelMaxOutputs + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
#pragma unroll
                        for (int k = 0; k < 8; ++k) v += (p0 + k < parts) ? q[k] : 0.f;
                    }
                    a.dw[(size_t)rel * FF + o] = v;
                }
                if (tid == 0) __hip_atomic_store(a.ticket + rel, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
            __syncthreads();
        }
    }
#undef GN_REL_STEP
#undef GN_REL_MFMA
#undef GN_REL_REQUEST
    GN_REL_STAMP(4);
#ifdef GN_STAMPS
    if (lane == 0 && blockIdx.x < 256) g_rel_wave[blockIdx.x][wave] = loop_time;
    if (tid == 0 && blockIdx.x < 256) g_rel_stamps[blockIdx.x][5] = (unsigned long long)(e_end - a.wg_off[blockIdx.x]);
#endif
}

template <int MT, int NT>
gn_status launch(const gn_rel_grad_plan* p, const RelArgs& a, hipStream_t st) {
    const size_t lds = (size_t)(p->num_nodes + 1) * 16 * NT * 4 + (size_t)gn_layout::kRelWaves * MT * 1024 + (size_t)1024 * MT * NT + 16;
    if (lds > (size_t)kLdsBytes)
        return gn::fail(GN_ERR_UNSUPPORTED, "the gradient table of %lld nodes and the waves' sums need %zu bytes of LDS", (long long)p->num_nodes, lds);
    const gn_status ls = gn::allow_large_lds(reinterpret_cast<const void*>(k_rel_weight_grad<MT, NT>), kLdsBytes);
    if (ls != GN_OK) return ls;
    k_rel_weight_grad<MT, NT><<<p->groups, kRelThreads, lds, st>>>(a);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

}  // namespace

extern "C" {

gn_status gn_rel_grad_plan_create(const gn_graph_plan* sums, int64_t num_nodes, int64_t num_relations, void* stream,
                                  gn_rel_grad_plan** out) {
    GN_REQUIRE(sums && out, "plan pointer is null");
    *out = nullptr;
    GN_REQUIRE(num_nodes >= 1 && num_relations >= 1 && num_nodes * num_relations == sums->rows && sums->table_rows == num_nodes,
               "the sum plan must have one row per (relation, source): %lld x %lld rows over %lld nodes, got %lld rows over %lld",
               (long long)num_relations, (long long)num_nodes, (long long)num_nodes, (long long)sums->rows, (long long)sums->table_rows);
    if (num_nodes >= (int64_t)gn_layout::kRelNoSource)
        return gn::fail(GN_ERR_UNSUPPORTED, "more than 65534 nodes: ids do not fit 16 bits");
    hipStream_t st = gn::as_stream(stream);
    std::vector<int32_t> rowptr((size_t)sums->rows + 1), col((size_t)sums->nnz);
    GN_HIP(hipMemcpyAsync(rowptr.data(), sums->rowptr.p, rowptr.size() * 4, hipMemcpyDeviceToHost, st));
    if (!col.empty()) GN_HIP(hipMemcpyAsync(col.data(), sums->col.p, col.size() * 4, hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    const gn_layout::RelGradLayout L = gn_layout::build_rel_grad_layout(rowptr.data(), col.data(), num_nodes, num_relations, gn::compute_units());
    if (!L.ok) return gn::fail(GN_ERR_UNSUPPORTED, "the (relation, source) layout does not fit its index types");
    gn_rel_grad_plan* p = new gn_rel_grad_plan();
    p->num_nodes = num_nodes; p->num_relations = num_relations; p->edges = sums->nnz;
    p->groups = L.groups; p->scratch_slots = L.scratch_slots; p->entries = (int)(L.entry.size() / 4);
    p->units = L.units;
    auto up = [&](auto& buf, const auto& v) -> hipError_t {
        hipError_t e = buf.alloc(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpyAsync(buf.p, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice, st);
    };
    hipError_t e = up(p->src, L.src);
    if (e == hipSuccess) e = up(p->ids, L.ids);
    if (e == hipSuccess) e = up(p->entry, L.entry);
    if (e == hipSuccess) e = up(p->wave_cnt, L.wave_cnt);
    if (e == hipSuccess) e = up(p->wg_off, L.wg_off);
    if (e == hipSuccess) e = up(p->wave_u0, L.wave_u0);
    if (e == hipSuccess) e = p->scratch.alloc((size_t)std::max(1, L.scratch_slots) * kRelMaxOutputs);
    if (e == hipSuccess) e = p->ticket.alloc((size_t)num_relations);
    if (e == hipSuccess) e = hipMemsetAsync(p->ticket.p, 0, (size_t)num_relations * 4, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);                   // the host vectors go out of scope
    if (e != hipSuccess) {
        gn_rel_grad_plan_destroy(p);
        return gn::fail(GN_ERR_HIP, "building the weight-gradient plan failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return GN_OK;
}

void gn_rel_grad_plan_destroy(gn_rel_grad_plan* p) {
    if (!p) return;
    p->src.release(); p->ids.release(); p->entry.release(); p->wave_cnt.release(); p->wg_off.release(); p->wave_u0.release();
    p->scratch.release(); p->ticket.release();
    delete p;
}

int gn_rel_weight_grad_supported(const gn_rel_grad_plan* plan, int64_t in_features, int64_t out_features) {
    if (!plan || gn::fast_paths_disabled()) return 0;
    if (in_features < 16 || in_features > 64 || in_features % 16 || !(out_features == 16 || out_features == 32)) return 0;
    const size_t lds = (size_t)(plan->num_nodes + 1) * out_features * 4 + (size_t)gn_layout::kRelWaves * (in_features / 16) * 1024 +
                       (size_t)in_features * out_features * 4 + 16;
    return lds <= (size_t)kLdsBytes ? 1 : 0;
}

gn_status gn_rel_weight_grad_f32(const gn_rel_grad_plan* plan, const float* x, int64_t ld_x, int64_t in_features,
                                 const float* gm, int64_t ld_gm, int64_t out_features, float* dw, void* stream) {
    GN_REQUIRE(plan != nullptr, "plan is null");
    GN_REQUIRE(x && gm && dw, "feature / gradient pointer is null");
    GN_REQUIRE((reinterpret_cast<uintptr_t>(dw) & 15) == 0, "dw must be 16-byte aligned");
    GN_REQUIRE(ld_x >= in_features && ld_gm >= out_features, "leading dimension smaller than the row length");
    GN_REQUIRE((plan->num_nodes + 1) * ld_x < ((int64_t)1 << 31), "x does not fit 32-bit element offsets");
    if (!gn_rel_weight_grad_supported(plan, in_features, out_features))
        return gn::fail(GN_ERR_UNSUPPORTED, "no fused weight gradient for %lld -> %lld features over %lld nodes",
                        (long long)in_features, (long long)out_features, (long long)plan->num_nodes);
    RelArgs a;
    a.x = x; a.ld_x = (uint32_t)ld_x; a.gm = gm; a.ld_gm = ld_gm; a.n = (int)plan->num_nodes;
    a.src = plan->src.p; a.ids = plan->ids.p;
    a.entry = plan->entry.p; a.wave_cnt = plan->wave_cnt.p; a.wg_off = plan->wg_off.p; a.wave_u0 = plan->wave_u0.p;
    a.dw = dw; a.scratch = plan->scratch.p; a.ticket = plan->ticket.p;
    hipStream_t st = gn::as_stream(stream);
    switch ((int)(in_features / 16) * 10 + (int)(out_features / 16)) {
        case 11: return launch<1, 1>(plan, a, st);
        case 21: return launch<2, 1>(plan, a, st);
        case 31: return launch<3, 1>(plan, a, st);
        case 41: return launch<4, 1>(plan, a, st);
        case 12: return launch<1, 2>(plan, a, st);
        case 22: return launch<2, 2>(plan, a, st);
        case 32: return launch<3, 2>(plan, a, st);
        default: return launch<4, 2>(plan, a, st);
    }
}

}  // extern "C"

#ifdef GN_STAMPS
extern "C" __attribute__((visibility("default"))) int gn_debug_read_rel_stamps(unsigned long long* host_out, unsigned long long* wave_out) {
    int e = (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rel_stamps), sizeof(unsigned long long) * 256 * 8);
    if (e == 0) e = (int)hipMemcpyFromSymbol(wave_out, HIP_SYMBOL(g_rel_wave), sizeof(unsigned long long) * 256 * 16);
    return e;
}
#endif
