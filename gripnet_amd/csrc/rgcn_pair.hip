// Multi-relational internal layer (myRGCN, gripnet/layers.py:165-197) for small supervertices, destination-major in
// basis space.
//
//   out[i] * deg_i = sum_{e: dst=i} x[src_e] W_{r(e)},  W_r = sum_b att[r,b] basis[b]            (layers.py:172-189)
//                  = sum_b ( sum_s P_i[s,b] x[s,:] ) basis[b],   P_i[s,:] = sum_{e: s -> i} att[r(e),:]
//
// W_r is never formed.  What is gathered per edge is the 128-byte att row of its relation (the whole att table sits in
// LDS), summed per (destination, source) PAIR; what the matrix cores contract is the dense [bases x sources] block P_i
// of a destination with the node table x (K = every source node, the same K order for all destinations, so x is a shared
// MFMA operand that stays in registers); what is left per destination is U_i [bases x in], contracted with basis in the
// epilogue together with the mean, the root term, the bias and the activation.  A workgroup owns its destination rows
// outright: no partial sums cross workgroups, no slabs, no finalisation launch, no atomics; results are bitwise
// reproducible.
//
// Lane mapping of the gather = the A-operand layout of v_mfma_f32_16x16x32_bf16: lane (c = lane & 15, kg = lane >> 4)
// holds bases {BT*c .. BT*c+BT-1} of the eight sources k = 8*kg + t, t = 0..7, of a 32-source chunk.  The four lane
// groups run four (destination, source) pairs in lock step; a pair's edges come in blocks of four (one 32-bit word per
// lane quad position, rotated through the quad with DPP so that one ds_read_b32 of the stream serves four steps).
// The products run as bf16 MFMAs on operands split into three bf16 terms (x = hi + mid + lo exactly): six products
// hi.hi, hi.mid, mid.hi, mid.mid, hi.lo, lo.hi in fp32 accumulators, dropping terms below 2^-24 of |p||x| - the same
// order as fp32's own rounding.  GN_RGCN_ARITH_FAST keeps two terms and three products (<= 2^-16 per product).
#include "common.h"

#include <rocprim/rocprim.hpp>

#include <numeric>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kWaves = gn_layout::kPairWaves, kThreads = kWaves * 64;   // four waves per SIMD: 128 registers each, one destination row per wave
constexpr int kRowBytes = gn_layout::kPairRowBytes;             // LDS stride of an att row (32 bases; fewer: zero padded)
constexpr int kRingBlocks = 32;            // per-wave window on its stream: 32 blocks of 64 bytes, refilled a quarter (512 B) at a time
constexpr int kRingBytes = kRingBlocks * 64;
constexpr int kMaxD = gn_layout::kPairMaxD;                   // destination rows per workgroup
constexpr int kMaxChunks = 255;            // chunks of 32 sources
constexpr int kSectionCap = gn_layout::kPairSectionCap;            // (<= 255: a descriptor holds a section's blocks in eight bits)
                                           // pair becomes several units (running sums stay short: fp32 chains of <= 64)
constexpr int kLdsBytes = 160 * 1024;
constexpr int kSlackBlocks = gn_layout::kPairSlackBlocks;          // readable blocks behind the last wave's stream (the window reads ahead)

struct PairArgs {
    const float* x;
    int64_t ld_x;
    const unsigned char* xp;             // x as bf16 split planes (gn_split_planes), or null: the kernel splits x itself
    const float* att;
    const float* basis;
    const float* root;
    const float* bias;
    const float* indeg;
    float* out;
    int64_t ld_out;
    const uint32_t* stream;
    const uint32_t* wave_first;          // [groups * 8] first block of a wave's stream
    const uint32_t* wave_units;          // [groups * 8] units of a wave
    const uint32_t* wave_desc;           // [groups * 16] first descriptor of a wave (32 dwords each, pages of two)
    const uint32_t* desc;
    const int32_t* wg_dst;
    int n, R, B, fout, chunks, relu, partial, att_dma;
    int basis_t = 0;                     // basis is stored [B][fout][fin] (GN_RGCN_BASIS_TRANSPOSED: the backward's W_r^T from the forward's own parameter)
    gn_side_copy side;
    float* psum = nullptr;               // MODE 1 writes, MODE 2 reads: the pair sums of every unit, [unit slot][4][64 lanes] 16-byte words
};

// MODE of k_rgcn_pair (round 6): 0 = the whole layer in one launch; 1 = only the x-INDEPENDENT half - the gather of the att rows
// into the (destination, source) pair sums P, stored per unit exactly as the lanes hold them; 2 = only the x-dependent half - P
// read back by the same lanes of the same unit, contracted with x, epilogue.  1 then 2 give the bits of 0 (same sums, same
// products, same order): the split exists so that 1 can run on a side stream beside the layers that produce x.
constexpr int kModeFused = 0, kModeSums = 1, kModeContract = 2;

#ifdef GN_STAMPS
__device__ unsigned long long g_pair_stamps[2][256 * kWaves][12];
#endif

__device__ __forceinline__ uint32_t fbits(float v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ float bfloat(uint32_t v) { return __builtin_bit_cast(float, v); }

// (v1, v0) -> their bf16 terms, packed {v0 | v1 << 16} per term: v = hi + mid + lo exactly (three 8-bit pieces of the
// 24-bit significand, each cut by truncation, which keeps every remainder exact): 11 instructions per pair.
// (Tried: round-to-nearest conversions of the pair, v_cvt_pk_bf16_f32, with the remainders from v_dot2c_f32_bf16 - 7
// instructions per pair, exact on [1e-25, 1e37], tools/probes/split_probe.hip - but the accumulate-in-place form keeps
// more values alive: 52 bytes of scratch per lane at 128 registers and 36.0 instead of 33.3 us.  hipcc 7.2 also folds the
// packed constant (-1, 0) into the inline operand -1.0, which that instruction reads as (0, -1).)
__device__ __forceinline__ void split_pair(float v0, float v1, uint32_t& hi, uint32_t& mid, uint32_t& lo) {
    const uint32_t a0 = fbits(v0), a1 = fbits(v1);
    hi = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
    const float r0 = v0 - bfloat(a0 & 0xffff0000u), r1 = v1 - bfloat(a1 & 0xffff0000u);
    const uint32_t b0 = fbits(r0), b1 = fbits(r1);
    mid = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    const float s0 = r0 - bfloat(b0 & 0xffff0000u), s1 = r1 - bfloat(b1 & 0xffff0000u);
    lo = __builtin_amdgcn_perm(fbits(s1), fbits(s0), 0x07060302u);
}

template <int BT>
struct Acc;                                  // BT floats per lane and (destination, source) pair
template <>
struct Acc<1> { typedef float type; };
template <>
struct Acc<2> { typedef f32x2 type; };

// State of a wave's walk over its stream (uniform, except the lane constants).
struct Walk {
    uint32_t lane_off;    // LDS address of this lane's bases inside att row 0
    uint32_t ring_lane;   // LDS address of this lane's word inside block 0 of the wave's window
    uint32_t lane4;       // lane * 4 (a quarter of this lane's byte offset inside a 1 KB refill)
    uint32_t soff;        // byte offset inside the window of the next word to request
    uint32_t sdma;        // byte offset inside the wave's stream of the next refill
    uint32_t ring_base;   // LDS address of the window
    const uint32_t* sbase;   // the wave's stream
};

// ---- the gather of one unit, as ONE block of assembly (tools/gen_pair_asm.py writes it; the comment there explains
// the pipeline).  Written in assembly because the compiler, given the same sequence as separate statements, copies the
// destination registers of LDS requests still in flight and spends ~45 instructions per block on the control flow of
// the pipeline; this is 22.  Physical registers v92-v119 and s92-s95 belong to the block (clobbers). ----
#include "rgcn_pair_asm.inc"

#define GN_UNIT_OPERANDS(p)                                                                                            \
    : [p0] "=&v"(p[0]), [p1] "=&v"(p[1]), [p2] "=&v"(p[2]), [p3] "=&v"(p[3]), [p4] "=&v"(p[4]), [p5] "=&v"(p[5]),     \
      [p6] "=&v"(p[6]), [p7] "=&v"(p[7]), [soff] "+s"(w.soff), [sdma] "+s"(w.sdma)                                    \
    : [lo] "v"(w.lane_off), [rlane] "v"(w.ring_lane), [l4] "v"(w.lane4), [rbase] "s"(w.ring_base),                    \
      [sbase] "s"(w.sbase), [c03] "s"(c03), [c47] "s"(c47)                                                            \
    : "memory", "scc", "s92", "s93", "s94", "s95", GN_PAIR_ASM_CLOBBERS

// c03 / c47: the block counts of the unit's eight sections, eight bits each.
template <int BT>
__device__ __forceinline__ void gather_unit(Walk& w, uint32_t c03, uint32_t c47, typename Acc<BT>::type (&p)[8]) {
    if constexpr (BT == 2) asm volatile(GN_PAIR_UNIT_ASM_B64 GN_UNIT_OPERANDS(p));
    else asm volatile(GN_PAIR_UNIT_ASM_B32 GN_UNIT_OPERANDS(p));
}

template <int NT>
struct XRaw { float v[8][NT]; };             // x[source(8 kg + t)][NT c + j]: this lane's share of a chunk, fp32
template <int NT>
struct XFrag { uint32_t w[NT][3][4]; };      // the same as MFMA B operands: [N tile][term][8 bf16]

// The sources of the unit's chunk come with its descriptor (32 ids of 16 bits in dwords 8..23 of the unit's 32, lane
// group kg's eight in dwords 8 + 4 kg ..): the page was requested two units ago, so the rows of x can be requested at
// once - read from a table they were a dependent load in front of every unit (1.4 k cycles of each unit's 7.3 k).
template <int NT>
__device__ __forceinline__ void load_chunk(const PairArgs& a, uint32_t descv, int o, int kg, int c, XRaw<NT>& raw) {
    int32_t s[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = (uint32_t)__shfl((int)descv, o + 8 + 4 * kg + i);
        s[2 * i] = (int32_t)(w & 0xffffu);
        s[2 * i + 1] = (int32_t)(w >> 16);
    }
    // 32-bit element offsets from the uniform base (24-bit factors: a full-rate multiply-add instead of the quarter-rate
    // 64-bit address arithmetic, 16 + 8 slow instructions per unit: 34.6 -> 33.6 us, tools/ab_rgcn.sh)
    const uint32_t ld = (uint32_t)a.ld_x, col = (uint32_t)(NT * c);
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const bool ok = s[t] < a.n;
        const float* row = a.x + (__umul24(ok ? (uint32_t)s[t] : 0u, ld) + col);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const float v = row[j];
            raw.v[t][j] = ok ? v : 0.f;
        }
    }
}

template <int NT, int TERMS>
__device__ __forceinline__ void split_chunk(const XRaw<NT>& raw, XFrag<NT>& f) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t hi, mid, lo;
            split_pair(raw.v[2 * i][j], raw.v[2 * i + 1][j], hi, mid, lo);
            f.w[j][0][i] = hi;
            f.w[j][1][i] = mid;
            f.w[j][2][i] = TERMS == 3 ? lo : 0u;
        }
}

// ---- x from the split planes its producer left (gn_split_planes: per row 16 cells - lane c's columns NT c .. NT c +
// NT - 1 - of 3 NT bf16: term t of column j is half t NT + j).  A lane fetches the cells of its eight sources (20 bytes
// each at NT = 3) and packs pairs of sources into the B operands with one v_perm per operand dword: the 132 instructions
// of the in-kernel split, the range checks and the selects of load_chunk are gone (a source that is none names the zero row).
// The cells come in two requests.  The hi and mid terms (the first 2 NT halves of a cell: NT dwords, the registers the
// raw fp32 chunk took) are requested IN FRONT of the gather and land behind it, as the raw rows did.  The lo terms are
// requested behind the gather, into registers the gather has released, and are consumed by the LAST of the six products:
// P's split, the packing of hi / mid and thirty matrix instructions run while they travel.  (All of a cell behind the
// gather: the round trip showed - 3.3 k instead of 1.6 k cycles per unit in the contraction; all of it in front: 40
// registers across the gather, spills.)  The cells' byte offsets are recomputed from the descriptor for the second
// request rather than kept across the gather.
// idsel: byte address (lane * 4) of the descriptor dword that holds this lane group's first two ids; lane_b: the byte
// offset of this lane's cell inside a row.  (ds_bpermute takes the four dwords with an immediate offset each.)
template <int NT>
__device__ __forceinline__ void plane_offsets(uint32_t descv, uint32_t idsel, uint32_t lane_b, uint32_t (&off)[8]) {
    constexpr int CELLB = 4 * ((3 * NT + 1) / 2), ROWB = 16 * CELLB;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t w = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(idsel + 4u * i), (int)descv);
        off[2 * i] = __umul24(w & 0xffffu, (uint32_t)ROWB) + lane_b;
        off[2 * i + 1] = __umul24(w >> 16, (uint32_t)ROWB) + lane_b;
    }
}

template <int N>
__device__ __forceinline__ void load_dwords(const uint32_t* p, uint32_t (&d)[N]) {
    if constexpr (N == 4) { const u32x4 v = *reinterpret_cast<const u32x4*>(p); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; d[3] = v[3]; }
    else if constexpr (N == 3) { const u32x3 v = *reinterpret_cast<const u32x3*>(p); d[0] = v[0]; d[1] = v[1]; d[2] = v[2]; }
    else if constexpr (N == 2) { const u32x2 v = *reinterpret_cast<const u32x2*>(p); d[0] = v[0]; d[1] = v[1]; }
    else if constexpr (N == 1) { d[0] = p[0]; }
}

template <int NT>
struct XHm { uint32_t d[8][NT]; };                       // halves [0, 2 NT) of the eight cells: hi and mid
template <int NT>
struct XLo { uint32_t d[8][(NT + 1) / 2]; };             // halves [2 NT, 3 NT): lo

template <int NT>
__device__ __forceinline__ void load_planes_hm(const unsigned char* xp, uint32_t descv, uint32_t idsel, uint32_t lane_b, XHm<NT>& hm) {
    uint32_t off[8];
    plane_offsets<NT>(descv, idsel, lane_b, off);
#pragma unroll
    for (int t = 0; t < 8; ++t) load_dwords<NT>(reinterpret_cast<const uint32_t*>(xp + off[t]), hm.d[t]);
}

template <int NT>
__device__ __forceinline__ void load_planes_lo(const unsigned char* xp, uint32_t descv, uint32_t idsel, uint32_t lane_b, XLo<NT>& lo) {
    uint32_t off[8];
    plane_offsets<NT>(descv, idsel, lane_b, off);
#pragma unroll
    for (int t = 0; t < 8; ++t) load_dwords<(NT + 1) / 2>(reinterpret_cast<const uint32_t*>(xp + off[t]) + NT, lo.d[t]);
}

// B operand dwords of N tile jn, term t, from the cells of the eight sources: half k = t NT + jn of a cell
template <int NT>
__device__ __forceinline__ void pack_term(const XHm<NT>& hm, const XLo<NT>& lo, int t, int jn, uint32_t (&w)[4]) {
    const int k = t * NT + jn;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t a0 = t < 2 ? hm.d[2 * i][k >> 1] : lo.d[2 * i][(k - 2 * NT) >> 1];
        const uint32_t a1 = t < 2 ? hm.d[2 * i + 1][k >> 1] : lo.d[2 * i + 1][(k - 2 * NT) >> 1];
        w[i] = __builtin_amdgcn_perm(a1, a0, (k & 1) ? 0x07060302u : 0x05040100u);
    }
}

__device__ __forceinline__ bf16x8 as_frag(const uint32_t (&w)[4]) {
    u32x4 v = {w[0], w[1], w[2], w[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// acc[jm][jn] += P (bases x 32 sources) . X (32 sources x in) for one destination row and one chunk: six products on the
// three-term splits (three on two terms), the small ones first; hi(P) . lo(x) goes LAST - with the planes its operand is
// the one still travelling.  (Both forms of the kernel keep this order: they give the same bits.)
__device__ __forceinline__ f32x4 mfma(const bf16x8& a, const bf16x8& b, const f32x4& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

template <int BT>
__device__ __forceinline__ void split_p(const typename Acc<BT>::type (&p)[8], uint32_t (&t0)[BT][4], uint32_t (&t1)[BT][4], uint32_t (&t2)[BT][4]) {
#pragma unroll
    for (int jm = 0; jm < BT; ++jm)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v0, v1;
            if constexpr (BT == 2) { v0 = p[2 * i][jm]; v1 = p[2 * i + 1][jm]; } else { v0 = p[2 * i]; v1 = p[2 * i + 1]; }
            split_pair(v0, v1, t0[jm][i], t1[jm][i], t2[jm][i]);
        }
}

template <int NT, int BT, int TERMS>
__device__ __forceinline__ void contract(const typename Acc<BT>::type (&p)[8], const XFrag<NT>& xf, f32x4 (&acc)[BT][NT]) {
    uint32_t t0[BT][4], t1[BT][4], t2[BT][4];
    split_p<BT>(p, t0, t1, t2);
#pragma unroll
    for (int jm = 0; jm < BT; ++jm) {
        const bf16x8 ph = as_frag(t0[jm]), pm = as_frag(t1[jm]), pl = as_frag(t2[jm]);
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            const bf16x8 xh = as_frag(xf.w[jn][0]), xm = as_frag(xf.w[jn][1]);
            f32x4 c = acc[jm][jn];
            if constexpr (TERMS == 3) {
                c = mfma(pl, xh, c);
                c = mfma(pm, xm, c);
            }
            c = mfma(pm, xh, c);
            c = mfma(ph, xm, c);
            c = mfma(ph, xh, c);
            if constexpr (TERMS == 3) c = mfma(ph, as_frag(xf.w[jn][2]), c);
            acc[jm][jn] = c;
        }
    }
}

// The same with x from the planes: one N tile at a time - its operand terms are packed from the cells (four registers
// each) right in front of the products that use them, so the cells, P's terms and the accumulators fit the 128 registers.
template <int NT, int BT, int TERMS>
__device__ __forceinline__ void contract_planes(const typename Acc<BT>::type (&p)[8], const XHm<NT>& hm, const XLo<NT>& lo,
                                                f32x4 (&acc)[BT][NT]) {
    uint32_t t0[BT][4], t1[BT][4], t2[BT][4];
    split_p<BT>(p, t0, t1, t2);
#pragma unroll
    for (int jn = 0; jn < NT; ++jn) {
        __builtin_amdgcn_sched_barrier(0);                 // (keeps the packing of one tile from drifting in front of another's: registers)
        uint32_t xh4[4], xm4[4];
        pack_term<NT>(hm, lo, 0, jn, xh4);
        pack_term<NT>(hm, lo, 1, jn, xm4);
        const bf16x8 xh = as_frag(xh4), xm = as_frag(xm4);
#pragma unroll
        for (int jm = 0; jm < BT; ++jm) {
            const bf16x8 ph = as_frag(t0[jm]), pm = as_frag(t1[jm]), pl = as_frag(t2[jm]);
            f32x4 c = acc[jm][jn];
            if constexpr (TERMS == 3) {
                c = mfma(pl, xh, c);
                c = mfma(pm, xm, c);
            }
            c = mfma(pm, xh, c);
            c = mfma(ph, xm, c);
            c = mfma(ph, xh, c);
            acc[jm][jn] = c;
        }
    }
    if constexpr (TERMS == 3) {
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            __builtin_amdgcn_sched_barrier(0);
            uint32_t xl4[4];
            pack_term<NT>(hm, lo, 2, jn, xl4);
            const bf16x8 xl = as_frag(xl4);
#pragma unroll
            for (int jm = 0; jm < BT; ++jm) acc[jm][jn] = mfma(as_frag(t0[jm]), xl, acc[jm][jn]);
        }
    }
}

template <int NT, int BT, int TERMS, bool XP, int MODE = kModeFused>
__global__ __launch_bounds__(kThreads) void k_rgcn_pair(PairArgs a, int stamp_set) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    typedef typename Acc<BT>::type acc_t;
    constexpr int FIN = 16 * NT, BP = 16 * BT;                                 // padded bases
    // U of a destination in LDS: feature 16 jn + c (= column NT c + jn) at (16 jn + c) * BPP, bases innermost, four pad
    // floats per feature: the lanes c = 0..7 of a ds_write_b128 group are then BPP = 4 (mod 32) dwords apart - eight
    // different bank quads.  (Feature NT c + jn at (NT c + jn) * BP put all eight on ONE quad: 64 instead of 8 cycles per
    // store, 2.6 us of the epilogue.)
    constexpr int BPP = BP + 4, KP = BPP * FIN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x;
    const int c = lane & 15, kg = lane >> 4;
    const uint32_t lds0 = (uint32_t)(uintptr_t)lds;
    const uint32_t att_bytes = (uint32_t)(a.R + 2) * kRowBytes;
    const uint32_t ring0 = (att_bytes + 1023u) & ~1023u;
#ifdef GN_STAMPS
    const unsigned long long st0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- prologue: this wave's window on its stream, the att table ----
    const uint32_t wave_id = (uint32_t)(g * kWaves + wave);
    const uint32_t first_block = a.wave_first[wave_id], n_units = a.wave_units[wave_id];
    // byte offset of this wave's descriptors inside a.desc: uniform, 32 bits (a lane adds its own 4 bytes; the base stays
    // the kernel argument, so that the loads take the scalar-base form and no lane holds a 64-bit address)
    const uint32_t desc_b = __builtin_amdgcn_readfirstlane(a.wave_desc[wave_id] * 128u);
    const unsigned char* __restrict__ descp = reinterpret_cast<const unsigned char*>(a.desc);
    const unsigned char* __restrict__ xp = a.xp;
    const uint32_t lane4 = (uint32_t)lane * 4u;
    Walk w;
    w.ring_base = __builtin_amdgcn_readfirstlane(lds0 + ring0 + (uint32_t)wave * kRingBytes);
    {
        // the stream address as scalars (a saddr operand of the refill), whatever the compiler thinks of its uniformity
        const uint64_t sb = reinterpret_cast<uint64_t>(a.stream + (size_t)first_block * 16);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)sb), hi = __builtin_amdgcn_readfirstlane((uint32_t)(sb >> 32));
        w.sbase = reinterpret_cast<const uint32_t*>((uint64_t)hi << 32 | lo);
    }
    w.lane_off = lds0 + (uint32_t)c * (BT * 4);
    w.ring_lane = w.ring_base + (uint32_t)kg * 16u + (uint32_t)(lane & 3) * 4u;
    w.lane4 = (uint32_t)lane * 4u;
    w.soff = 256u;                                                             // a unit reads its first four words itself (behind soff)
    w.sdma = 3u * 512u;                                                        // the window starts with blocks 0..23
    if (MODE != kModeContract && lane < 32) {
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(w.sbase) + lane;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)q * 32),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(w.ring_base + q * 512u), 16, 0, 0);
    }
    if constexpr (MODE == kModeContract) {
        // (no att table, no stream window: this launch reads the pair sums)
    } else if (a.att_dma) {
        // rows of 32 bases are the LDS rows: 1 KB per wave instruction straight into LDS
        const int pieces = a.R / 8;
        const u32x4* __restrict__ src = reinterpret_cast<const u32x4*>(a.att);
        for (int i = wave; i < pieces; i += kWaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * 64 + lane),
                                             (__attribute__((address_space(3))) void*)(uintptr_t)(lds0 + (uint32_t)i * 1024u), 16, 0, 0);
        for (int e = pieces * 256 + tid; e < a.R * 32; e += kThreads)
            reinterpret_cast<float*>(lds)[e] = a.att[e];
    } else {
        for (int e = tid; e < a.R * 32; e += kThreads) {
            const int r = e >> 5, b = e & 31;
            reinterpret_cast<float*>(lds)[e] = b < a.B ? a.att[(size_t)r * a.B + b] : 0.f;
        }
    }
    // the workgroup's destination rows (read here, not in front of the epilogue's loads: one round trip less behind the loop)
    int32_t my_dst[4];
    {
        const int32_t* __restrict__ wd = a.wg_dst + (size_t)g * 4;
#pragma unroll
        for (int d = 0; d < 4; ++d) my_dst[d] = wd[d];
    }
    int nd = 0;
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) nd += my_dst[d] >= 0 ? 1 : 0;
    const uint32_t ranges = (uint32_t)my_dst[3];                               // first wave of rows 1 and 2 (eight bits each)
    if (MODE != kModeContract && tid < 64) reinterpret_cast<float*>(lds)[a.R * 32 + tid] = 0.f;        // the two zero rows padded slots name
    if (MODE != kModeSums && a.side.dst) {                                                          // concat slot 0, by the whole grid: its round
        const int64_t total = a.side.rows * a.side.cols;                       // trip hides behind the table fill (it was the kernel's last act)
        for (int64_t t = (int64_t)g * kThreads + tid; t < total; t += (int64_t)gridDim.x * kThreads) {
            const int64_t i = t / a.side.cols, cc = t - i * a.side.cols;
            const float v = a.side.src[i * a.side.ld_src + cc];
            a.side.dst[i * a.side.ld_dst + cc] = a.side.mode ? fabsf(v) : v;
        }
    }
    // unit descriptors of this wave, 32 dwords each (block counts, the chunk's 32 source ids): lane L holds dword L of the
    // current page of two units, and of the page after it
    uint32_t descv = n_units ? *reinterpret_cast<const uint32_t*>(descp + (desc_b + lane4)) : 0u;
    uint32_t descn = n_units > 2 ? *reinterpret_cast<const uint32_t*>(descp + (desc_b + 256u + lane4)) : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef GN_STAMPS
    const unsigned long long st1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long cyc_gather = 0, cyc_contract = 0, cyc_x = 0;
#endif

    f32x4 acc[BT][NT];                                                         // this wave's share of U of ITS destination row
#pragma unroll
    for (int jm = 0; jm < BT; ++jm)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[jm][jn] = (f32x4)(0.f);

    f32x4 pnext[2 * BT];                                                       // MODE 2: the pair sums of the next unit, in flight
    if constexpr (MODE == kModeContract) {
        if (n_units) {
            const f32x4* __restrict__ p0 = reinterpret_cast<const f32x4*>(a.psum) + (size_t)(desc_b >> 7) * (2 * BT * 64) + lane;
#pragma unroll
            for (int q = 0; q < 2 * BT; ++q) pnext[q] = p0[q * 64];
        }
    }
#pragma unroll 1
    for (uint32_t u = 0; u < n_units; ++u) {
        if (u && (u & 1u) == 0u) {                                             // next page: requested a page (two units) ago
            descv = descn;
            if (u + 2 < n_units) descn = *reinterpret_cast<const uint32_t*>(descp + (desc_b + ((u >> 1) + 1) * 256u + lane4));
        }
        const int o = (int)(u & 1u) * 32;
        const uint32_t c03 = __builtin_amdgcn_readlane(descv, o), c47 = __builtin_amdgcn_readlane(descv, o + 1);
        // this unit's chunk of x: requested here, split into bf16 terms behind the gather (the other waves of the SIMD
        // cover the L2 round trips)
#ifdef GN_STAMPS
        const unsigned long long cx0 = __builtin_amdgcn_s_memtime();
#endif
        XRaw<NT> raw;
        XHm<NT> hm;
        // (lane constants of the planes path, recomputed from lane * 4 where they are used: registers are what is short)
        const uint32_t idsel = ((lane4 >> 2) & 0x30u) + 32u + (uint32_t)o * 4u;      // dword 8 + 4 kg (+ the unit's half of the page)
        const uint32_t lane_b = (lane4 & 0x3cu) * (uint32_t)((3 * NT + 1) / 2);      // c * bytes of a cell
        acc_t p[8];
        // the unit's slot in the pair-sum buffer: its descriptor's index (unique per unit); a lane's 8 BT floats as 16-byte words
        // [word][lane]: a wave instruction moves 1 KB of consecutive bytes
        constexpr int PW = 2 * BT;                                                   // 16-byte words per lane and unit
        f32x4* __restrict__ pslot = reinterpret_cast<f32x4*>(a.psum) + ((size_t)(desc_b >> 7) + u) * (PW * 64) + lane;
        if constexpr (MODE == kModeSums) {
            gather_unit<BT>(w, c03, c47, p);
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                if constexpr (BT == 2) pslot[q * 64] = (f32x4){p[2 * q][0], p[2 * q][1], p[2 * q + 1][0], p[2 * q + 1][1]};
                else pslot[q * 64] = (f32x4){p[4 * q], p[4 * q + 1], p[4 * q + 2], p[4 * q + 3]};
            }
            continue;
        }
        if constexpr (MODE == kModeContract) {
            // this unit's sums were requested a unit ago; the next unit's are requested behind this unit's x cells (requests return
            // in order: the cells must not wait behind sums that are not needed yet)
#pragma unroll
            for (int q = 0; q < PW; ++q) {
                const f32x4 v = pnext[q];
                if constexpr (BT == 2) { p[2 * q] = (f32x2){v[0], v[1]}; p[2 * q + 1] = (f32x2){v[2], v[3]}; }
                else { p[4 * q] = v[0]; p[4 * q + 1] = v[1]; p[4 * q + 2] = v[2]; p[4 * q + 3] = v[3]; }
            }
        }
        if constexpr (XP) load_planes_hm<NT>(xp, descv, idsel, lane_b, hm); else load_chunk<NT>(a, descv, o, kg, c, raw);
#ifdef GN_STAMPS
        const unsigned long long cg0 = __builtin_amdgcn_s_memtime();
        cyc_x += cg0 - cx0;
#endif
        if constexpr (MODE != kModeContract) gather_unit<BT>(w, c03, c47, p);
#ifdef GN_STAMPS
        const unsigned long long cg1 = __builtin_amdgcn_s_memtime();
        cyc_gather += cg1 - cg0;
#endif
        if constexpr (XP) {
            XLo<NT> lo;
            if constexpr (TERMS == 3) load_planes_lo<NT>(xp, descv, idsel, lane_b, lo);
            if constexpr (MODE == kModeContract) {
                // the next unit's sums: requested BEHIND this unit's last x cells - requests return in order, and the products that
                // consume the cells must not wait for sums that are needed a unit later
                __builtin_amdgcn_sched_barrier(0);
                if (u + 1 < n_units) {
#pragma unroll
                    for (int q = 0; q < PW; ++q) pnext[q] = pslot[PW * 64 + q * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            contract_planes<NT, BT, TERMS>(p, hm, lo, acc);
        } else {
            XFrag<NT> xf;
            split_chunk<NT, TERMS>(raw, xf);
            contract<NT, BT, TERMS>(p, xf, acc);
        }
#ifdef GN_STAMPS
        asm volatile("s_nop 0" :: "v"(acc[0][0][0]) : "memory");
        cyc_contract += __builtin_amdgcn_s_memtime() - cg1;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                // window refills retire
    if constexpr (MODE == kModeSums) return;                                   // (no epilogue: the sums are in memory)
#ifdef GN_STAMPS
    const unsigned long long st2 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- epilogue ----
    // Rows of basis this thread will contract: thread = (four outputs o4, slice), rows j * slices + slice in basis' own
    // order (a wave instruction reads whole consecutive rows).  Requested BEFORE the barrier: a wave that finishes early
    // fetches while the others still gather; every compute unit pulls all of basis (196 KB on PoSE) through its own L2
    // port, which is what this epilogue costs.
    constexpr int kRowsMax = NT <= 2 ? 16 : 12;                                // rows of a slice held in registers (fewer accumulators: more rows)
    const int wb1 = (int)(ranges & 0xffu), wb2 = (int)((ranges >> 8) & 0xffu);
    const int fout = a.fout, og = fout >> 2;                                   // fout % 4 == 0, og in 1..16
    const int slices = kThreads / og;
    const int rows = a.B * FIN;                                                // rows of basis
    // (a transposed basis is contiguous along the FEATURE: there consecutive lanes take consecutive rows - slices - of one output
    // group, so that a wave's loads are whole lines again; which thread holds which (slice, output group) changes nothing else)
    const int o4 = a.basis_t ? tid / slices : tid % og;
    const int sl = a.basis_t ? (tid < slices * og ? tid - o4 * slices : slices) : tid / og;
    const int per = (rows + slices - 1) / slices;                              // <= kRowsMax (checked on the host)
    // (twelve rows are requested here, while this wave's accumulators are still alive; rows 12.. of the narrow layers - kRowsMax 16 -
    // behind the shares' store, where the accumulators are dead: with all sixteen alive next to them the NT <= 2 kernels spilled
    // 14-24 registers, 19 MB of scratch traffic per launch of the reversed layer of a training step)
    constexpr int kRowsPre = 12;
    f32x4 bv[kRowsMax];
    const f32x4* __restrict__ bp = reinterpret_cast<const f32x4*>(a.basis);
    // row nr = base * FIN + feature, outputs 4 o4 .. 4 o4 + 3: one 16-byte load, or - basis stored transposed, [B][fout][FIN]: the
    // reversed layer of a training step reads the forward's parameter as it is, a transposed copy was a launch of its own - four
    // loads FIN floats apart (196 KB, L2-resident)
    auto basis_row = [&](int nr) -> f32x4 {
        if (!a.basis_t) return bp[(uint32_t)(nr * og + o4)];
        const int b = nr / FIN, f = nr - b * FIN;
        const float* __restrict__ q = a.basis + ((size_t)b * fout + 4 * o4) * FIN + f;
        return (f32x4){q[0], q[FIN], q[2 * FIN], q[3 * FIN]};
    };
#pragma unroll
    for (int j = 0; j < kRowsPre; ++j) {
        const int nr = j * slices + sl;                                        // row base * FIN + feature of basis
        bv[j] = (sl < slices && j < per && nr < rows) ? basis_row(min(nr, rows - 1)) : (f32x4)(0.f);
    }
    // the output element this thread writes at the very end: its divisor, bias and share of the root term x_i . root
    // (eight adjacent lanes per element), requested before the barrier too
    const int fin_pair = tid >> 3, fin_part = tid & 7;
    const bool fin_live = fin_pair < nd * fout;
    const int fin_d = fin_live ? fin_pair / fout : 0, fin_o = fin_live ? fin_pair - fin_d * fout : 0;
    const int fin_i = fin_live ? my_dst[fin_d] : 0;
    float fin_div = 1.f, fin_bias = 0.f, fin_root = 0.f;
    if (fin_live && !a.partial) {
        fin_div = fmaxf(a.indeg[fin_i], 1.0f);
        if (a.bias) fin_bias = a.bias[fin_o];
#pragma unroll
        for (int f = 0; f < FIN / 8; ++f)
            fin_root += a.x[(int64_t)fin_i * a.ld_x + fin_part + 8 * f] * a.root[(fin_part + 8 * f) * fout + fin_o];
    }
    __syncthreads();                                                           // att table and windows are dead from here
#ifdef GN_STAMPS
    const unsigned long long st3 = __builtin_amdgcn_s_memrealtime();
#endif
    // U_i = sum over the waves of row i of their shares, through LDS: part[wave][k'] (bases innermost: the eight values
    // a lane holds of one feature, bases BT (4 kg + v) + jm, are 32 contiguous bytes)
    float* part = reinterpret_cast<float*>(lds);
    {
        float* dstp = part + (size_t)wave * KP;
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) {
            float* q = dstp + (16 * jn + c) * BPP + 4 * BT * kg;
            if constexpr (BT == 2) {
                *reinterpret_cast<f32x4*>(q) = (f32x4){acc[0][jn][0], acc[1][jn][0], acc[0][jn][1], acc[1][jn][1]};
                *reinterpret_cast<f32x4*>(q + 4) = (f32x4){acc[0][jn][2], acc[1][jn][2], acc[0][jn][3], acc[1][jn][3]};
            } else {
                *reinterpret_cast<f32x4*>(q) = acc[0][jn];
            }
            // the feature's pad floats are zero: they are summed like the rest, and a thread without a row of basis in
            // a pass (bv = 0) multiplies the LAST entry of U - a pad - by that zero
            if (kg == 3) *reinterpret_cast<f32x4*>(q + 4 * BT) = (f32x4)(0.f);
        }
    }
    if constexpr (kRowsMax > kRowsPre) {
        asm volatile("" ::: "memory");                                         // (behind the stores above: the accumulators are dead)
#pragma unroll
        for (int j = kRowsPre; j < kRowsMax; ++j) {
            const int nr = j * slices + sl;
            bv[j] = (sl < slices && j < per && nr < rows) ? basis_row(min(nr, rows - 1)) : (f32x4)(0.f);
        }
    }
    __syncthreads();
    // thread -> four consecutive k' of one row; the sums go behind the shares
    float* usum = part + (size_t)kWaves * KP;                                  // [nd][KP]
    for (int e = tid; e < nd * (KP / 4); e += kThreads) {
        const int d = e / (KP / 4), k4 = e - d * (KP / 4);
        const int w0 = d == 0 ? 0 : (d == 1 ? wb1 : wb2), w1 = d == 0 ? (nd > 1 ? wb1 : kWaves) : (d == 1 ? (nd > 2 ? wb2 : kWaves) : kWaves);
        // a row has a third of the sixteen waves on average: its shares are requested four at a time (a loop that
        // waited for every read before asking for the next paid an LDS round trip per wave), added in wave order
        f32x4 s = (f32x4)(0.f);
        for (int ww = w0; ww < w1; ww += 4) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(part + (size_t)min(ww + i, w1 - 1) * KP + 4 * k4);
#pragma unroll
            for (int i = 0; i < 4; ++i) s += (ww + i < w1) ? v[i] : (f32x4)(0.f);
        }
        *reinterpret_cast<f32x4*>(usum + (size_t)d * KP + 4 * k4) = s;
    }
    __syncthreads();
#ifdef GN_STAMPS
    const unsigned long long st4 = __builtin_amdgcn_s_memrealtime();
#endif
    // out_i = act( (U_i . basis) / max(1, deg_i) + x_i . root + bias ): row base * FIN + feature of basis meets entry
    // feature * BP + base of U
    f32x4 sum[kMaxD];
#pragma unroll
    for (int d = 0; d < kMaxD; ++d) sum[d] = (f32x4)(0.f);
    {
        int base = sl / FIN, feat = sl - base * FIN;                           // of row j * slices + sl, advanced by slices per step
        const int dbase = slices / FIN, dfeat = slices - dbase * FIN;
#pragma unroll
        for (int j = 0; j < kRowsMax; ++j) {
            const int k = min((16 * (feat % NT) + feat / NT) * BPP + base, KP - 1);
            // (all three rows, whether the workgroup has them or not: what a missing row's slot of usum holds is never
            // stored, and without the branches the compiler keeps the sums where they are - with them it moved the
            // twelve accumulator registers around at every merge: 700 moves per thread, 3 us of the epilogue)
#pragma unroll
            for (int d = 0; d < kMaxD; ++d) sum[d] += usum[(size_t)d * KP + k] * bv[j];
            base += dbase; feat += dfeat;
            if (feat >= FIN) { feat -= FIN; ++base; }
            // (sixteen rows: the LDS reads stay four rows at a time - hoisted in front of the products, their 48 values lived
            // next to all sixteen rows of basis and the narrow kernels spilled 14-24 registers)
            if constexpr (kRowsMax > 12) { if ((j & 3) == 3) asm volatile("" ::: "memory"); }
        }
    }
#ifdef GN_STAMPS
    asm volatile("" : "+v"(sum[0]));
    const unsigned long long st5 = __builtin_amdgcn_s_memrealtime();
#endif
    // [slices][kMaxD][og][4] = 12,288 floats: over the waves' shares (dead) where those are at least that long; narrow
    // layers (KP < 768) keep them behind the rows' sums instead - there the shares are shorter than `red`, and a thread
    // that has finished its contraction would write into the sums other threads still read
    // a slice's sums are four floats longer than they need to be: the eight lanes that fold one output read eight slices
    // that far apart - 4 (mod 32) banks, so the 32 lanes of a read hit 32 banks (unpadded: eight addresses on every bank)
    const int red_stride = kMaxD * og * 4 + 4;
    constexpr bool kRedOverShares = kWaves * KP >= kThreads * kMaxD * 4 + kThreads * 4;
    float* red = kRedOverShares ? part : usum + (size_t)kMaxD * KP;
    if (sl < slices)
#pragma unroll
        for (int d = 0; d < kMaxD; ++d)
            if (d < nd) *reinterpret_cast<f32x4*>(red + (size_t)sl * red_stride + ((size_t)d * og + o4) * 4) = sum[d];
    __syncthreads();
    // (destination, output) x eight partial sums over the slices, folded inside eight adjacent lanes: 128 pairs per pass
    // (three rows of up to 42 outputs are one pass; the values of the first pass were requested before the barriers)
    for (int p0 = 0; p0 < nd * fout; p0 += kThreads / 8) {
        const int pair = p0 + fin_pair;
        const bool lv = pair < nd * fout;
        const int pd = lv ? pair / fout : 0, po = lv ? pair - pd * fout : 0, pi = lv ? my_dst[pd] : 0;
        float div = fin_div, bias_v = fin_bias, t = fin_root;
        if (p0 > 0) {
            div = 1.f; bias_v = 0.f; t = 0.f;
            if (lv && !a.partial) {
                div = fmaxf(a.indeg[pi], 1.0f);
                if (a.bias) bias_v = a.bias[po];
#pragma unroll
                for (int f = 0; f < FIN / 8; ++f)
                    t += a.x[(int64_t)pi * a.ld_x + fin_part + 8 * f] * a.root[(fin_part + 8 * f) * fout + po];
            }
        }
        float s = 0.f;
        if (lv) {
            // (sixteen slices per lane at 32 outputs; requested eight at a time, added in slice order)
            for (int q0 = fin_part; q0 < slices; q0 += 64) {
                float rv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int q = min(q0 + 8 * i, slices - 1);
                    rv[i] = red[(size_t)q * red_stride + ((size_t)pd * og + (po >> 2)) * 4 + (po & 3)];
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s += (q0 + 8 * i < slices) ? rv[i] : 0.f;
            }
        }
        s += __shfl_xor(s, 1); t += __shfl_xor(t, 1);
        s += __shfl_xor(s, 2); t += __shfl_xor(t, 2);
        s += __shfl_xor(s, 4); t += __shfl_xor(t, 4);
        if (lv && fin_part == 0) {
            if (!a.partial) {
                s = s / div + t + bias_v;
                if (a.relu) s = fmaxf(s, 0.f);
            }
            a.out[(int64_t)pi * a.ld_out + po] = s;
        }
    }
#ifdef GN_STAMPS
    if (lane == 0 && g < 256) {
        unsigned long long* o = g_pair_stamps[stamp_set & 1][g * kWaves + wave];
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memrealtime();
        o[4] = cyc_gather; o[5] = cyc_contract; o[6] = cyc_x; o[7] = ((unsigned long long)n_units << 32) | (w.sdma >> 6);
        o[8] = st3; o[9] = st4; o[10] = st5;
    }
#endif
}

// ---- plan --------------------------------------------------------------------------------------------------------
// (a workgroup counts in LDS first: 2 M global atomics on 645 counters took 520 us)
__global__ void k_pair_outdeg(const int64_t* __restrict__ src, int64_t lo, int64_t hi, int n, int32_t* __restrict__ cnt) {
    extern __shared__ int32_t hist[];                                          // [n]
    for (int i = threadIdx.x; i < n; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (int64_t e = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < hi; e += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(hist + src[e], 1);                                           // ids validated by the general plan builder
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (hist[i]) atomicAdd(cnt + i, hist[i]);
}

__global__ void k_pair_keys(const int64_t* __restrict__ src, const int64_t* __restrict__ dst,
                            const int64_t* __restrict__ starts, int R, int64_t lo, int64_t hi, int kpad,
                            const int32_t* __restrict__ kpos, uint32_t* __restrict__ key, uint32_t* __restrict__ val) {
    for (int64_t e = lo + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < hi; e += (int64_t)gridDim.x * blockDim.x) {
        int a = 0, b = R;                                                      // relation of edge e: last start <= e
        while (b - a > 1) {
            const int mid = (a + b) >> 1;
            if (starts[mid] <= e) a = mid; else b = mid;
        }
        key[e - lo] = (uint32_t)(dst[e] * kpad + kpos[src[e]]);
        val[e - lo] = (uint32_t)a;
    }
}

__global__ void k_pair_rowptr(const uint32_t* __restrict__ sorted, int n, int64_t count, int32_t* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i > count) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (sorted[mid] < (uint32_t)i) lo = mid + 1; else hi = mid;
    }
    out[i] = lo;
}

using gn::Scratch;   // scoped device scratch (common.h)

int bits_for(int64_t n) {
    int b = 1;
    while (((int64_t)1 << b) < n) ++b;
    return b;
}

bool pair_disabled() { return gn::fast_paths_disabled(); }     // (a kernel is chosen by flag, GN_RGCN_PATH_*; GN_DISABLE_FAST=1 turns every fast path off)

template <int NT, int BT, int TERMS, int MODE>
gn_status launch_pair_mode(const gn_rgcn_plan* plan, const PairArgs& a, hipStream_t st) {
    constexpr bool XP = MODE == kModeContract;                  // (the sums' own launch reads no x at all; one instantiation per BT)
    gn_status s = gn::allow_large_lds(reinterpret_cast<const void*>(k_rgcn_pair<NT, BT, TERMS, XP, MODE>), kLdsBytes);
    if (s != GN_OK) return s;
    const gn::LaunchEvents ev = gn::take_launch_events();
    if (ev.start || ev.stop) hipExtLaunchKernelGGL((k_rgcn_pair<NT, BT, TERMS, XP, MODE>), dim3(plan->pair_groups), dim3(kThreads), kLdsBytes, st, ev.start, ev.stop, 0, a, 0);
    else k_rgcn_pair<NT, BT, TERMS, XP, MODE><<<plan->pair_groups, kThreads, kLdsBytes, st>>>(a, 0);
    GN_LAUNCH_CHECK();
    return GN_OK;
}

template <int NT, int BT, int TERMS>
gn_status launch_pair(const gn_rgcn_plan* plan, const PairArgs& a, hipStream_t st, int mode = kModeFused) {
    static int stamp = 0;
    if (mode == kModeSums) return launch_pair_mode<1, BT, 3, kModeSums>(plan, a, st);
    if (mode == kModeContract) {
        if constexpr (TERMS == 3) { if (a.xp) return launch_pair_mode<NT, BT, 3, kModeContract>(plan, a, st); }
        return gn::fail(GN_ERR_UNSUPPORTED, "the pair sums are contracted with x given as split planes, on three-term splits");
    }
    if (a.xp) {
        gn_status s = gn::allow_large_lds(reinterpret_cast<const void*>(k_rgcn_pair<NT, BT, TERMS, true>), kLdsBytes);
        if (s != GN_OK) return s;
        const gn::LaunchEvents ev = gn::take_launch_events();
        if (ev.start || ev.stop) hipExtLaunchKernelGGL((k_rgcn_pair<NT, BT, TERMS, true>), dim3(plan->pair_groups), dim3(kThreads), kLdsBytes, st, ev.start, ev.stop, 0, a, stamp++);
        else k_rgcn_pair<NT, BT, TERMS, true><<<plan->pair_groups, kThreads, kLdsBytes, st>>>(a, stamp++);
    } else {
        gn_status s = gn::allow_large_lds(reinterpret_cast<const void*>(k_rgcn_pair<NT, BT, TERMS, false>), kLdsBytes);
        if (s != GN_OK) return s;
        const gn::LaunchEvents ev = gn::take_launch_events();
        if (ev.start || ev.stop) hipExtLaunchKernelGGL((k_rgcn_pair<NT, BT, TERMS, false>), dim3(plan->pair_groups), dim3(kThreads), kLdsBytes, st, ev.start, ev.stop, 0, a, stamp++);
        else k_rgcn_pair<NT, BT, TERMS, false><<<plan->pair_groups, kThreads, kLdsBytes, st>>>(a, stamp++);
    }
    GN_LAUNCH_CHECK();
    return GN_OK;
}

template <int TERMS>
gn_status dispatch_pair(const gn_rgcn_plan* plan, const PairArgs& a, int nt, int bt, hipStream_t st, int mode = kModeFused) {
    switch (nt * 2 + (bt - 1)) {
        case 2: return launch_pair<1, 1, TERMS>(plan, a, st, mode);
        case 3: return launch_pair<1, 2, TERMS>(plan, a, st, mode);
        case 4: return launch_pair<2, 1, TERMS>(plan, a, st, mode);
        case 5: return launch_pair<2, 2, TERMS>(plan, a, st, mode);
        case 6: return launch_pair<3, 1, TERMS>(plan, a, st, mode);
        case 7: return launch_pair<3, 2, TERMS>(plan, a, st, mode);
        case 8: return launch_pair<4, 1, TERMS>(plan, a, st, mode);
        case 9: return launch_pair<4, 2, TERMS>(plan, a, st, mode);
    }
    return gn::fail(GN_ERR_UNSUPPORTED, "no destination-major relational kernel for these widths");
}

}  // namespace

// Builds the per-workgroup destination lists and per-wave streams of the shard.  Leaves plan->pair_ok = 0 when the graph
// does not qualify (too many nodes for three rows per compute unit, an att table beyond the LDS, nothing to do).
gn_status gn_rgcn_build_pair_plan(gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst,
                                  const std::vector<int64_t>& ranges, hipStream_t st) {
    plan->pair_ok = 0;
    const int64_t N = plan->num_nodes, R = plan->num_relations, E = plan->shard_edges;
    if (pair_disabled() || N < 1 || R < 1) return GN_OK;
    int cus = 256;
    {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            cus = prop.multiProcessorCount;
    }
    const int chunks = (int)gn::ceil_div(N, 32), kpad = chunks * 32;
    const int D = (int)gn::ceil_div(N, cus);
    if (D > kMaxD || chunks > kMaxChunks || chunks > 24) return GN_OK;            // up to 768 nodes
    const int64_t ring0 = ((R + 2) * kRowBytes + 1023) & ~(int64_t)1023;
    if (ring0 + kWaves * kRingBytes > kLdsBytes) return GN_OK;
    const int G = (int)std::min<int64_t>(N, cus);

    GN_LAP(nullptr);
    gn::ArenaHold arena;                                       // (before every host array of this build: host_layout.hpp)
    Scratch tmp;
    GN_HIP(tmp.reserve((size_t)24 * (size_t)E + (size_t)4 * (size_t)N * kpad + (size_t)8 * (size_t)(N + R) + ((size_t)1 << 20)));
    int64_t* starts_dev;
    int32_t *outdeg_dev, *kpos_dev, *rowptr_dev;
    uint32_t *key, *key_sorted, *val, *val_sorted;
    GN_HIP(tmp.get(&starts_dev, R + 1));
    GN_HIP(tmp.get(&outdeg_dev, N));
    GN_HIP(tmp.get(&kpos_dev, N));
    GN_HIP(tmp.get(&key, E));
    GN_HIP(tmp.get(&key_sorted, E));
    GN_HIP(tmp.get(&val, E));
    GN_HIP(tmp.get(&val_sorted, E));
    GN_HIP(tmp.get(&rowptr_dev, N * kpad + 1));
    // K order: sources by out-degree (inside the shard), largest first: lock-step partners expect similar runs
    std::vector<int32_t> outdeg(N, 0);
    GN_HIP(hipMemsetAsync(outdeg_dev, 0, N * sizeof(int32_t), st));
    if (E > 0) {
        k_pair_outdeg<<<gn::stream_grid(E, 256, 256), 256, (size_t)N * sizeof(int32_t), st>>>(src, plan->edge_lo, plan->edge_hi, (int)N, outdeg_dev);
        GN_LAUNCH_CHECK();
    }
    GN_HIP(hipMemcpyAsync(outdeg.data(), outdeg_dev, N * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("  pair: allocations + out-degrees (sync)");
    std::vector<int32_t> perm(kpad, (int32_t)N), kpos(N);
    {
        std::vector<int32_t> order(N);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return outdeg[x] > outdeg[y]; });
        for (int64_t k = 0; k < N; ++k) { perm[k] = order[k]; kpos[order[k]] = (int32_t)k; }
    }
    std::vector<int32_t> rp((size_t)N * kpad + 1, 0);
    std::vector<uint32_t> rels(E);
    if (E > 0) {
        std::vector<int64_t> starts(R + 1, plan->input_edges);
        for (int64_t r = 0; r < R; ++r) starts[r] = ranges[2 * r];
        GN_HIP(hipMemcpyAsync(starts_dev, starts.data(), (R + 1) * sizeof(int64_t), hipMemcpyHostToDevice, st));
        GN_HIP(hipMemcpyAsync(kpos_dev, kpos.data(), N * sizeof(int32_t), hipMemcpyHostToDevice, st));
        k_pair_keys<<<gn::stream_grid(E, 256), 256, 0, st>>>(src, dst, starts_dev, (int)R, plan->edge_lo, plan->edge_hi, kpad,
                                                             kpos_dev, key, val);
        GN_LAUNCH_CHECK();
        size_t bytes = 0;
        GN_HIP(rocprim::radix_sort_pairs(nullptr, bytes, key, key_sorted, val, val_sorted, (size_t)E, 0, bits_for(N * kpad), st));
        char* scratch = nullptr;
        GN_HIP(tmp.get(&scratch, bytes));
        GN_HIP(rocprim::radix_sort_pairs(scratch, bytes, key, key_sorted, val, val_sorted, (size_t)E, 0, bits_for(N * kpad), st));
        k_pair_rowptr<<<(int)gn::ceil_div(N * kpad + 1, 256), 256, 0, st>>>(key_sorted, (int)E, N * kpad, rowptr_dev);
        GN_LAUNCH_CHECK();
        GN_HIP(hipMemcpyAsync(rp.data(), rowptr_dev, ((size_t)N * kpad + 1) * sizeof(int32_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipMemcpyAsync(rels.data(), val_sorted, (size_t)E * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        GN_HIP(hipStreamSynchronize(st));
    }

    GN_LAP("  pair: keys + sort + D2H (sync)");
    gn_layout::PairLayout pl = gn_layout::build_pair_layout(N, R, chunks, kpad, G, D, rp, rels, perm);
    GN_LAP("  pair: host layout");
    if (!pl.ok) return GN_OK;
    gn::RawVec<uint32_t>& stream = pl.stream; std::vector<uint32_t>& first = pl.wave_first; std::vector<uint32_t>& desc = pl.desc;
    std::vector<uint32_t>& wave_units = pl.wave_units; std::vector<uint32_t>& wave_desc = pl.wave_desc;
    std::vector<int32_t>& wg_dst = pl.wg_dst;
    const size_t total = (size_t)pl.blocks * 16;
    GN_HIP(plan->pair_stream.alloc(stream.size()));
    GN_HIP(plan->pair_wave_first.alloc(first.size()));
    GN_HIP(plan->pair_desc.alloc(desc.size()));
    GN_HIP(plan->pair_wave_units.alloc(wave_units.size()));
    GN_HIP(plan->pair_wave_desc.alloc(wave_desc.size()));
    GN_HIP(plan->pair_wg_dst.alloc(wg_dst.size()));
    GN_HIP(hipMemcpyAsync(plan->pair_stream.p, stream.data(), stream.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->pair_wave_first.p, first.data(), first.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->pair_desc.p, desc.data(), desc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->pair_wave_units.p, wave_units.data(), wave_units.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->pair_wave_desc.p, wave_desc.data(), wave_desc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipMemcpyAsync(plan->pair_wg_dst.p, wg_dst.data(), wg_dst.size() * sizeof(int32_t), hipMemcpyHostToDevice, st));
    GN_HIP(hipStreamSynchronize(st));
    GN_LAP("  pair: upload (sync)");
    plan->pair_groups = G;
    plan->pair_unit_slots = (int64_t)desc.size() / 32 + 2;      // descriptors (one per unit; a wave's last page may be half empty)
    plan->pair_d = D;
    plan->pair_chunks = chunks;
    plan->pair_blocks = (int64_t)(total / 16);
    plan->pair_ok = 1;
    return GN_OK;
}

#ifdef GN_STAMPS
extern "C" GN_API int gn_debug_read_pair_stamps(unsigned long long* out) {   // diagnostic build only: [2048][8] of the last launch
    static int which = 0;
    (void)which;
    static unsigned long long both[2][256 * kWaves][12];
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(both, HIP_SYMBOL(g_pair_stamps), sizeof(both)) != hipSuccess) return 2;
    // the set written last: the one with the larger entry stamp
    const int set = both[0][0][0] > both[1][0][0] ? 0 : 1;
    memcpy(out, both[set], sizeof(both[0]));
    return 0;
}
#endif

bool gn_rgcn_pair_applicable(const gn_rgcn_plan* plan, int64_t fin, int64_t fout, int64_t bases) {
    if (!plan || !plan->pair_ok || pair_disabled()) return false;
    if (fin % 16 != 0 || fin < 16 || fin > 64 || bases < 1 || bases > 32 || fout % 4 != 0 || fout < 4 || fout > 64) return false;
    const int64_t nt = fin / 16, bt = (bases + 15) / 16;
    // the waves' shares of U in LDS: 8 waves x 3 rows x (16 bt x fin) floats, plus the slices' sums behind wave 0's
    const int64_t kp = (16 * bt + 4) * fin;                                    // (four pad floats per feature, see the kernel)
    const int64_t part = (int64_t)(kWaves + kMaxD) * kp * 4;                   // the waves' shares, the rows' sums
    const int64_t red = (int64_t)kThreads * kMaxD * 16 + (int64_t)kThreads * 16;   // the slices' sums (padded) reuse the shares' space
    const int64_t slices = kThreads / (fout / 4), per = (bases * fin + slices - 1) / slices;
    const int64_t need = (int64_t)kWaves * kp >= (int64_t)kThreads * kMaxD * 4 + kThreads * 4 ? std::max(part, red) : part + red;   // (narrow layers: behind the sums)
    return need <= kLdsBytes && nt * bt <= 6 && per <= (nt <= 2 ? 16 : 12);   // (rows of basis a thread holds in registers)
}

size_t gn_rgcn_pair_sums_bytes(const gn_rgcn_plan* plan, int64_t bases) {
    if (!plan || !plan->pair_ok) return 0;
    return (size_t)plan->pair_unit_slots * 64 * 16 * (size_t)(2 * ((bases + 15) / 16));
}

gn_status gn_rgcn_pair_forward(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t fin, const float* basis,
                               const float* att, int64_t bases, const float* root, const float* bias, int64_t fout,
                               int relu, int partial, int fast_arith, float* out, int64_t ld_out, const gn_side_copy& side,
                               const void* x_planes, hipStream_t st, int mode, void* pair_sums, int basis_transposed) {
    PairArgs a;
    a.basis_t = basis_transposed ? 1 : 0;
    a.psum = static_cast<float*>(pair_sums);
    GN_REQUIRE(mode == kModeFused || (pair_sums && (reinterpret_cast<uintptr_t>(pair_sums) & 15) == 0), "the pair sums need a 16-byte aligned buffer");
    a.xp = static_cast<const unsigned char*>(x_planes);
    GN_REQUIRE(ld_x < (1ll << 21), "x rows more than 2^21 floats apart are not supported by the destination-major kernel");
    a.x = x; a.ld_x = ld_x; a.att = att; a.basis = basis; a.root = root; a.bias = bias;
    a.indeg = plan->indeg.p;
    a.out = out; a.ld_out = ld_out;
    a.stream = plan->pair_stream.p; a.wave_first = plan->pair_wave_first.p; a.desc = plan->pair_desc.p;
    a.wave_units = plan->pair_wave_units.p; a.wave_desc = plan->pair_wave_desc.p;
    a.wg_dst = plan->pair_wg_dst.p;
    a.n = (int)plan->num_nodes; a.R = (int)plan->num_relations; a.B = (int)bases; a.fout = (int)fout;
    a.chunks = plan->pair_chunks; a.relu = relu; a.partial = partial;
    a.att_dma = bases == 32 && (reinterpret_cast<uintptr_t>(att) & 15) == 0;
    a.side = side;
    const int nt = (int)(fin / 16), bt = (int)((bases + 15) / 16);
    if (mode != kModeFused && fast_arith) return gn::fail(GN_ERR_UNSUPPORTED, "the split launches run the default arithmetic");
    return fast_arith ? dispatch_pair<2>(plan, a, nt, bt, st) : dispatch_pair<3>(plan, a, nt, bt, st, mode);
}
