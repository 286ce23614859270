// Shared host-side helpers for the gfx950 supergraph propagation library.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "gripnet_hip.h"
#include "host_layout.hpp"

namespace gn {

constexpr int kWave = 64;  // CDNA wavefront width

// ---- per-thread error message ------------------------------------------------------------
inline char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

inline gn_status fail(gn_status code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define GN_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess)                                                                     \
            return gn::fail(GN_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, \
                            __LINE__);                                                            \
    } while (0)

#define GN_REQUIRE(cond, ...)                                   \
    do {                                                        \
        if (!(cond)) return gn::fail(GN_ERR_INVALID_ARG, __VA_ARGS__); \
    } while (0)

#define GN_LAUNCH_CHECK() GN_HIP(hipGetLastError())

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }


// Compute units of the current device (persistent kernels launch one workgroup each); 256 if the query fails.
inline int compute_units() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        hipDeviceProp_t prop;
        cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return cached[dev];
}

// GN_DISABLE_FAST=1 forces the general kernels (used by the parity tests to cover both paths).
inline bool fast_paths_disabled() {
    const char* e = getenv("GN_DISABLE_FAST");
    return e && e[0] == '1';
}

// Device buffer owned by a plan.
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    hipError_t alloc(size_t count) {
        n = count;
        if (count == 0) {
            p = nullptr;
            return hipSuccess;
        }
        return hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

// Scoped device scratch of a plan builder, carved out of a few large blocks: a builder asks for ten to twenty arrays, and a
// hipMalloc / hipFree pair per array (the free waits for the device and unmaps) was a third of a relational plan's build
// time.  Every array is 256-byte aligned; a request that does not fit the current block opens a new one.
struct Scratch {
    static constexpr size_t kBlockBytes = (size_t)2 << 20;
    std::vector<void*> blocks;
    char* at = nullptr;
    size_t left = 0;
    Scratch() = default;
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    ~Scratch() {
        GN_LAP(nullptr);
        for (void* b : blocks) (void)hipFree(b);
        GN_LAP("  scratch: frees");
    }
    // `reserve_bytes`: what the builder knows it will ask for in total (one block then serves all of it)
    hipError_t reserve(size_t reserve_bytes) { return reserve_bytes > left ? open(reserve_bytes) : hipSuccess; }
    template <typename T>
    hipError_t get(T** out, size_t count) {
        const size_t bytes = (((count ? count : 1) * sizeof(T)) + 255) & ~(size_t)255;
        *out = nullptr;
        if (bytes > left) {
            const hipError_t e = open(std::max(bytes, kBlockBytes << std::min<size_t>(blocks.size(), 5)));
            if (e != hipSuccess) return e;
        }
        *out = reinterpret_cast<T*>(at);
        at += bytes; left -= bytes;
        return hipSuccess;
    }
  private:
    hipError_t open(size_t bytes) {
        void* b = nullptr;
        const hipError_t e = hipMalloc(&b, bytes);
        if (e != hipSuccess) return e;
        blocks.push_back(b);
        at = static_cast<char*>(b); left = bytes;
        return hipSuccess;
    }
};

// Grid size for a memory-bound grid-stride kernel: enough blocks to fill 256 CUs, capped.
inline int stream_grid(int64_t work_items, int block, int max_blocks = 256 * 8) {
    int64_t g = ceil_div(work_items, block);
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return static_cast<int>(g);
}

// Opt a kernel in to more than 64 KB of dynamic LDS, once per (kernel, device).
gn_status allow_large_lds(const void* kernel, int bytes);

// HIP events that the calling thread asked the NEXT timed kernel launch to carry (gn_time_next_launch): the launch site
// that supports it hands them to hipExtLaunchKernelGGL, which stamps the dispatch itself - no marker packets in the stream
// (an event record in front of and behind a launch costs ~4.5 us of stream time each on this stack).
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; };
LaunchEvents take_launch_events();          // the pending pair (or nulls), cleared

}  // namespace gn

// ---- plan layouts (shared between the builders and the kernels' launchers) --------------
struct gn_graph_plan {
    int64_t input_edges = 0;  // E the plan was built from (the reference's cache key)
    int64_t nnz = 0;          // stored coefficients (E' for GCN, E for bipartite)
    int64_t rows = 0;         // destination rows of the CSR
    int64_t table_rows = 0;   // rows of the gathered table (N for GCN, num_sources for bipartite)
    int64_t max_row_nnz = 0;
    int is_gcn = 0;
    // destination-major CSR streamed by the aggregation kernel
    gn::DevBuf<int32_t> rowptr;   // [rows + 1]
    gn::DevBuf<int32_t> col;      // [nnz]
    gn::DevBuf<float> coef;       // [nnz]
    // reference-order copies kept for myGCN.norm parity (GCN plans only)
    gn::DevBuf<int64_t> ref_edge_index;  // [2, nnz]
    gn::DevBuf<float> ref_norm;          // [nnz]
    // padded rows (bipartite plans whose longest row has at most 64 entries): lane L of the wave that owns row r reads entry
    // r * 64 + L directly - one dependent round trip less than row pointers -> columns (the external layer is three of them)
    int ell_ok = 0;
    gn::DevBuf<uint32_t> ell_col;        // [rows * 64] source row, 0xffffffff beyond the row's end
    gn::DevBuf<float> ell_coef;          // [rows * 64]
    // source-major CSR of the same coefficients (built on demand for the backward pass)
    int has_transpose = 0;
    gn::DevBuf<int32_t> t_rowptr;        // [table_rows + 1]
    gn::DevBuf<int32_t> t_col;           // [nnz] destination row of every stored coefficient
    gn::DevBuf<float> t_coef;            // [nnz]
    // LDS-staged encoding (gcn_blocked.hip; GCN plans whose stored weights are all 1, built on demand by
    // gn_graph_plan_build_blocked): destination rows in ranges x 16-row tiles, 16-bit source ids ordered for the LDS
    gn::DevBuf<float> dis;               // [rows] deg^-1/2 of every node (GCN plans)
    int unit_weights = 0;                // every stored weight (self loops included) is exactly 1
    int plain_ones = 0;                  // a plain sum without weights: every coefficient is exactly 1
    int blk_ok = 0, blk_cols = 0;        // built for layers of up to blk_cols output features
    int blk_cw = 0;                      // columns per column group (2, or 1 for larger graphs)
    int blk_rows = 0;                    // rows of one column group of the table (nodes + the zero row, padded to 1 KB pieces)
    int blk_cells = 0;                   // ranges of destination rows (workgroups per column group)
    int64_t blk_iters = 0;               // 512-byte iterations of the id stream
    gn::DevBuf<float> blk_dis;           // [blk_rows + 16] dis, zero padded
    gn::DevBuf<int32_t> blk_tile_off;    // [tiles + 4] first iteration of every tile
    gn::DevBuf<int32_t> blk_tile_rows;   // [tiles + 4][16] destination row of every quad of a tile (-1: none)
    gn::DevBuf<float> blk_tile_dis;      // [tiles + 4][16] dis of that row
    gn::DevBuf<uint32_t> blk_ids;        // per iteration 64 lanes x 4 uint16 source ids (512 bytes)
    gn::DevBuf<int32_t> blk_cell;        // [blk_cells][waves][12] tile and iteration range of every wave of a range, the ends of its first five tiles
    gn::DevBuf<float> blk_table;         // [blk_cols / blk_cw][blk_rows][blk_cw] dis * (x W) (scratch of the plan)
};

struct gn_rgcn_plan {
    int64_t input_edges = 0;   // full E
    int64_t edge_lo = 0, edge_hi = 0;
    int64_t shard_edges = 0;
    int64_t num_nodes = 0;
    int64_t num_relations = 0;
    int64_t max_row_nnz = 0;
    gn::DevBuf<float> indeg;       // [N] in-degree over the FULL graph, as float (divisor of the mean)
    // general path: destination-major CSR over the shard, column = relation * N + src
    gn::DevBuf<int32_t> rowptr;    // [N + 1]
    gn::DevBuf<uint32_t> key;      // [shard_edges]
    gn::DevBuf<uint32_t> skey;     // [shard_edges] the same rows' edges as src * R + relation, sorted by (destination, source, relation)
    gn::DevBuf<int32_t> row_order; // [N] destination rows by the shard's in-degree, largest first (rgcn_basis.hip deals rows in this order)
    int64_t heavy_rows = 0;        // rows of more than gn_layout::kBasisHeavyEdges edges (the first entries of row_order)
    // relation-major work items of the general weight gradient (rgcn_basis.hip): (relation, first edge, end edge, part | parts << 16)
    // over the shard's edges in the caller's (type-sorted) order, <= gn_layout::kRelDwItemEdges edges each
    gn::DevBuf<int32_t> dw_items;  // [n_dw_items][4]
    int64_t n_dw_items = 0, n_dw_parts = 0;   // parts: items of relations cut into more than one (their sums meet in a workspace)
    gn::DevBuf<int32_t> dw_multi;  // [n_dw_multi][4] relations of several parts: (relation, first slot, parts, 0)
    int64_t n_dw_multi = 0;
    // LDS-resident path (rgcn_fast.hip): work items = (relation, source tile, <= chunk edges)
    gn::DevBuf<int32_t> seg_rel;    // [n_items] relation of each work item
    gn::DevBuf<int32_t> item_tile;  // [n_items] source tile of each work item
    gn::DevBuf<int32_t> seg_begin;  // [n_items * buckets + 1] bucket offsets into packed
    gn::DevBuf<uint32_t> packed;    // [shard_edges] (dst << 16 | src - tile base), bucket-sorted inside an item
    gn::DevBuf<int32_t> wg_begin;   // [groups + 1] ranges into wg_items
    gn::DevBuf<int32_t> wg_items;   // item ids per persistent workgroup
    int64_t n_seg = 0;
    int fast_groups = 0, fast_ts = 0, fast_ts_pad = 0;
    size_t fast_lds_bytes = 0;
    int fast_cols = 32;             // output columns per workgroup of the LDS-resident kernel (32, or 16 = column halves)
    int fast_ok = 0;
    // destination-major path (rgcn_pair.hip): a workgroup owns up to three destination rows outright; per wave one
    // flat stream of 64-byte blocks (4 lane groups x 4 edges, each a 32-bit LDS byte offset of a relation's att row)
    gn::DevBuf<uint32_t> pair_stream;     // blocks of 16 words
    gn::DevBuf<uint32_t> pair_wave_first; // [groups * 8] first block of every wave
    gn::DevBuf<uint32_t> pair_desc;       // 32 dwords per unit: eight 8-bit block counts, [8..23] the chunk's 32 source ids (16 bits; num_nodes: none); pages of two units per wave
    gn::DevBuf<uint32_t> pair_wave_units; // [groups * 8] units of every wave
    gn::DevBuf<uint32_t> pair_wave_desc;  // [groups * 16] first descriptor of every wave
    gn::DevBuf<int32_t> pair_wg_dst;      // [groups][4] destination rows of a workgroup (-1: none)
    int pair_groups = 0, pair_d = 0, pair_chunks = 0;
    int64_t pair_unit_slots = 0;          // unit descriptors (the pair-sum buffer of the split launches has one 4 KB slot each at 32 bases)
    int64_t pair_blocks = 0;
    int pair_ok = 0;
};
