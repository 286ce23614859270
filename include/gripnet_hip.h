/*
 * gripnet_hip.h - C ABI of the MI355X (gfx950) supergraph propagation engine.
 *
 * The reference (NYXFLOWER/GripNet) has no FFI layer: its hot path is Python that calls into
 * ATen / torch_geometric / torch_scatter.  This header is the boundary a maintainer binds
 * instead of those calls; every entry point names the reference lines it replaces
 * (paths relative to the reference tree).  INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host;
 *   - feature matrices are row-major fp32 with an explicit leading dimension (elements);
 *   - node / edge indices are int64, exactly as the reference's torch.long tensors;
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous with respect to
 *     the host except the *_plan_create functions, which synchronise `stream` once;
 *   - outputs and scratch are caller-allocated; the library owns only what hangs off an
 *     opaque plan handle (the cached CSR / normalisation / counts of one static graph);
 *   - every function returns a gn_status; gn_last_error() gives the calling thread's message;
 *   - re-entrant and thread-safe on distinct plans; no global mutable state.
 */
#ifndef GRIPNET_HIP_H
#define GRIPNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GN_VERSION 154 /* 0.1.48 */

#if defined(GN_BUILDING)
#define GN_API __attribute__((visibility("default")))
#else
#define GN_API
#endif

typedef enum gn_status {
    GN_OK = 0,
    GN_ERR_INVALID_ARG = 1,   /* null pointer, negative size, unsupported shape          */
    GN_ERR_HIP = 2,           /* a HIP runtime call failed (message carries hipGetErrorString) */
    GN_ERR_INDEX_RANGE = 3,   /* an edge endpoint / relation id is outside its table     */
    GN_ERR_UNSUPPORTED = 4,   /* sizes beyond what the 32-bit plan encoding can hold     */
    GN_ERR_EDGE_COUNT = 5     /* plan was built for a different number of edges          */
} gn_status;

/* Optional row-wise side copy fused into an aggregation launch: the concat slots the reference fills
 * with torch.cat (gripnet/layers.py:264-266,309,376).  dst[i, 0:cols] = src[i, 0:cols] (mode 0) or
 * |src[i, 0:cols]| (mode 1) for i in [0, rows); rows must not exceed the rows of the launch. */
typedef struct gn_side_copy {
    const float* src; int64_t ld_src;
    float* dst; int64_t ld_dst;
    int64_t rows, cols;
    int mode;
} gn_side_copy;

/* Optional third output of an aggregation launch, and optional input of gn_rgcn_forward_f32: a row-major matrix
 * X [rows, 16 nt] as the three bf16 terms the relational layer's matrix products run on (x = hi + mid + lo exactly,
 * each cut by truncation).  The relational kernel otherwise splits the same 645 x 48 values again in every one of its
 * 13,500 units (gripnet/layers.py:172-186 is the product this feeds); the layer that PRODUCES x - the external layer
 * writing y, its side copy writing |target_feat| (layers.py:370,376) - can leave the terms behind for free.
 * Layout: planes[(row * 16 + cell) * cb + (term * nt + j) * 2], two bytes each: cell = column / nt, j = column % nt,
 * cb = 4 * ceil(3 nt / 2) bytes per cell, term 0 / 1 / 2 = hi / mid / lo.  The buffer holds rows + 1 rows and the
 * caller keeps row `rows` ZERO (a source that is none names it): (rows + 1) * 16 * cb bytes (gn_split_planes_bytes). */
typedef struct gn_split_planes {
    void* planes;
    int64_t rows;
    int nt;           /* columns of X / 16 */
    int col_main;     /* column of X where the launch's main output starts */
    int col_side;     /* column of X where the launch's side copy starts (ignored without one) */
} gn_split_planes;

typedef struct gn_graph_plan gn_graph_plan; /* GCN-style graphs (square or bipartite)  */
typedef struct gn_rgcn_plan gn_rgcn_plan;   /* multi-relational graph of one supervertex */

GN_API int gn_version(void);
/* Message of the last failing call made by this thread ("" if none). */
GN_API const char* gn_last_error(void);

/* ---------------------------------------------------------------------------------------
 * GCN-style normalised graph.  Replaces myGCN.norm + its cache (gripnet/layers.py:52-69,
 * 83-90): drops existing self loops, appends one loop per node (weight = that node's last
 * listed loop weight, else 1, or 2 if `improved`), deg = sum of weights by DESTINATION,
 * norm = deg[src]^-1/2 * w * deg[dst]^-1/2 (inf -> 0).  The plan holds edge_index'
 * (non-loops in input order, then loops 0..N-1) and norm in the reference's order, plus a
 * destination-major CSR (int32 columns, fp32 coefficients) used by gn_graph_aggregate_f32.
 * w may be NULL (all ones).
 */
GN_API gn_status gn_gcn_plan_create(const int64_t* src, const int64_t* dst, const float* w, int64_t num_edges,
                             int64_t num_nodes, int improved, void* stream, gn_graph_plan** plan);

/* External (inter-supervertex) layer graph.  Replaces the index shift, zero padding and the
 * dead self loops of interGraph.forward (gripnet/layers.py:363-368) by their closed form:
 * coefficient of edge s->t is w_e * (1 + sum of weights into t)^-1/2; rows are targets. */
GN_API gn_status gn_bipartite_plan_create(const int64_t* src, const int64_t* dst, const float* w, int64_t num_edges,
                                   int64_t num_sources, int64_t num_targets, void* stream, gn_graph_plan** plan);

/* Plain (un-normalised) weighted-sum graph: out[t] = sum_{e: dst(e)=t} w_e * table[src(e)], w NULL = ones.
 * Used by the backward pass of the relational layer (rows = (relation, source) pairs, autograd of
 * gripnet/layers.py:178-189). */
GN_API gn_status gn_sum_plan_create(const int64_t* src, const int64_t* dst, const float* w, int64_t num_edges,
                             int64_t num_sources, int64_t num_targets, void* stream, gn_graph_plan** plan);

GN_API void gn_graph_plan_destroy(gn_graph_plan* plan);
/* Number of edges the plan was built from (the reference's cache key, layers.py:76-84). */
GN_API int64_t gn_graph_plan_input_edges(const gn_graph_plan* plan);
/* Number of stored coefficients: E' = non-loop edges + N for a GCN plan, E for a bipartite plan. */
GN_API int64_t gn_graph_plan_nnz(const gn_graph_plan* plan);
/* Copy edge_index' ([2, E'] int64) and norm ([E'] fp32) in the reference's order (GCN plans). */
GN_API gn_status gn_graph_plan_export(const gn_graph_plan* plan, int64_t* edge_index_out, float* norm_out, void* stream);

/* out[i, :] = act( sum_{e: dst(e)=i} coef_e * xw[src(e), :] + bias ),  i in [0, rows).
 * Replaces propagate/message/update of myGCN (gripnet/layers.py:92-100) and the ReLU that
 * always follows (layers.py:279,305,370).  xw = x @ W is computed first with gn_gemm_f32,
 * as the reference does (layers.py:73).  bias may be NULL.  `out` may be a column slice of a
 * wider matrix (ld_out), which is how the concat of layers.py:309,376 is written in place. */
GN_API gn_status gn_graph_aggregate_f32(const gn_graph_plan* plan, const float* xw, int64_t ld_xw, int64_t num_features,
                                 const float* weight /* nullable */, int64_t out_features,
                                 const float* bias, int relu, float* out, int64_t ld_out,
                                 const gn_side_copy* side /* nullable */, const gn_split_planes* planes /* nullable */,
                                 void* stream);
/* `planes`: the launch also leaves the bf16 split planes of what it writes (main output and side copy, as columns
 * col_main.. / col_side.. of X).  The wave-per-row transform kernel writes them from its epilogue; the other kernels
 * are followed by gn_split_planes_f32 launches.  The same planes from a matrix that already exists: */
GN_API size_t gn_split_planes_bytes(int64_t rows, int nt);
GN_API gn_status gn_split_planes_f32(const float* src, int64_t ld_src, int64_t rows, int64_t cols, int64_t col0,
                              const gn_split_planes* planes, void* stream);
/* With weight != NULL ([num_features, out_features], row-major, contiguous) the table is the layer INPUT x
 * and the call computes act( (A_norm x) W + bias ) = act( A_norm (x W) + bias ): the contraction of
 * layers.py:73 runs on the aggregated row, so no x W launch is needed.  Supported for (num_features,
 * out_features) in {16,32,64} x {16} and {32,64} x {32}, and for the wide layers gn_graph_transform_fusable describes
 * (GN_ERR_UNSUPPORTED otherwise: call gn_gemm_f32 first).  gn_transform_fusable / gn_graph_transform_fusable tell
 * without launching. */
GN_API int gn_transform_fusable(int64_t in_features, int64_t out_features);
/* The same question for one plan: besides the widths above, layers at least as wide as their input (64 or 128 input
 * features, out_features a multiple of 16 in [in_features, 128]) contract on the matrix cores in the same launch when
 * the plan's rows average fewer than 48 neighbours and there are at least 4,096 of them (the second layer of the
 * node-classification stacks: no x W launch, no [N, out] round trip). */
GN_API int gn_graph_transform_fusable(const gn_graph_plan* plan, int64_t in_features, int64_t out_features);

/* Column-group encoding of a GCN plan for LDS-staged gathers (the gene supervertex of PoSE: 19,081 nodes, 1.45 M
 * stored edges; gcn_blocked.hip).  Applies to plans whose stored weights are all exactly 1, self loops included
 * (edge_weight NULL or ones and not `improved`: every GripNet caller, GripNet-pose.py:52,117-120): norm_e =
 * dis[src] dis[dst] then factorises and an edge is a 16-bit source id.  The table T = dis * (x W) is cut by COLUMNS:
 * a workgroup holds two fp32 columns (one above ~19,900 nodes, up to ~39,800) of EVERY node in 160 KB of LDS, so it
 * owns its destination rows outright - no partial sums, no combine pass.  gn_graph_aggregate_f32 takes this path (two
 * launches: the exact-fp32 transform writing T column-group-major into scratch the PLAN owns, then the gather from
 * LDS; fixed summation order) when the plan has the encoding and the shapes are covered: out_features 16 or 32 <= cols,
 * with `weight` num_features in {16,32,64}, x 16-byte aligned with ld % 4 == 0.  Calls on one plan must be
 * stream-ordered (the scratch table).  Graphs that do not qualify (weighted, `improved`, fewer than 4,096 nodes or
 * fewer than 16 stored edges per node, too many nodes for one column in LDS) keep the wave-per-row kernels and the call
 * returns GN_OK; gn_graph_plan_blocked_cols tells (0 = not built).  Copies the CSR to the host once, schedules it there
 * and synchronises `stream`: meant for graphs that are kept (the Python layer builds it only for `cached=True`). */
GN_API gn_status gn_graph_plan_build_blocked(gn_graph_plan* plan, int64_t cols, void* stream);
GN_API int64_t gn_graph_plan_blocked_cols(const gn_graph_plan* plan);

/* bf16 STORAGE of the gathered table (the reference is fp32 throughout, this is the build's own
 * reduced-traffic variant for the node-classification suite): gn_cast_bf16 rounds x W to bf16 once,
 * gn_graph_aggregate_bf16 computes out[i,:] = act( sum_e coef_e * table[src(e),:] + bias ) from it with fp32 sums and an
 * fp32 result.  num_features % 8 == 0, 16-byte aligned rows.  Error against the fp32 layer: one bf16 rounding (2^-9
 * relative) of every gathered element; exact against the same sum over the rounded table. */
GN_API gn_status gn_cast_bf16(const float* src, int64_t ld_src, uint16_t* dst, int64_t ld_dst, int64_t rows, int64_t cols,
                       void* stream);
GN_API gn_status gn_graph_aggregate_bf16(const gn_graph_plan* plan, const uint16_t* table, int64_t ld_table, int64_t num_features,
                                  const float* bias, int relu, float* out, int64_t ld_out,
                                  const gn_side_copy* side /* nullable */, void* stream);

/* Backward pass of the GCN-style layers (autograd of GripNet-pose.py:140-146 through layers.py:92-100).
 * gn_graph_plan_build_transpose adds the source-major CSR of the same coefficients to a plan (once;
 * synchronises `stream`); gn_graph_aggregate_t_f32 then computes, for every source row s,
 *   gxw[s, :] = sum_{e: src(e)=s} coef_e * g[dst(e), :]        (d loss / d (x W) from d loss / d out). */
GN_API gn_status gn_graph_plan_build_transpose(gn_graph_plan* plan, void* stream);
GN_API gn_status gn_graph_aggregate_t_f32(const gn_graph_plan* plan, const float* g, int64_t ld_g, int64_t num_features,
                                   float* out, int64_t ld_out, void* stream);

/* ---------------------------------------------------------------------------------------
 * Dense fp32 contraction on the matrix cores:
 *   C[b] = act( gather_rows(A[b]) @ B[b] + bias ),  b in [0, batch).
 * Replaces torch.matmul at gripnet/layers.py:73,172-173,193,383 and decoder.py:42.
 * a_rows (int64, may be NULL) selects rows of A (z[node_list], decoder.py:42).
 * stride_* are batch strides in elements (0 = shared operand).
 * Arithmetic (part of the contract, never read from the environment): by default fp32-faithful - the fp32 matrix
 * instruction (v_mfma_f32_16x16x4_f32, exact) or, for tall-skinny products, bf16 matrix instructions on operands split
 * into THREE bf16 terms (six products; what is dropped is below 2^-23 of a product).  GN_GEMM_ARITH_FAST keeps two terms
 * and three products (<= 2^-16 relative per product). */
#define GN_GEMM_RELU 1           /* flags of gn_gemm_f32 */
#define GN_GEMM_ARITH_FAST 2
#define GN_GEMM_B_TRANSPOSED 4  /* b is given as [n, k] row-major (leading dimension ldb >= k): c = a b^T - the dx = g W^T of the backward passes */
#define GN_GEMM_ACCUMULATE 8    /* c += a b (+ bias) instead of c = ... */
#define GN_GEMM_A_TRANSPOSED 16  /* a is given as [k, m] row-major (lda >= m): c = a^T b; m <= 64 or n <= 32 only: dbasis = att^T dW */
#define GN_GEMM_JOIN_BATCH 32   /* the product may leave with the open batch (gn_dense_batch_begin / _end, below) */
#define GN_GEMM_SPLIT_KERNEL 128 /* take the tall-skinny split kernel whatever the row count (k % 32 == 0, aligned rows): a caller that cuts one
                                 * product into slabs of rows gets the same bits for every slab size */
#define GN_GEMM_OUT_BF16 64     /* c is a bf16 table [m, n] (ldc in bf16 elements, rows 8-byte aligned, n % 4 == 0): every value is rounded to
                                 * nearest even ONCE, where the product is stored - gn_cast_bf16's rounding without its pass (the table of a
                                 * bf16-storage layer, gn_graph_aggregate_bf16).  Tall-skinny products only (m >= 2048, k % 32 == 0, no
                                 * accumulate / addend / row gather): GN_ERR_UNSUPPORTED otherwise, and the caller casts a fp32 product */
GN_API gn_status gn_gemm_f32(const float* a, int64_t lda, int64_t stride_a, const int64_t* a_rows, int64_t a_table_rows,
                      const float* b, int64_t ldb, int64_t stride_b, float* c, int64_t ldc, int64_t stride_c,
                      int64_t m, int64_t n, int64_t k, int64_t batch, const float* bias, int flags, void* stream);
/* The same product with an ADDEND: c = a b (+ bias) (+ c with GN_GEMM_ACCUMULATE) + addend, addend [m, n] with rows ld_addend
 * floats apart (may be NULL: gn_gemm_f32), batch == 1.  For the backward pass of a layer whose input also sits in a concat
 * (layers.py:280-281,307-309): dx = g W^T + (the concat's gradient columns of that input) leaves one launch instead of the
 * product and an element-wise sum.  The addend is only read; it may be a column slice of a wider matrix. */
GN_API gn_status gn_gemm_addend_f32(const float* a, int64_t lda, int64_t stride_a, const int64_t* a_rows, int64_t a_table_rows,
                      const float* b, int64_t ldb, int64_t stride_b, float* c, int64_t ldc, int64_t stride_c,
                      int64_t m, int64_t n, int64_t k, int64_t batch, const float* bias, const float* addend, int64_t ld_addend,
                      int flags, void* stream);

/* out[k1, k2] = x^T g over m rows (x [m, k1], g [m, k2]; k1 * k2 <= 4096): the weight gradients dW = x^T (A_norm^T g)
 * of the GCN-style layers and d root = x^T g of the relational one (autograd of layers.py:73,193).  Row slices are
 * reduced in a fixed order: bitwise reproducible.  Scratch is caller-provided (gn_xtg_workspace_bytes).
 * flags: GN_XTG_TICKET_ZEROED promises that the 64 bytes at offset gn_xtg_workspace_bytes(k1, k2) - 64 of the workspace
 * were zeroed once by the caller and are touched by nobody else (calls on one workspace stream-ordered): up to 64 x 32 outputs then take ONE launch on the matrix
 * cores (slices added in slice order by the last slice to arrive - beyond 32 slices, the products over tens of thousands of
 * rows, in two levels: sets of sixteen slices, then the sets) instead of a partial-sum launch and a fold. */
#define GN_XTG_TICKET_ZEROED 1
#define GN_XTG_JOIN_BATCH 2      /* the product may leave with the open batch (gn_dense_batch_begin / _end) */
GN_API size_t gn_xtg_workspace_bytes(int64_t k1, int64_t k2);
/* 1 when gn_xtg_f32 takes a product of these sizes as ONE wide product (k1 in {64, 128, 256}, k2 in {32, 64, 128}, 2-16 tiles of
 * 64 x 32, at least 4096 rows: a workgroup per row slice computes every tile, a fold launch adds the slices in slice order; x and g
 * are read once) - the weight gradients of the 128 x 64 ... 256 x 128 layers of the node-classification models.  Such a product may
 * have more than 4096 outputs; it is never queued in a dense batch. */
GN_API int gn_xtg_wide_supported(int64_t m, int64_t k1, int64_t k2);
GN_API gn_status gn_xtg_f32(const float* x, int64_t ld_x, const float* g, int64_t ld_g, int64_t m, int64_t k1, int64_t k2,
                     float* out, int64_t ld_out, void* workspace, size_t workspace_bytes, int flags, void* stream);

/* Element-wise merges of the external layer (gripnet/layers.py:375-384) and slot copies of the
 * concat outputs:  mode 0: dst = src;  1: dst = |src|;  2: dst = (dst + |src|) / 2;
 * 3: dst = (dst + relu(src)) / 2;  4: dst = (dst + src + src2) / 3 (freebase-c merge,
 * GripNet-freebase-c.py:158-162).  Backward helpers:  5: dst = src2 > 0 ? src : 0 (ReLU mask by the
 * saved output);  6: dst = src * sign(src2) (gradient of |.|);  7: dst = src / 3 (every operand's share of mode 4);
 * 8: dst = src / 2 (the first operand's share of modes 2 and 3);  9: dst = src * sign(src2) / 2 (the |.| operand of
 * mode 2);  10: dst = src2 > 0 ? src / 2 : 0 (the ReLU operand of mode 3). */
GN_API gn_status gn_merge_f32(float* dst, int64_t ld_dst, const float* src, int64_t ld_src, const float* src2,
                       int64_t ld_src2, int64_t rows, int64_t cols, int mode, void* stream);

/* Row softmax in place (decoder.py:43). */
GN_API gn_status gn_softmax_rows_f32(float* x, int64_t ld, int64_t rows, int64_t cols, void* stream);
/* Its gradient (autograd of decoder.py:43 in the training loops, GripNet-aminer.py:124-147): dx = probs * (grad - sum_c grad_c probs_c) per row. */
GN_API gn_status gn_softmax_rows_backward_f32(const float* probs, int64_t ld_probs, const float* grad, int64_t ld_grad, float* dx,
                                       int64_t ld_dx, int64_t rows, int64_t cols, void* stream);
/* out[i, :] = softmax?( z[node_list[i], :] @ W ), i in [0, m): multiClassInnerProductDecoder.forward in one launch
 * (gripnet/decoder.py:42-43: `pred = torch.matmul(z[node_list], self.weight)`, then `torch.softmax(pred, dim=1)`).
 * node_list may be NULL (rows 0..m-1); an entry outside [0, table_rows) scores a row of zeros, as gn_gemm_f32's a_rows.
 * W is [k, n] row-major.  Up to 16 classes run as one pass over the selected rows; more go through gn_gemm_f32 +
 * gn_softmax_rows_f32. */
GN_API gn_status gn_class_scores_f32(const float* z, int64_t ld_z, int64_t table_rows, const int64_t* node_list, int64_t m,
                              const float* w, int64_t ld_w, int64_t k, int64_t n, int softmax, float* out, int64_t ld_out,
                              void* stream);

/* ---------------------------------------------------------------------------------------
 * Multi-relational internal layer (myRGCN, gripnet/layers.py:165-197).
 * The plan caches what depends only on the (static) dd graph: per-destination in-degree over
 * ALL relations (the global mean of layers.py:131) and the relation-major / destination-major
 * edge encodings the kernels stream.  range_list ([R,2] int64, half-open, must tile [0,E) in
 * order as produced by utils.get_range_list) may live on the host or on the device.
 * edge_lo/edge_hi select a contiguous edge sub-range (multi-GPU sharding by relation / edge
 * range); in-degrees are always over the full [0,E) graph.
 */
GN_API gn_status gn_rgcn_plan_create(const int64_t* src, const int64_t* dst, const int64_t* range_list,
                              int range_list_on_host, int64_t num_relations, int64_t num_edges,
                              int64_t num_nodes, int64_t edge_lo, int64_t edge_hi, void* stream,
                              gn_rgcn_plan** plan);
/* flags & GN_RGCN_PLAN_LIGHT: only what the general O(E) kernel reads (the destination-major key list, sorted on the device,
 * in-degrees, the rows' degree order) - none of the host-built schedules of the LDS-resident kernels: for a caller whose edge
 * list changes from call to call (the reference's myRGCN has no set-up cost at all, layers.py:165-169), a fraction of the
 * full plan's build time; its forwards take GN_RGCN_PATH_GENERAL (or the table path beyond its shapes). */
#define GN_RGCN_PLAN_LIGHT 1
GN_API gn_status gn_rgcn_plan_create_ex(const int64_t* src, const int64_t* dst, const int64_t* range_list,
                                 int range_list_on_host, int64_t num_relations, int64_t num_edges,
                                 int64_t num_nodes, int64_t edge_lo, int64_t edge_hi, int flags, void* stream,
                                 gn_rgcn_plan** plan);
GN_API void gn_rgcn_plan_destroy(gn_rgcn_plan* plan);
GN_API int64_t gn_rgcn_plan_input_edges(const gn_rgcn_plan* plan);

/* Timing without marker packets: the NEXT kernel launched by this thread through gn_rgcn_forward_f32 (destination-major
 * kernel) or gn_distmult_plan_forward_f32 (row-class kernel) carries the two HIP events (hipEvent_t, created by the caller)
 * as the start / stop stamps of its own dispatch (hipExtLaunchKernel); hipEventElapsedTime(start, stop) is then the
 * kernel's duration on its stream.  A pair of hipEventRecord calls around a launch costs ~9 us of stream time on this
 * stack - a tenth of the PoSE forward.  gn_time_launch_pending: 1 when the events are still waiting (the call in
 * between took a kernel that does not carry them: time it with event records instead); clears them. */
GN_API gn_status gn_time_next_launch(void* start_event, void* stop_event);

/* Stream ordering as an entry point (so that a recorded call sequence can carry it): what is given to `later_stream` after
 * this call starts only when everything given to `earlier_stream` before this call has finished (one hipEventRecord +
 * hipStreamWaitEvent on an event the library owns; no host synchronisation). */
GN_API gn_status gn_stream_order(void* earlier_stream, void* later_stream);
GN_API int gn_time_launch_pending(void);

/* The plan builders (`*_plan_create`, gn_graph_plan_build_blocked) keep two things of the PROCESS between builds: their host
 * threads (GN_PLAN_THREADS of them, parked; a forked child starts its own at its first build) and ONE block of host memory
 * their large arrays come out of (it grows to 5/4 of what the largest build asked for, at most 1 GB; ~110 MB after the
 * decoder plan of PoSE-0, ~500 MB after that of a four times larger list).  gn_host_scratch_release gives the block back
 * to the system (unless a build is using it right now) and returns the bytes freed; the next build starts a new one. */
GN_API size_t gn_host_scratch_release(void);

/* Bytes of caller-provided scratch a gn_rgcn_forward_f32 call with these shapes and flags needs: none on the
 * destination-major kernel, W_r and the slabs of the LDS-accumulator kernel, a slab of rows (<= 256 MB) and the stacked
 * weights on the general path (independent of R and N), the [R, N, out] table on the table path.  (The forward checks the workspace against the kernel IT takes and refuses a smaller one.) */
GN_API size_t gn_rgcn_workspace_bytes(const gn_rgcn_plan* plan, int64_t in_features, int64_t out_features,
                               int64_t num_bases, int flags);

#define GN_RGCN_PARTIAL 1        /* flags of gn_rgcn_forward_f32 */
#define GN_RGCN_ARITH_FAST 4     /* dense products on two-term bf16 splits (<= 2^-16 relative per product) instead of the
                                  * default fp32-faithful arithmetic (three-term splits / the fp32 matrix instruction) */
/* The layer as TWO launches (destination-major kernel, default arithmetic, x given as split planes; round 6).  The per-edge work
 * of the layer - the att rows of the edges of every (destination, source) pair summed: P_i[s,:] = sum_{e: s -> i} att[r(e),:] -
 * does not depend on x.  GN_RGCN_PAIR_SUMS_ONLY: only that half, written to `workspace` (gn_rgcn_workspace_bytes with this
 * flag; ~55 MB on PoSE-0); reads `att` and the plan only (x, basis, root, out may be NULL): it can run on another stream
 * while the layers that produce x still run.  GN_RGCN_PAIR_SUMS_READY: `workspace` holds the sums of THIS step (the caller
 * orders the two calls, e.g. gn_stream_order); the launch reads them back, contracts with x and finishes the layer - the
 * bits of the one-launch form.  The reference recomputes W_r from att every forward (layers.py:172-173): so must the sums
 * be - every step, never kept across steps.  GN_ERR_UNSUPPORTED where the destination-major kernel does not apply. */
#define GN_RGCN_PAIR_SUMS_ONLY 16
#define GN_RGCN_PAIR_SUMS_READY 32
/* GN_RGCN_BASIS_TRANSPOSED: `basis` is stored [bases][fout][fin] - the layer's own parameter seen from the reversed layer of
 * its backward (dx = the same layer on the reversed graph with W_r^T, autograd.rgcn_edge_gradients): no transposed copy per
 * step.  Read by the destination-major kernel only; GN_ERR_UNSUPPORTED on the others (pass a transposed copy there). */
#define GN_RGCN_BASIS_TRANSPOSED 64
/* Kernel choice, for tests and measurements (0 = the library decides; a kernel that does not cover the shapes is not
 * forced): (GN_RGCN_PATH_x << GN_RGCN_PATH_SHIFT) in flags.  gn_rgcn_forward_path tells which one a call would take. */
#define GN_RGCN_PATH_SHIFT 8
#define GN_RGCN_PATH_PAIR 1      /* destination-major in basis space (rgcn_pair.hip): no W_r, no workspace, one launch */
#define GN_RGCN_PATH_LDS 3       /* LDS accumulator rows (rgcn_fast.hip) */
#define GN_RGCN_PATH_GENERAL 4   /* any size, O(E) memory: basis-space sums per destination on the fp32 matrix instruction, one dense
                                 * product per slab of rows with [basis ; root] (rgcn_basis.hip); up to 64 bases and 128 input
                                 * features, else the table path */
#define GN_RGCN_PATH_TABLE 5     /* transform-then-gather through an [R, N, out] table in HBM (rgcn.hip): any shape */
GN_API int gn_rgcn_forward_path(const gn_rgcn_plan* plan, int64_t in_features, int64_t out_features, int64_t num_bases, int flags);

/* flags & GN_RGCN_PARTIAL == 0:  out[i,:] = act( (sum_{e: dst=i} x[src_e] W_{r(e)}) / max(1, indeg_i) + x[i] root + bias )
 * flags & GN_RGCN_PARTIAL:       out[i,:] = sum_{e in [edge_lo,edge_hi): dst=i} x[src_e] W_{r(e)}   (un-normalised
 *                                shard contribution; all-reduce it, then call gn_rgcn_finalize_f32)
 * with W_r = sum_b att[r,b] basis[b]  (layers.py:172-173).  bias may be NULL. */
GN_API gn_status gn_rgcn_forward_f32(const gn_rgcn_plan* plan, const float* x, int64_t ld_x, int64_t in_features,
                              const float* basis, const float* att, int64_t num_bases, const float* root,
                              const float* bias, int64_t out_features, int relu, int flags, float* out,
                              int64_t ld_out, const gn_side_copy* side /* nullable */,
                              const void* x_planes /* nullable: gn_split_planes.planes of x, written by its producer */,
                              void* workspace, size_t workspace_bytes, void* stream);

/* out[i,:] = act( summed[i,:] / max(1, indeg_i) + x[i] root + bias )  (layers.py:131,191-197). */
GN_API gn_status gn_rgcn_finalize_f32(const gn_rgcn_plan* plan, const float* summed, int64_t ld_summed, const float* x,
                               int64_t ld_x, int64_t in_features, const float* root, const float* bias,
                               int64_t out_features, int relu, float* out, int64_t ld_out,
                               const gn_side_copy* side /* nullable */, void* stream);

/* Weight gradient of the relational layer (autograd of layers.py:165-197 under the loss of GripNet-pose.py:140-146): with
 * P = sum_e x[src_e] W_{r(e)} and gm = dL/dP ([n, out_features]),
 *   dw[r] = sum_{e in r} x[src_e]^T gm[dst_e]      ([num_relations][in_features][out_features], row-major, overwritten),
 * from which dbasis = att^T dw and datt = dw basis^T (layers.py:172-173).  The plan is built from the plain-sum plan of
 * the layer's edges keyed dst -> relation * n + src (gn_sum_plan_create with num_relations * n targets over n
 * sources: one row per (relation, source), its entries the destinations) and holds what depends on the edges only.
 * One launch: the (relation, source) sums of gm never exist in memory (they were 80 MB per step on PoSE).  No float
 * atomics: the same bits every run.  GN_ERR_UNSUPPORTED (see gn_rel_weight_grad_supported) for in_features other than
 * 16 / 32 / 48 / 64, out_features other than 16 / 32, or a gm table that does not fit the LDS: sum the rows with
 * gn_graph_aggregate_f32 on the plain-sum plan and contract them with x per relation instead. */
/* The same gradient for supervertices beyond the fused kernel (any number of nodes; up to 128 input and 64 output features):
 * per relation the outer products of its edges' endpoint rows on the fp32 matrix instruction, O(E) memory - the reference's
 * own per-relation formulation (layers.py:178-186 under autograd; the all-nodes baseline rgcn_pose.py:53-106).  src / dst:
 * the rows of the caller's [2, E] int64 edge_index the plan was built from (type-sorted; a shard's plan covers its edge
 * range).  workspace: gn_rgcn_weight_grad_workspace_bytes (the parts of relations of more than 512 edges, added in part
 * order: the same bits every run). */
GN_API size_t gn_rgcn_weight_grad_workspace_bytes(const gn_rgcn_plan* plan, int64_t in_features, int64_t out_features);
GN_API int gn_rgcn_weight_grad_supported(const gn_rgcn_plan* plan, int64_t in_features, int64_t out_features);
GN_API gn_status gn_rgcn_weight_grad_f32(const gn_rgcn_plan* plan, const int64_t* src, const int64_t* dst, const float* x, int64_t ld_x,
                                  int64_t in_features, const float* gm, int64_t ld_g, int64_t out_features, float* dw, void* workspace,
                                  size_t workspace_bytes, void* stream);
typedef struct gn_rel_grad_plan gn_rel_grad_plan;
GN_API gn_status gn_rel_grad_plan_create(const gn_graph_plan* relation_source_sums, int64_t num_nodes, int64_t num_relations,
                                  void* stream, gn_rel_grad_plan** plan);
GN_API void gn_rel_grad_plan_destroy(gn_rel_grad_plan* plan);
GN_API int gn_rel_weight_grad_supported(const gn_rel_grad_plan* plan, int64_t in_features, int64_t out_features);
GN_API gn_status gn_rel_weight_grad_f32(const gn_rel_grad_plan* plan, const float* x, int64_t ld_x, int64_t in_features,
                                 const float* gm, int64_t ld_gm, int64_t out_features, float* dw, void* stream);

/* ---------------------------------------------------------------------------------------
 * DistMult decoder (multiRelaInnerProductDecoder.forward, gripnet/decoder.py:19-23):
 *   out[e] = sigma?( sum_k z[u_e,k] * z[v_e,k] * D[r_e,k] ).
 * Consumes the raw int64 edge_index rows and edge_type every call (negative samples change
 * every epoch, GripNet-pose.py:131), so nothing is cached.  An endpoint or relation outside its
 * table yields NaN for that edge and sets bit 0 of *error_flag (device int32, may be NULL). */
GN_API gn_status gn_distmult_forward_f32(const float* z, int64_t ld_z, int64_t num_nodes, int64_t num_features,
                                  const int64_t* u, const int64_t* v, const int64_t* edge_type,
                                  const float* d, int64_t ld_d, int64_t num_relations, int64_t num_edges,
                                  int apply_sigmoid, float* out, int32_t* error_flag, void* stream);

/* The same decoder (same kernel, same arithmetic, same bits) on PACKED triples: packed_uv[e] = u | v << 16 and
 * relation[e] (uint16): 6 bytes per edge and column phase instead of 24.  For edge lists that change every epoch, where
 * a plan cannot pay off: the negative samples of GripNet-pose.py:131,138, which gn_negative_sampler_sample_packed writes
 * in this form; `relation` is the static edge_type of the positives (GripNet-pose.py:138 scores the negatives with
 * train_et) narrowed once.  GN_ERR_UNSUPPORTED when ids need more than 16 bits or the node table does not fit the LDS
 * in four column phases. */
GN_API gn_status gn_distmult_packed_forward_f32(const float* z, int64_t ld_z, int64_t num_nodes, int64_t num_features,
                                         const uint32_t* packed_uv, const uint16_t* relation, const float* d, int64_t ld_d,
                                         int64_t num_relations, int64_t num_edges, int apply_sigmoid, float* out,
                                         int32_t* error_flag, void* stream);

/* The same decoder on a cached, re-encoded STATIC edge list: the positive edges, which the reference scores with the
 * same train_idx / train_et tensors every epoch (GripNet-pose.py:137,185).  The plan validates the triples once
 * (GN_ERR_INDEX_RANGE), pairs up triples with the same unordered node pair and relation (both directions of an edge,
 * utils.py:132-138: the same score bit for bit, computed once and written to both positions), packs the scored ones
 * into 32-bit words and orders them inside every 64-edge batch for conflict-free LDS gathers; scores come out in the
 * caller's edge order, bitwise equal to gn_distmult_forward_f32.
 * Plan creation copies the triples to the host and synchronises `stream`.  GN_ERR_UNSUPPORTED: more than 8192 nodes
 * or 65535 relations (create), or a node table that does not fit the LDS in four column phases (forward) - use
 * gn_distmult_forward_f32 then. */
typedef struct gn_distmult_plan gn_distmult_plan;
GN_API gn_status gn_distmult_plan_create(const int64_t* u, const int64_t* v, const int64_t* edge_type, int64_t num_edges,
                                  int64_t num_nodes, int64_t num_relations, int64_t num_features, void* stream,
                                  gn_distmult_plan** plan);
GN_API void gn_distmult_plan_destroy(gn_distmult_plan* plan);
GN_API int64_t gn_distmult_plan_edges(const gn_distmult_plan* plan);
GN_API gn_status gn_distmult_plan_forward_f32(const gn_distmult_plan* plan, const float* z, int64_t ld_z, int64_t num_features,
                                       const float* d, int64_t ld_d, int apply_sigmoid, float* out, void* stream);
/* The same scores in several launches: feature columns [col_lo, col_hi) only (col_lo a multiple of 4; z and d point at
 * column 0, ld_z >= col_hi).  The launch with col_lo = 0 starts the sums; a launch with col_hi < num_features leaves raw
 * partial sums in `out` (at the edges' own positions), the next one continues from them; the launch with
 * col_hi = num_features applies the sigmoid and fills the mirror positions.  Split at the column where the
 * single launch would change phase (48 of 80 at 645 nodes) the scores are the same bits.  For the relation-sharded
 * forward: the columns that are the relational layer's INPUT (concat slot 0 of homoGraph / interGraph, gripnet/layers.py:264-266)
 * are scored while the all-reduce of the layer's partial sums is in flight. */
GN_API gn_status gn_distmult_plan_forward_cols_f32(const gn_distmult_plan* plan, const float* z, int64_t ld_z, int64_t num_features,
                                            int64_t col_lo, int64_t col_hi, const float* d, int64_t ld_d, int apply_sigmoid,
                                            float* out, void* stream);

/* Backward of the DistMult decoder (autograd of decoder.py:19-23 under the loss of GripNet-pose.py:140-146).
 * grad_logit[e] = d loss / d s_e (the caller folds the sigmoid derivative in).  dz [n, F] and dD [R, F]
 * are overwritten:  dz[u_e] += g_e z[v_e] * D[r_e],  dz[v_e] += g_e z[u_e] * D[r_e],  dD[r_e] += g_e z[u_e] * z[v_e].
 * Edges with an id outside its table contribute nothing.  Sort-based segmented reduction: no atomics,
 * bitwise reproducible.  Scratch is caller-provided (gn_distmult_backward_workspace_bytes).
 * gn_distmult_backward_ex_f32 takes flags: GN_DM_TYPES_SORTED promises that edge_type is non-decreasing (the
 * reference's layout, utils.py:168-198), which saves the sort of the dD pass.  sigmoid_scores (the probabilities the
 * forward returned, decoder.py:23) makes grad_logit the gradient with respect to those: the factor p (1 - p) is
 * applied where the records are built.  type_offsets ([R + 1] int32 on the device, with GN_DM_TYPES_SORTED): the first
 * edge of every relation, for callers that keep it with a static edge_type instead of having it searched per call.
 * GN_DM_TYPE_TASKS: type_offsets points at a buffer that gn_distmult_type_tasks filled for this (R, E) - the offsets and,
 * behind them, the task list of the relation-major reduction, which then is not rebuilt per call (a one-workgroup launch
 * of 7 us).  Node and relation tables that fit the LDS in 16-column blocks ((n + R) * 64 B <= 150 KB) are reduced from
 * there; larger ones from L2. */
#define GN_DM_TYPES_SORTED 1
#define GN_DM_TYPE_TASKS 2
GN_API size_t gn_distmult_type_tasks_bytes(int64_t num_relations, int64_t num_edges);
/* out (16-byte aligned, gn_distmult_type_tasks_bytes bytes; may be where type_offsets already sits): type_offsets [R + 1],
 * then the task list. */
GN_API gn_status gn_distmult_type_tasks(const int32_t* type_offsets, int64_t num_relations, int64_t num_edges, void* out,
                                 size_t out_bytes, void* stream);
/* The same gradients from the triples as the negative sampler leaves them next to its int64 output
 * (gn_negative_sampler_sample_packed / _stepped): packed_uv[e] = u | v << 16 and a 16-bit relation id per position (the
 * static edge_type, narrowed once by the caller) - 6 instead of 24 bytes per edge in each pass of the counting sort.  Takes
 * the counting-sort / LDS path only: GN_DM_TYPES_SORTED with type_offsets, tables that fit the LDS, at most 4,096 nodes;
 * GN_ERR_UNSUPPORTED otherwise (call gn_distmult_backward_ex_f32 on the int64 arrays).  Same bits as the int64 call. */
GN_API gn_status gn_distmult_backward_packed_f32(const float* z, int64_t ld_z, int64_t num_nodes, int64_t num_features,
                                          const uint32_t* packed_uv, const uint16_t* rel16, const float* d, int64_t ld_d,
                                          int64_t num_relations, int64_t num_edges, const float* grad_logit, float* dz,
                                          int64_t ld_dz, float* dd, int64_t ld_dd, int flags, const float* sigmoid_scores,
                                          const int32_t* type_offsets, void* workspace, size_t workspace_bytes, void* stream);
GN_API size_t gn_distmult_backward_workspace_bytes(int64_t num_nodes, int64_t num_features, int64_t num_relations,
                                            int64_t num_edges);
GN_API gn_status gn_distmult_backward_f32(const float* z, int64_t ld_z, int64_t num_nodes, int64_t num_features,
                                   const int64_t* u, const int64_t* v, const int64_t* edge_type, const float* d,
                                   int64_t ld_d, int64_t num_relations, int64_t num_edges, const float* grad_logit,
                                   float* dz, int64_t ld_dz, float* dd, int64_t ld_dd, void* workspace,
                                   size_t workspace_bytes, void* stream);
GN_API gn_status gn_distmult_backward_ex_f32(const float* z, int64_t ld_z, int64_t num_nodes, int64_t num_features,
                                      const int64_t* u, const int64_t* v, const int64_t* edge_type, const float* d,
                                      int64_t ld_d, int64_t num_relations, int64_t num_edges, const float* grad_logit,
                                      float* dz, int64_t ld_dz, float* dd, int64_t ld_dd, int flags,
                                      const float* sigmoid_scores /* nullable */,
                                      const int32_t* type_offsets /* nullable */, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* The same gradients for a STATIC edge list (the positive edges, scored with the same tensors every epoch,
 * GripNet-pose.py:137).  What the sort and the reductions derive from the triples alone is built once: the two
 * directions of an edge (same unordered node pair and relation) are paired up - their records differ only in g, so a
 * pair is reduced as one triple with g1 + g2 -, and the scanned offsets of the counting sort and the task lists of
 * what is left are kept; a call sums the pairs' gradients, starts at the scatter pass and runs the two segment
 * reductions (6 launches instead of 12, on half the records for a bidirectional list).  Create copies the triples to
 * the host, validates them (GN_ERR_INDEX_RANGE) and synchronises `stream`; GN_ERR_UNSUPPORTED when the tables do not
 * fit the LDS path or edge_type is not sorted (create), when rows are not aligned float4 columns or the list is too
 * long for the staged scatter (backward): use gn_distmult_backward_f32 then.  Results equal the plan-less call up to
 * the rounding of g1 + g2 and are bitwise reproducible. */
typedef struct gn_distmult_bwd_plan gn_distmult_bwd_plan;
GN_API gn_status gn_distmult_bwd_plan_create(const int64_t* u, const int64_t* v, const int64_t* edge_type, int64_t num_edges,
                                      int64_t num_nodes, int64_t num_relations, void* stream, gn_distmult_bwd_plan** plan);
GN_API void gn_distmult_bwd_plan_destroy(gn_distmult_bwd_plan* plan);
GN_API size_t gn_distmult_bwd_plan_workspace_bytes(const gn_distmult_bwd_plan* plan, int64_t num_features);
GN_API gn_status gn_distmult_backward_planned_f32(const gn_distmult_bwd_plan* plan, const float* z, int64_t ld_z,
                                           int64_t num_features, const float* d, int64_t ld_d,
                                           const float* grad_logit, const float* sigmoid_scores /* nullable */,
                                           float* dz, int64_t ld_dz, float* dd, int64_t ld_dd, void* workspace,
                                           size_t workspace_bytes, void* stream);

/* The decoder's backward fed by the LINK LOSS directly (round 6): the scores of a list go into the loss of
 * GripNet-pose.py:140-142, -mean(log(p + eps)) over the positives' list or -mean(log(1 - p + eps)) over the negatives', and
 * d loss / d s_e = (-+ upstream / num_edges) / (p_e (or 1 - p_e) + eps) * p_e (1 - p_e) is a function of the edge's own probability
 * only - it is computed where the edge records are built, in the arithmetic of gn_link_loss_backward_f32 followed by the
 * sigmoid factor (the SAME bits as the two-step path), and the loss's backward launch and the [E] gradient vector's round
 * trip through memory disappear.  upstream: the loss's one upstream gradient on the device (NULL: 1).
 * dz_addend / dd_addend (packed form): the gradients of the SAME z and D from the step's other list (the positives' call): added
 * where this call stores its sums (dz = sums + dz_addend, may alias dz), so that the two lists' gradients need no adding launch;
 * GN_ERR_UNSUPPORTED where the two reductions do not share their combine launch (call without addends and add). */
typedef struct gn_link_loss_grad {
    const float* upstream;
    float eps;
    int negative;              /* 0: the list's scores are the loss's positives, 1: its negatives */
} gn_link_loss_grad;
GN_API gn_status gn_distmult_backward_loss_packed_f32(const float* z, int64_t ld_z, int64_t num_nodes, int64_t num_features,
                                               const uint32_t* packed_uv, const uint16_t* rel16, const float* d, int64_t ld_d,
                                               int64_t num_relations, int64_t num_edges, const gn_link_loss_grad* loss,
                                               const float* sigmoid_scores, float* dz, int64_t ld_dz, float* dd, int64_t ld_dd,
                                               int flags, const int32_t* type_offsets,
                                               const float* dz_addend /* nullable */, int64_t ld_dz_addend,
                                               const float* dd_addend /* nullable */, int64_t ld_dd_addend,
                                               void* workspace, size_t workspace_bytes, void* stream);
GN_API gn_status gn_distmult_backward_loss_planned_f32(const gn_distmult_bwd_plan* plan, const float* z, int64_t ld_z,
                                                int64_t num_features, const float* d, int64_t ld_d,
                                                const gn_link_loss_grad* loss, const float* sigmoid_scores,
                                                float* dz, int64_t ld_dz, float* dd, int64_t ld_dd, void* workspace,
                                                size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Typed negative sampling on the device (replaces gripnet/utils.py:98-119 and its per-epoch host round
 * trip, GripNet-pose.py:131).  The sampler sorts the positive pairs of every relation block once
 * (range_list on the host, [R,2], must tile [0,E)); gn_negative_sampler_sample then writes, for every
 * positive edge e of block r, one pair (u, v) drawn uniformly (with replacement) from the n^2 pairs
 * that are not positives of block r.  Deterministic in `seed`.  Bit 1 of *error_flag (device int32,
 * may be NULL) is set if a block's positives leave (almost) no pair to draw. */
typedef struct gn_negative_sampler gn_negative_sampler;
GN_API gn_status gn_negative_sampler_create(const int64_t* u, const int64_t* v, const int64_t* range_list_host,
                                     int64_t num_relations, int64_t num_edges, int64_t num_nodes, void* stream,
                                     gn_negative_sampler** sampler);
GN_API void gn_negative_sampler_destroy(gn_negative_sampler* sampler);
GN_API gn_status gn_negative_sampler_sample(const gn_negative_sampler* sampler, uint64_t seed, int64_t* out_u,
                                     int64_t* out_v, int32_t* error_flag, void* stream);
/* The same draw, which also leaves every pair as one 32-bit word u | v << 16 (graphs of up to 65,535 nodes) for
 * gn_distmult_packed_forward_f32: the sampler produces the pairs, so it can hand the decoder 4 bytes per edge instead
 * of 16.  GN_ERR_UNSUPPORTED for larger graphs. */
GN_API gn_status gn_negative_sampler_sample_packed(const gn_negative_sampler* sampler, uint64_t seed, int64_t* out_u,
                                            int64_t* out_v, uint32_t* packed_uv, int32_t* error_flag, void* stream);
/* The draw of a RECORDED training loop (a hipGraph replays its launch arguments): the seed of the draw is seed + *step, with
 * `step` a 64-bit counter in device memory (8-byte aligned, set by the caller once) that a one-thread launch behind the draw
 * advances by one.  Replay k of a captured call therefore writes what gn_negative_sampler_sample(_packed) writes for
 * seed + step0 + k: new negatives every epoch (GripNet-pose.py:131) without a launch outside the graph.  packed_uv may be
 * NULL.  Stepped draws of ONE sampler must not run concurrently (they share the sampler's arrival counter): one stream, or
 * one sampler per stream. */
GN_API gn_status gn_negative_sampler_sample_stepped(const gn_negative_sampler* sampler, uint64_t seed, uint64_t* step, int64_t* out_u,
                                             int64_t* out_v, uint32_t* packed_uv, int32_t* error_flag, void* stream);

/* Independent small products in ONE launch.  Between gn_dense_batch_begin() and gn_dense_batch_end(stream) (per host thread,
 * not nested) the calls that carry GN_GEMM_JOIN_BATCH / GN_XTG_JOIN_BATCH - those of gn_gemm_f32 that take the deep-and-narrow kernel (a single product with at most 64 rows or 32 columns
 * of output and A given transposed or K >= 256) or the tall-skinny fp32 kernel (a single product of >= 256 rows whose B fits 64 KB
 * of LDS, no row gather; a product of >= 2048 rows with K a multiple of 32 takes the split-bf16 kernel instead and launches at once) and of gn_xtg_f32 that take the one-launch kernel are queued instead of launched,
 * and leave together at _end as one grid (up to four per launch; a product alone in its batch launches as usual); every other
 * call inside the bracket (without the flag, of another shape, or made by the library inside another entry point) launches at once.  The caller promises that the queued products neither depend on each other nor
 * share a workspace (a second gn_xtg_f32 on a queued workspace launches at once), and that their operands stay valid until
 * _end.  The dense products of the relational layer's backward (dbasis, datt, droot, dx += g root^T: layers.py:165-197 under
 * autograd) are four ~10 us launches, or one of 13 us; a GCN layer's dx and dW (layers.py:71-100) two, or one. */
GN_API gn_status gn_dense_batch_begin(void);
GN_API gn_status gn_dense_batch_end(void* stream);

/* The element-wise glue in front of a layer's backward pass, in one launch (autograd of layers.py:71-100,165-197 with the
 * ReLU of layers.py:279,305,370):  gm = saved_out > 0 ? g : 0 (saved_out NULL: gm = g);  gd = gm / rowdiv[row] (the mean
 * of the relational layer, layers.py:191);  colsum[c] = sum over the rows of gm[:, c] (the bias gradient).  gm, gd and
 * colsum are each optional (NULL).  Deterministic (fixed row slices, added in workgroup order by the last workgroup to
 * arrive).  `workspace` (needed for colsum): gn_grad_prologue_workspace_bytes() bytes, 4-byte aligned, ZEROED ONCE by the
 * caller; calls on one workspace must be stream-ordered.  Any width (one launch per block of 256 columns). */
GN_API size_t gn_grad_prologue_workspace_bytes(void);
GN_API gn_status gn_grad_prologue_f32(const float* g, int64_t ld_g, const float* saved_out, int64_t ld_saved, const float* rowdiv,
                               int64_t rows, int64_t cols, float* gm, int64_t ld_gm, float* gd, int64_t ld_gd, float* colsum,
                               void* workspace, size_t workspace_bytes, void* stream);

/* One Adam step over all parameter tensors in one launch (torch.optim.Adam(model.parameters(), lr) + optimizer.step(),
 * GripNet-pose.py:104,146; the update of torch/optim/adam.py without amsgrad / maximize):
 *   t = *step + 1;  g += weight_decay p;  m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g g;
 *   p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps);   *step = t.
 * `tensors` is a HOST array (copied into the launch's arguments, so a captured launch replays with it): device pointers
 * of parameter, gradient, first and second moment, and the element count.  `step`: one float on the device, the number of
 * steps taken so far (0 before the first), advanced by the launch.  `workspace`: 4 bytes on the device, 4-byte aligned,
 * ZEROED ONCE by the caller.  Calls on one (step, workspace) must be stream-ordered. */
#define GN_ADAM_MAX_TENSORS 64   /* per launch; longer tables take several launches */
typedef struct gn_adam_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    int64_t numel;
} gn_adam_tensor;
GN_API gn_status gn_adam_step_f32(const gn_adam_tensor* tensors, int num_tensors, float* step, void* workspace, size_t workspace_bytes,
                           float lr, float beta1, float beta2, float eps, float weight_decay, void* stream);

/* The link-prediction loss of the training loop and its gradient (GripNet-pose.py:140-142 with EPS of gripnet/utils.py:10):
 *   loss = - mean(log(pos + eps)) - mean(log(1 - neg + eps)),
 *   dpos[i] = - g / (num_pos (pos[i] + eps)),   dneg[i] = g / (num_neg (1 - neg[i] + eps)),   g = *upstream_grad (NULL: 1).
 * One launch each (written with torch ops: ~20 element-wise and reduction launches per step).  Deterministic: fixed
 * slices summed in a fixed order, partial sums added in workgroup order by the last workgroup to arrive.  `workspace`:
 * gn_link_loss_workspace_bytes() bytes, 8-byte aligned, ZEROED ONCE by the caller (the launch leaves it ready for the
 * next one); calls on one workspace must be stream-ordered. */
GN_API size_t gn_link_loss_workspace_bytes(void);
/* The node-classification loss of the NC training loops and its gradient (GripNet-aminer.py:133 and every freebase driver):
 *   loss = - mean_i log(score[i, classes[i]] + eps),   dscore[i, c] = c == classes[i] ? - g / (n (score[i, c] + eps)) : 0.
 * One launch each (the torch expression is an advanced-indexing gather, a log, a mean and their backward kernels).
 * A class id outside [0, num_classes) sets bit 0 of *error_flag (nullable) and contributes nothing. */
GN_API gn_status gn_class_loss_forward_f32(const float* score, int64_t ld_score, const int64_t* classes, int64_t num_nodes,
                                    int64_t num_classes, float eps, float* loss, int32_t* error_flag, void* stream);
GN_API gn_status gn_class_loss_backward_f32(const float* score, int64_t ld_score, const int64_t* classes, int64_t num_nodes,
                                     int64_t num_classes, float eps, const float* upstream_grad, float* dscore, int64_t ld_dscore,
                                     void* stream);
GN_API gn_status gn_link_loss_forward_f32(const float* pos_score, int64_t num_pos, const float* neg_score, int64_t num_neg, float eps,
                                   float* loss, void* workspace, size_t workspace_bytes, void* stream);
GN_API gn_status gn_link_loss_backward_f32(const float* pos_score, int64_t num_pos, const float* neg_score, int64_t num_neg, float eps,
                                    const float* upstream_grad /* device, nullable */, float* dpos, float* dneg, void* stream);

/* ---------------------------------------------------------------------------------------
 * Per-relation evaluation metrics (replaces the R scikit-learn calls and device -> host copies per epoch
 * of GripNet-pose.py:148-160,188-199 / gripnet/utils.py:28-35).  pos_score / neg_score hold E scores
 * each, segmented by the same range_list ([R,2] on the host).  out is [3, R] float64 on the device:
 * row 0 area under the precision-recall curve (trapezoid, with the point recall 0 / precision 1),
 * row 1 ROC AUC, row 2 average precision, per relation, with scikit-learn's treatment of tied scores;
 * NaN for a relation without edges. */
typedef struct gn_link_metrics_plan gn_link_metrics_plan;
/* What depends on the range list only (the (relation, class) segments of the two score vectors, their chunk and tile maps):
 * built once per list (synchronises `stream` once); a planned call is asynchronous - chunk sort in LDS, merge rounds for the
 * relations longer than 4,096 edges, the curves' terms from binary searches between a relation's two sorted classes, a fold in
 * fixed order (bitwise reproducible).  workspace: gn_link_metrics_plan_workspace_bytes. */
GN_API gn_status gn_link_metrics_plan_create(const int64_t* range_list_host, int64_t num_relations, int64_t num_edges, void* stream,
                                      gn_link_metrics_plan** plan);
GN_API void gn_link_metrics_plan_destroy(gn_link_metrics_plan* plan);
GN_API size_t gn_link_metrics_plan_workspace_bytes(const gn_link_metrics_plan* plan);
GN_API gn_status gn_link_metrics_planned_f32(const gn_link_metrics_plan* plan, const float* pos_score, const float* neg_score,
                                      double* out, void* workspace, size_t workspace_bytes, void* stream);
/* The same without a kept plan (builds one, uses it, drops it: synchronises `stream`). */
GN_API size_t gn_link_metrics_workspace_bytes(int64_t num_relations, int64_t num_edges);
GN_API gn_status gn_link_metrics_f32(const float* pos_score, const float* neg_score, const int64_t* range_list_host,
                              int64_t num_relations, int64_t num_edges, double* out, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * One-shot direct exchange: the all-reduce of the sharded relational layer's partial sums ([n_d, 32] floats,
 * 82,560 B on PoSE: latency-bound) as ONE hop over xGMI's point-to-point links instead of a ring's 2 (G - 1): every rank
 * writes its partial into its slot of every peer's buffer (peer-mapped device memory: hipIpc handles exchanged by the caller),
 * raises a flag, waits for its own G flags on the device (bounded: timeout_ms, then bit 2 of *error_flag) and adds the G
 * slots in RANK ORDER - every rank holds the same bits.  Slot buffer per rank: gn_exchange_buffer_bytes(n, world) bytes
 * ([2][world][n] floats, slots alternate by step parity); flags: [world] int32, zero before step 1; ticket: one zeroed
 * uint32; step = 1, 2, 3, ... the same on every rank.  No host synchronisation. */
GN_API size_t gn_exchange_buffer_bytes(int64_t n, int world);
GN_API gn_status gn_exchange_push_f32(const float* src, int64_t n, void* const* peer_slots /* [world] device pointers, host array */,
                               void* const* peer_flags /* [world] */, unsigned int* ticket, int world, int rank, int step, void* stream);
GN_API gn_status gn_exchange_wait_sum_f32(const void* my_slots, const void* my_flags, int64_t n, int world, int step, float* dst,
                                   int timeout_ms, int32_t* error_flag, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GRIPNET_HIP_H */
