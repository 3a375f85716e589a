/*
 * spr.h -- C ABI of libspr_hip.so, the MI355X (gfx950) implementation of the
 * Superpoints_Registration hot path.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - varlen batches are packed [sum N, C] row-major with cu_seqlens
 *     int32[n_seg + 1] (cu[0] = 0) on the device; clouds are stacked
 *     [src_0..src_{B-1}, tgt_0..tgt_{B-1}] as in qk_regtr_full.py:152;
 *   - `stream` is a hipStream_t passed as void*; all work is stream ordered,
 *     nothing synchronises, nothing allocates: scratch comes from the caller
 *     (`ws`, size from the matching *_workspace_bytes query);
 *   - return value 0 = success, anything else = failure; spr_last_error()
 *     returns a thread-local message.  The Python binding raises RuntimeError,
 *     which is what the reference's CPython modules raise
 *     (cpp_neighbors/wrapper.cpp:201-205, cpp_subsampling/wrapper.cpp:266-270).
 *
 * Each function cites the reference interface it replaces (paths relative to
 * /root/reference/src).
 */
#ifndef SPR_H_
#define SPR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPR_VERSION 5

/* activation codes for spr_linear / spr_instnorm */
#define SPR_ACT_NONE 0
#define SPR_ACT_RELU 1
#define SPR_ACT_SIGMOID 2

/* output ordering of spr_grid_subsample */
#define SPR_ORDER_REFERENCE 0 /* libstdc++ unordered_map iteration order     */
#define SPR_ORDER_CANONICAL 1 /* ascending (cloud, voxel key)                 */

int spr_version(void);
const char* spr_last_error(void);

/* ---- a1: grid subsampling ------------------------------------------------
 * Replaces grid_subsampling.subsample_batch(points, batches, sampleDl=, max_p=)
 * (models/backbone_kpconv/cpp_wrappers/cpp_subsampling/wrapper.cpp:62-82 ->
 *  grid_subsampling/grid_subsampling.cpp:109 batch_grid_subsampling), called
 * from models/backbone_kpconv/kpconv.py:174-185 (and the GPU variant :217).
 * Barycentres are bit-exact float32 (in-order sum * (float)(1.0/count)).
 *   xyz      [n,3] f32        cu [nb+1] i32
 *   out_xyz  [n,3] f32 (first *out_total rows valid)
 *   out_lens [nb] i32         out_total [1] i32 (both device)
 */
size_t spr_grid_subsample_workspace_bytes(int n, int nb);
int spr_grid_subsample(const float* xyz, const int* cu, int n, int nb, float dl,
                       int max_p, int order_mode, float* out_xyz, int* out_lens,
                       int* out_total, void* ws, size_t ws_bytes, void* stream);

/* ---- voxel pre-downsampling, one point per voxel ------------
 * Replaces voxel_down_sample of the KITTI loader (data_loaders/kitti_pred.py:12-14,
 * :203-204 -> kiss_icp): voxel = trunc(p / voxel_size) per axis in float64, the first point of
 * every voxel is kept.  out_idx [n] i32: the first out_count[0] entries = kept
 * original indices, ascending; out_count[0] = -1 when a coordinate exceeds 2^20 voxels.
 */
size_t spr_voxel_downsample_workspace_bytes(int n);
int spr_voxel_downsample(const float* xyz, int n, double voxel_size, int* out_idx,
                         int* out_count, void* ws, size_t ws_bytes, void* stream);

/* ---- a2: batched fixed-radius neighbours ---------------------------------
 * Replaces radius_neighbors.batch_query(queries, supports, q_batches,
 * s_batches, radius=) (cpp_wrappers/cpp_neighbors/wrapper.cpp:58-75 ->
 * neighbors/neighbors.cpp:211 batch_nanoflann_neighbors) plus the
 * [:, :max_neighbors] slice of kpconv.py:258-262.
 * Rows: supports of the same cloud with d2 < r*r (float32, exact reference
 * arithmetic), ascending (d2, index), global indices, padded with ns.
 *   out_idx [nq, limit] i32, 1 <= limit <= 128; max_count [1] i32 device = untruncated max row
 *   count (the reference's row width); the caller may slice to
 *   min(max_count, limit).
 * algo 0: dense per-cloud cell table (fast; *max_count = -2 if the clouds'
 *   bounding boxes need more cells than the table holds -> retry with algo 1);
 * algo 1: sorted cell keys + binary search (no geometry limit besides 8191
 *   cells per axis, *max_count = -1).  Results are identical.
 */
size_t spr_radius_neighbors_workspace_bytes(int nq, int ns, int nb);
int spr_radius_neighbors(const float* q_xyz, const int* q_cu, int nq,
                         const float* s_xyz, const int* s_cu, int ns, int nb,
                         float radius, int limit, int algo, int* out_idx,
                         int* max_count, void* ws, size_t ws_bytes, void* stream);

/* Cell table of spr_radius_neighbors (algo 0) as an object: built once per (supports, radius),
 * queried several times -- the pyramid runs the conv search, the pool search and the previous
 * level's up-sampling search against the same supports with the same radius (kpconv.py:352,
 * :377, :384).  Rows are those of spr_radius_neighbors, entry for entry.
 *   table: spr_radius_table_bytes(ns, nb) bytes, 256-byte aligned, owned by the caller;
 *   self != 0: the queries are the table's supports (same array and cu) -> cell-order walk;
 *   slot in [0, spr_radius_table_slots()): one per query call against a build (its max row count);
 *   *max_count as in spr_radius_neighbors (-2: geometry too large for the table -> algo 1).
 */
size_t spr_radius_table_bytes(int ns, int nb);
size_t spr_radius_table_build_workspace_bytes(int ns, int nb);
size_t spr_radius_table_query_workspace_bytes(int nq);
int spr_radius_table_slots(void);
int spr_radius_table_build(const float* s_xyz, const int* s_cu, int ns, int nb, float radius,
                           void* table, size_t table_bytes, void* ws, size_t ws_bytes, void* stream);
int spr_radius_table_query(const float* q_xyz, const int* q_cu, int nq, int self, int ns, int nb,
                           float radius, int limit, int slot, const void* table, int* out_idx,
                           int* max_count, void* ws, size_t ws_bytes, void* stream);
/* Round 5: the same with the K-nearest selection chosen by the caller -- algo 0: one thread per query (cell scan into
 * a scratch row + rank sort in registers; cheaper for sparse rows), algo 1: one wave per query (one pass over the
 * coalesced record runs, the row kept sorted in the wave's registers, no scratch, no sort kernel; faster when more
 * supports lie in range than `limit`: LiDAR-shaped clouds).  Identical rows. */
int spr_radius_table_query_a(const float* q_xyz, const int* q_cu, int nq, int self, int ns, int nb,
                             float radius, int limit, int slot, const void* table, int* out_idx,
                             int* max_count, int algo, void* ws, size_t ws_bytes, void* stream);

/* ---- a4: KPConv forward ---------------------------------------------------
 * Replaces KPConv.forward(q_pts, s_pts, neighb_inds, x)
 * (models/backbone_kpconv/kpconv_blocks.py:269-414; rigid kernel, linear
 * influence, 'sum' aggregation -- the only mode any shipped config selects).
 *   nbr [nq, nbr_stride] i32, first `kmax` columns used, shadow index = ns
 *   x [ns, cin]  weights [n_kp, cin, cout]  kernel_points [n_kp, 3]
 *   out [nq, cout]
 * rows_sorted != 0 promises that shadow entries only trail valid ones (true
 * for spr_radius_neighbors output) and enables early exit.
 * impl: 0 = default (ring kernel for 32/64-channel inputs, streamed MFMA tile kernel otherwise),
 *       1 = simple reference kernel, 2 = the streamed MFMA tile kernel for every MFMA shape (A/B).
 */
size_t spr_kpconv_workspace_bytes(int nq, int ns, int cin, int cout);
int spr_kpconv_fwd(const float* q_xyz, int nq, const float* s_xyz, int ns,
                   const int* nbr, int nbr_stride, int kmax, int rows_sorted,
                   const float* x, int cin, const float* weights, int cout,
                   const float* kernel_points, int n_kp, float kp_extent,
                   float* out, int impl, void* ws, size_t ws_bytes,
                   void* stream);

/* ---- a5: per-cloud InstanceNorm (+ residual add) (+ LeakyReLU) ------------
 * Replaces BatchNormBlock.forward (kpconv_blocks.py:497-525: per-cloud
 * nn.InstanceNorm1d, affine=False, eps, biased variance) fused with the
 * LeakyReLU that follows it (kpconv_blocks.py:553-561, :645, :727) and with
 * the residual add of ResnetBottleneckBlock (:741):
 *   out = lrelu(IN(x) + add, slope)      slope = 1 -> no activation
 * x,out [n,c] (may alias); add [n,c] or NULL; cu [nb+1]; max_len_host = an
 * upper bound of the longest cloud (sizes the statistics grid: fixed 512-row
 * slices relative to each cloud, so results are batch-invariant bit for bit).
 * norm = 0 skips the normalisation (out = lrelu(x + add)).
 */
size_t spr_instnorm_workspace_bytes(int max_len_host, int nb, int c);
int spr_instnorm(const float* x, const int* cu, int n, int nb, int max_len_host,
                 int c, float eps, int norm, const float* add, float slope,
                 float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- a5: tail of a ResNet bottleneck block, fused -----------------------------
 * Replaces the last three statements of ResnetBottleneckBlock.forward
 * (kpconv_blocks.py:733-741): unary2 (Linear without bias -> per-cloud
 * InstanceNorm, :556-561), unary_shortcut (the same, or the identity) and
 * leaky_relu(x + shortcut):
 *   out = lrelu(IN(xa wa^T) + (kb > 0 ? IN(xb wb^T) : add), slope)
 * xa [n,ka], wa [n_out,ka]; xb [n,kb], wb [n_out,kb] (kb = 0: xb = wb = NULL and
 * add [n,n_out] or NULL is added instead); cu [nb+1].  The projections are computed
 * twice (statistics pass, output pass) and never written: only `out` goes to memory.
 * Split-fp16 product mode only (spr_set_gemm_mode(1)); ranges as in spr_linear_r
 * (NULL: measured here); out_range as in spr_instnorm_r.
 * spr_block_tail_tile_rows: rows per statistics tile of a supported (ka, kb, n_out), 0 if
 * the shape has no kernel (the caller then uses spr_linear_r + spr_instnorm_r);
 * spr_block_tail_tiles: the tile table of a batch (int32 [spr_block_tail_tiles_len], 16-byte
 * aligned): the exclusive prefix of ceil(len / tile_rows) per cloud and one {first row, valid
 * rows, cloud} record per tile -- tiles start at each cloud's first row, so a cloud's result
 * never depends on its batch mates.  Depends on cu and tile_rows only: build once per level.
 */
int spr_block_tail_tile_rows(int ka, int kb, int n_out);
size_t spr_block_tail_tiles_len(int n, int nb, int tile_rows);
int spr_block_tail_tiles(const int* cu, int n, int nb, int tile_rows, int* tiles, void* stream);
size_t spr_block_tail_workspace_bytes(int n, int nb, int kb, int n_out, int tile_rows);
int spr_block_tail(const float* xa, int ka, const float* wa, const float* xb, int kb,
                   const float* wb, const float* add, const int* cu, const int* tiles, int n,
                   int nb, int n_out, float eps, float slope, float* out,
                   const float* xa_range, int xa_range_n, const float* wa_range, int wa_range_n,
                   const float* xb_range, int xb_range_n, const float* wb_range, int wb_range_n,
                   float* out_range, int out_range_n, void* ws, size_t ws_bytes, void* stream);
/* Round 5: the same with xa = the RAW input of a per-cloud InstanceNorm + LeakyReLU(xa_slope) (the KPConv output of a
 * bottleneck block, kpconv_blocks.py:717-719 of the reference) whose statistics the caller holds (spr_instnorm_stats:
 * xa_mean, xa_rstd [nb][ka]); the normalisation runs while a tile is staged -- the normalised tensor is never written.
 * xa_range (required) bounds the NORMALISED values: sqrt(longest cloud) is always valid. */
int spr_block_tail_n(const float* xa, int ka, const float* wa, const float* xb, int kb, const float* wb,
                     const float* add, const int* cu, const int* tiles, int n, int nb, int n_out, float eps,
                     float slope, float* out, const float* xa_range, int xa_range_n, const float* wa_range,
                     int wa_range_n, const float* xb_range, int xb_range_n, const float* wb_range,
                     int wb_range_n, float* out_range, int out_range_n, const float* xa_mean,
                     const float* xa_rstd, float xa_slope, void* ws, size_t ws_bytes, void* stream);
/* mean / rstd [nb][c] of spr_instnorm's statistics passes alone (workspace: spr_instnorm_workspace_bytes). */
int spr_instnorm_stats(const float* x, const int* cu, int n, int nb, int max_len_host, int c, float eps,
                       float* mean, float* rstd, void* ws, size_t ws_bytes, void* stream);

/* ---- a5: strided max pooling ----------------------------------------------
 * Replaces max_pool(x, inds) (kpconv_blocks.py:127-143): max over the pooling
 * neighbours, shadow index ns reads a zero row.  idx [nq, idx_stride], first
 * k columns used.
 */
int spr_maxpool_gather(const float* x, int ns, int c, const int* idx, int nq,
                       int idx_stride, int k, float* out, void* stream);

/* ---- dense projection -------------------------------------------------------
 * out[m,n] = act(x[m,k] @ w[n,k]^T + bias[n] + residual[m,n])
 * Replaces the nn.Linear calls on the path: UnaryBlock.mlp
 * (kpconv_blocks.py:549,:557), feat_proj / overlap_predictor
 * (qk_regtr_full.py:47,:85,:176,:248), MultiheadAttention in/out projections
 * and linear1/linear2 (transformer/transformers.py:96-104).
 * bias, residual may be NULL.  k must be a multiple of 32 unless n <= 64.
 * ws: spr_linear_workspace_bytes() bytes of device scratch (the operands'
 * max-|x| partials; may be NULL in mode 0).
 */
size_t spr_linear_workspace_bytes(void);
int spr_linear(const float* x, int m, int k, const float* w, int n,
               const float* bias, const float* residual, int act, float* out,
               void* ws, size_t ws_bytes, void* stream);
/* Operand-range hand-over.  The split-fp16 arithmetic scales every operand tensor by a power of
 * two derived from max |x|; by default spr_linear measures it with a pass over x.  A producer
 * that has just written x can publish the range instead, as an array of per-workgroup partial
 * maxima in device memory (any count; their maximum must bound max |x| from above):
 *   x_range / x_range_n     range of x (NULL: measure);
 *   w_range / w_range_n     range of the weights (NULL: measure).  Weights do not change between
 *     inference calls: measure them once with spr_absmax (spr_range_parts() floats) and pass
 *     the result on every call; with both ranges handed in the call has no pre-pass at all;
 *   out_range (capacity out_range_cap floats): if the GEMM runs with <= out_range_cap
 *     workgroups it writes one partial per workgroup and sets *out_range_n_host (a HOST int)
 *     to their number, else 0 (nothing published).
 * spr_layernorm_r publishes the ranges of its two outputs into spr_layernorm_range_count(m)
 * slots each, which the CALLER MUST ZERO beforehand (the kernel combines its workgroups'
 * maxima with atomic max); spr_attn_inproj_varlen_fwd_r accepts the input ranges and publishes a bound of
 * its output (1 float: the attention output is a convex combination of value rows).
 * spr_instnorm_r and spr_maxpool_gather_r publish |out| the same way as spr_layernorm_r
 * (out_range_n zeroed slots, a power of two chosen by the caller); spr_kpconv_fwd_r accepts x_range / w_range. */
int spr_range_parts(void);
int spr_absmax(const float* x, long rows, int cols, long stride, float* parts, void* stream);
/* max |x| of many contiguous tensors in ONE launch (all weights of a model at the start of a training step).
 * jobs_dev: njobs device records {const float* x; long long n_elements;}; parts_out [njobs][parts_per_job]:
 * row j is a range (parts_per_job partials) of tensor j, usable wherever spr_absmax partials are. */
int spr_absmax_multi(const void* jobs_dev, int njobs, int parts_per_job, float* parts_out, void* stream);
int spr_linear_r(const float* x, int m, int k, const float* w, int n, const float* bias,
                 const float* residual, int act, float* out, const float* x_range,
                 int x_range_n, const float* w_range, int w_range_n, float* out_range,
                 int out_range_cap, int* out_range_n_host, void* ws, size_t ws_bytes, void* stream);
int spr_kpconv_fwd_r(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                     int nbr_stride, int kmax, int rows_sorted, const float* x, int cin,
                     const float* weights, int cout, const float* kernel_points, int n_kp,
                     float kp_extent, float* out, int impl, const float* x_range, int x_range_n,
                     const float* w_range, int w_range_n, void* ws, size_t ws_bytes, void* stream);
/* Static inputs of the ring KPConv (csrc/kpconv.hip, k_kpconv_ring), hoisted out of the per-call
 * path.  Both are pure functions of their arguments; callers cache them per neighbour matrix / per
 * weight version (ops.py does).  The reference recomputes everything per call
 * (kpconv_blocks.py:269-414); these have no counterpart there.
 *   spr_kpconv_plan: tile descriptors of a neighbour matrix -- tiles of 16 queries (in `order` if
 *     given: a permutation of [0, nq), e.g. a spatial order; NULL = natural order), the 16 queries of
 *     a tile dealt to the eight waves by live neighbour-block count.  The plan depends on
 *     (nbr contents, nq, ns, nbr_stride, kmax, rows_sorted, order) only.
 *   spr_kpconv_prep_weights: the [15, cin, cout] weights as range-scaled split-fp16 planes in
 *     MFMA-fragment order; w_range = spr_absmax partials of the same weights (required).
 *   spr_kpconv_fwd_p: spr_kpconv_fwd_r with plan / wplanes handed in (NULL = built per call). */
size_t spr_kpconv_plan_bytes(int nq);
int spr_kpconv_plan(const int* nbr, int nq, int ns, int nbr_stride, int kmax, int rows_sorted,
                    const int* order, void* plan, size_t plan_bytes, void* stream);
size_t spr_kpconv_wplanes_bytes(int cin, int cout);
int spr_kpconv_prep_weights(const float* weights, int n_kp, int cin, int cout, const float* w_range,
                            int w_range_n, void* wplanes, size_t wplanes_bytes, void* stream);
int spr_kpconv_fwd_p(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                     int nbr_stride, int kmax, int rows_sorted, const float* x, int cin,
                     const float* weights, int cout, const float* kernel_points, int n_kp,
                     float kp_extent, float* out, int impl, const float* x_range, int x_range_n,
                     const float* w_range, int w_range_n, const void* plan, const void* wplanes,
                     void* ws, size_t ws_bytes, void* stream);
int spr_instnorm_r(const float* x, const int* cu, int n, int nb, int max_len_host, int c,
                   float eps, int norm, const float* add, float slope, float* out,
                   float* out_range, int out_range_n, void* ws, size_t ws_bytes, void* stream);
int spr_maxpool_gather_r(const float* x, int ns, int c, const int* idx, int nq, int idx_stride,
                         int k, float* out, float* out_range, int out_range_n, void* stream);
/* The same rows with the queries WALKED in `order` (int32 [nq], a permutation: spr_cell_order of the query points;
 * NULL = spr_maxpool_gather_r).  Results are identical; queries in flight together then share support rows, which L2
 * serves instead of HBM (reference: kpconv_blocks.py:127-140 max_pool, no counterpart of the order there). */
int spr_maxpool_gather_o(const float* x, int ns, int c, const int* idx, int nq, int idx_stride, int k,
                         const int* order, float* out, float* out_range, int out_range_n, void* stream);
/* order [n] = the points of every cloud (cu [nb + 1]) sorted by the Morton code of their cell of size `cell`
 * (clouds in batch order): a spatial walk order for gather operators, not part of any result. */
size_t spr_cell_order_workspace_bytes(int n);
int spr_cell_order(const float* xyz, const int* cu, int n, int nb, float cell, int* order, void* ws, size_t ws_bytes,
                   void* stream);
int spr_layernorm_range_count(int m);
int spr_layernorm_r(const float* x, int m, int c, const float* gamma, const float* beta,
                    float eps, const float* pos, float* out_norm, float* out_pos,
                    float* range_norm, float* range_pos, void* stream);
/* w_prep (or NULL): the weight-side inputs of the fused in-projection -- spr_range_parts() partial
 * maxima of |w_in| followed by its 3 d row L1 norms -- from spr_attn_inproj_prepare, measured once
 * per weight version. */
int spr_attn_inproj_prepare(const float* w_in, int d, float* out, void* stream);
int spr_attn_inproj_varlen_fwd_r(const float* x_qk, const float* x_v, int t, const float* w_in,
                                 const float* b_in, const int* cu, const int* kv_seg, int nseg,
                                 int max_len_host, int nhead, int head_dim, float scale,
                                 float* out, int o_stride, const float* xqk_range,
                                 int xqk_range_n, const float* xv_range, int xv_range_n,
                                 float* out_range, const float* w_prep, void* ws, size_t ws_bytes, void* stream);

/* Arithmetic of spr_linear (and of the correlation GEMMs inside the matching
 * head):
 *   1 (default) = split-fp16 MFMA: each operand tensor is scaled by a power of
 *     two derived from its measured max |x| (no overflow possible, whatever the
 *     magnitudes), then carried as fp16 hi + fp16 lo; three MFMAs per product,
 *     fp32 accumulation.  Elements within 2^-18 of their tensor's maximum keep 22
 *     significand bits (~2^-22 relative per product); smaller ones keep an
 *     absolute error of 2^-39 max|x|.  ~5x the exact-f32 MFMA rate.
 *   0 = exact f32 MFMA (a k-ordered fmaf chain). */
int spr_set_gemm_mode(int mode);

/* ---- LayerNorm (+ positional embedding add) --------------------------------
 * Replaces norm1/2/3 + with_pos_embed (transformers.py:121,:196-197,:212-214,
 * :234) and the final encoder norm (:46-48).
 *   out_norm = LN(x) (may be NULL); out_pos = LN(x) + pos (NULL if pos NULL)
 */
int spr_layernorm(const float* x, int m, int c, const float* gamma,
                  const float* beta, float eps, const float* pos,
                  float* out_norm, float* out_pos, void* stream);

/* ---- a7: sine positional embedding -----------------------------------------
 * Replaces PositionEmbeddingCoordsSine.forward (transformer/
 * position_embedding.py:29-50).  xyz [n,3] -> out [n,d_model].
 */
int spr_posemb_sine(const float* xyz, int n, int d_model, float scale,
                    float temperature, float* out, void* stream);

/* ---- a9: varlen multi-head attention core -----------------------------------
 * Replaces the scaled-dot-product core of the four nn.MultiheadAttention
 * calls per layer (transformers.py:198-227) on packed tokens: segment s
 * attends from its own queries to the keys/values of segment kv_seg[s]
 * (self: kv_seg[s] = s; cross: the partner cloud).  Key padding masks of the
 * reference become segment bounds.
 *   q,k,v: [T, *] f32 with row strides (in floats); head h uses columns
 *   [h*head_dim, (h+1)*head_dim); head_dim must be 32.
 *   cu [nseg+1]; kv_seg [nseg]; out [T, nhead*head_dim] (row stride o_stride)
 *   t = cu[nseg] (total tokens, host copy); max_len_host: upper bound of any
 *   segment length (sizes the grid).  ws: spr_attn_workspace_bytes(t, nseg,
 *   nhead, head_dim) bytes of device scratch (the split-fp16 operand planes;
 *   may be NULL in mode 0).
 */
size_t spr_attn_workspace_bytes(int t, int nseg, int nhead, int head_dim);
int spr_attn_varlen_fwd(const float* q, int q_stride, const float* k,
                        int k_stride, const float* v, int v_stride,
                        const int* cu, const int* kv_seg, int t, int nseg,
                        int max_len_host, int nhead, int head_dim, float scale,
                        float* out, int o_stride, void* ws, size_t ws_bytes,
                        void* stream);
/* The same, additionally handing out lse [t, nhead] = log2 sum_j 2^(log2(e) scale q_i.k_j) per query and head
 * for spr_attn_varlen_bwd_lse (training: what torch's SDPA backward keeps as `logsumexp`).  *lse_written = 0 when
 * the configured core does not produce it (exact-f32 mode, the eager-softmax experiment): lse is then untouched. */
int spr_attn_varlen_fwd_lse(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                            int v_stride, const int* cu, const int* kv_seg, int t, int nseg, int max_len_host,
                            int nhead, int head_dim, float scale, float* out, int o_stride, float* lse,
                            int* lse_written, void* ws, size_t ws_bytes, void* stream);

/* In-projection + attention core in one call: replaces
 * F.multi_head_attention_forward's packed in-projection (q, k from x_qk,
 * v from x_v; in_proj_weight [3d, d], in_proj_bias [3d]) followed by the core
 * above -- nn.MultiheadAttention as called at transformers.py:198-227 minus the
 * out-projection.  In split-fp16 mode the projection GEMM writes the attention
 * operand planes straight from its accumulators (no fp32 [t, 3d] round trip);
 * in exact mode it equals spr_linear + spr_attn_varlen_fwd.  d_model = 256.
 *   x_qk, x_v [t, d] contiguous (may be the same pointer); out [t, d].
 */
size_t spr_attn_inproj_workspace_bytes(int t, int nseg, int nhead, int head_dim);
int spr_attn_inproj_varlen_fwd(const float* x_qk, const float* x_v, int t,
                               const float* w_in, const float* b_in,
                               const int* cu, const int* kv_seg, int nseg,
                               int max_len_host, int nhead, int head_dim,
                               float scale, float* out, int o_stride, void* ws,
                               size_t ws_bytes, void* stream);

/* ---- a9, fused: the whole cross-encoder stack of an inference forward -----------------------
 * Replaces TransformerCrossEncoder.forward (models/transformer/transformers.py:45-80) over
 * TransformerCrossEncoderLayer.forward_pre (:184-245) for the configuration every shipped experiment
 * uses (pre_norm, sa_val_has_pos_emb, ca_val_has_pos_emb, ReLU, dropout 0, d_model 256 = 8 x 32):
 * per layer two attention cores and two fused row chains (csrc/xenc.hip) -- out-projection +
 * residual, LayerNorm (+ positional embedding), feed-forward block, next in-projection -- that keep
 * every intermediate on the chip and write the attention operand planes directly.
 * spr_xenc_prepare: once per weight version.  layer_ptrs_host = SPR_XENC_PTRS_PER_LAYER device
 *   pointers per layer in the order self_attn.{in_proj_weight, in_proj_bias, out_proj.weight,
 *   out_proj.bias}, multihead_attn.{same four}, linear1.{weight, bias}, linear2.{weight, bias},
 *   norm1.{weight, bias}, norm2.{..}, norm3.{..}; eps_host = 3 floats per layer (norm1..3);
 *   final_g / final_b = the stack's final LayerNorm or NULL; pos_bound = max |pos| (1 for the sine
 *   embedding).  Lays the weights out as split-fp16 MFMA fragments in `prepared`
 *   (spr_xenc_prepared_bytes, device) and fills the host-side plan (spr_xenc_plan_bytes, opaque).
 *   Synchronises the stream (one device->host read of the parameter statistics).  The plan refers to
 *   `prepared` and to the bias / LayerNorm parameters themselves: keep all of them alive and unchanged.
 * spr_xenc_forward: x, pos, out [t, 256]; cu [nseg + 1]; kv_self / kv_cross [nseg] = key segment of
 *   every query segment.  Split-fp16 arithmetic only (gemm mode 1, attention mode 1 or 2).
 */
#define SPR_XENC_PTRS_PER_LAYER 18
size_t spr_xenc_prepared_bytes(int n_layers, int d_ff);
size_t spr_xenc_plan_bytes(void);
int spr_xenc_prepare(const void* const* layer_ptrs_host, const float* eps_host, int n_layers,
                     int d_model, int nhead, int d_ff, const float* final_g, const float* final_b,
                     float final_eps, float pos_bound, void* prepared, size_t prepared_bytes,
                     void* plan_host, size_t plan_bytes, void* stream);
size_t spr_xenc_workspace_bytes(int t, int nseg);
int spr_xenc_forward(const void* plan_host, const float* x, const float* pos, const int* cu,
                     const int* kv_self, const int* kv_cross, int t, int nseg, int max_len_host,
                     float* out, void* ws, size_t ws_bytes, void* stream);

/* Backward of spr_attn_varlen_fwd (the graph torch autograd builds for the attention core of
 * nn.MultiheadAttention, transformers.py:198-227), flash style: the Lq x Lk matrices are recomputed
 * tile by tile and never written.  out = the forward's output, dout = its gradient; kv_seg must be a
 * permutation of the segments, q_seg its inverse.  Exact f32 MFMA, fixed summation order (bitwise
 * reproducible).  dq, dk, dv [t, nhead * 32] contiguous, fully written.  head_dim = 32.
 */
size_t spr_attn_bwd_workspace_bytes(int t, int nhead);
/* the same with room for the pre-split operand planes of the split-fp16 form (faster; optional) */
size_t spr_attn_bwd_workspace_bytes2(int t, int nseg, int nhead);
int spr_attn_varlen_bwd(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                        int v_stride, const float* out, int o_stride, const float* dout,
                        int do_stride, const int* cu, const int* kv_seg, const int* q_seg, int t,
                        int nseg, int max_len_host, int nhead, int head_dim, float scale,
                        float* dq, float* dk, float* dv, void* ws, size_t ws_bytes, void* stream);
/* lse: what spr_attn_varlen_fwd_lse handed out for the same q, k (the backward then skips its own pass over the
 * keys); NULL = spr_attn_varlen_bwd */
int spr_attn_varlen_bwd_lse(const float* q, int q_stride, const float* k, int k_stride, const float* v,
                            int v_stride, const float* out, int o_stride, const float* dout, int do_stride,
                            const float* lse, const int* cu, const int* kv_seg, const int* q_seg, int t, int nseg,
                            int max_len_host, int nhead, int head_dim, float scale, float* dq, float* dk,
                            float* dv, void* ws, size_t ws_bytes, void* stream);

/* Arithmetic of the attention core:
 *   1 = split-fp16 MFMA (Q, K, V and the probabilities carried as fp16
 *     hi + lo; Q/K balanced and V scaled by powers of two derived from measured
 *     or derived bounds, so no magnitude overflows fp16; fp32 accumulation and
 *     softmax; fp32-level accuracy);
 *   0 = exact f32 MFMA;
 *   2 = single-pass fp16 MFMA (hi planes only: 11-bit operands, fp32 softmax and
 *     accumulators; 1/3 of the matrix-core work -- the throughput mode BASELINE
 *     configs[4] names; ~1e-3 relative on the attention output);
 *   3 = split-fp16 scores, the probabilities as ONE fp16 plane (rounded toward
 *     zero) whose rounded values also form the row sum: exact convex combination
 *     of the value rows with weights perturbed by < 2^-10 each (5e-6 of the
 *     output scale on flat rows of ~2 000 keys, up to ~3e-4 on rows carried by a
 *     few keys); 20 instead of 24 MFMAs per 64-key tile;
 *   4 (default since round 5) = as 1, the lo plane of the probabilities only on
 *     the 32-key blocks that hold a weight of at least 2^-5 of the running row
 *     sum: every key that carries a row is exact as in mode 1, the many small
 *     weights travel in one fp16 plane rounded to nearest (their rounded values
 *     also form the row sum).  Error of the small weights: 2^-12 / sqrt(3 n) of
 *     the value spread for n comparable keys -- <= 3e-5 of the output scale on
 *     every tested distribution, 3e-6 on the bench's rows of ~2 000 keys (mode
 *     1: 1e-6), and equal to mode 1 on peaked rows (DESIGN.md section 4,
 *     profiles/r05_attn_mode_accuracy.txt).  22-23 instead of 24 MFMAs per
 *     64-key tile and a third fewer conversion instructions. */
int spr_set_attn_mode(int mode);

/* ---- a11: dual-softmax matching ---------------------------------------------
 * Replaces the correlation / dual softmax / arg-max block of
 * RegTR.softmax_correlation (qk_regtr_full.py:453-479, :565-588) for all pairs
 * at once.  Pair b: src tokens = segment b, tgt tokens = segment b + npairs.
 *   feat [T, d]; cu [2*npairs+1]
 *   If N_b > M_b : one match per tgt token  (val,ind over src, len M_b)
 *   else         : one match per src token  (val,ind over tgt, len N_b)
 *   match_val / match_ind are written at the packed position of the token
 *   that owns the match (tgt token when N>M, else src token); ind is LOCAL to
 *   the partner cloud.  corr_ws: caller scratch for the correlation matrices,
 *   size from spr_match_workspace_bytes (host needs the seg lengths).
 */
size_t spr_match_workspace_bytes(const int* cu_host, int npairs);
int spr_match_dualsoftmax(const float* feat, int d, const int* cu,
                          const int* cu_host, int npairs, float* match_val,
                          int* match_ind, void* ws, size_t ws_bytes,
                          void* stream);

/* Variant that also returns the runner-up value of every match (match_val2, may be NULL) --
 * what RegTR.ratio_test needs (Lowe ratio, qk_regtr_full.py:370-384; cfg.use_ratio_test). */
int spr_match_dualsoftmax2(const float* feat, int d, const int* cu,
                           const int* cu_host, int npairs, float* match_val,
                           float* match_val2, int* match_ind, void* ws, size_t ws_bytes,
                           void* stream);

/* ---- pose-hypothesis residuals (config-off refinements) -------
 * spr_pose_residuals: res[i] = || b_i - T_s a_i || with one pose per set s of pair_cu
 *   (RegTR.recompute_weights / local_global_registration, qk_regtr_full.py:386-398);
 *   total_host = pair_cu[npairs] (host copy, sizes the grid).
 * spr_pose_scores: score[h] = mean_i || b_i - T_h a_i || for nh hypotheses over ONE point
 *   set (the RANSAC loop of qk_regtr_full.py:400-421, all hypotheses in one launch).
 */
int spr_pose_residuals(const float* pose, const float* a, const float* b, const int* pair_cu,
                       int npairs, int total_host, float* res, void* stream);
int spr_pose_scores(const float* poses, int nh, const float* a, const float* b, int n,
                    float* score, void* stream);

/* ---- a12: weighted Procrustes / Kabsch --------------------------------------
 * Replaces compute_rigid_transform(a, b, weights) (utils/se3_torch.py:109-163)
 * batched over pairs; 3x3 SVD by one-sided Jacobi in registers.
 *   a,b [T,3] packed by pair_cu [npairs+1]; w [T]; out [npairs,3,4].
 */
int spr_weighted_procrustes(const float* a, const float* b, const float* w,
                            const int* pair_cu, int npairs, float* out_pose,
                            void* stream);

/* ---- a13: Sinkhorn (slack) soft correspondences -----------------------------
 * Replaces the Sinkhorn branch of softmax_correlation
 * (qk_regtr_full.py:525-536 / :635-647) + sinkhorn() and the weighted target
 * of compute_rigid_transform_with_sinkhorn (utils/se3_torch.py:166-239):
 *   score = clamp(F_s F_t^T / sqrt(d), min 0)
 *   affinity = -(score - softplus(alpha)) / (exp(beta) + 0.02)
 *   n_iters x (row normalise incl. slack col; col normalise incl. slack row)
 *   P = exp(.), w_i = sum_j P_ij, t_hat_i = sum_j P_ij tgt_j / (w_i + 1e-6)
 * alpha, beta: DEVICE pointers to the model's learnable scalars
 * (qk_regtr_full.py:77-78) -- read by the kernel, no host round trip.
 * Outputs per src token: w [Tsrc], t_hat [Tsrc,3] (packed like the src
 * segments).  Feed (src_xyz, t_hat, w) to spr_weighted_procrustes.
 */
size_t spr_sinkhorn_workspace_bytes(const int* cu_host, int npairs);
int spr_sinkhorn_correspondences(const float* feat, int d, const float* xyz,
                                 const int* cu, const int* cu_host, int npairs,
                                 const float* alpha, const float* beta, int n_iters,
                                 int slack, float* out_w, float* out_that,
                                 void* ws, size_t ws_bytes, void* stream);

/* spr_match_dualsoftmax2 + spr_sinkhorn_correspondences of the same features in one call -- RegTR's inference
 * forward runs them back to back on the conditioned features (qk_regtr_full.py:453-479 then :525-536).  The scaled
 * correlation matrices are computed and stored once; the Sinkhorn passes evaluate the affinity as they read them.
 * Outputs bit for bit those of the two separate calls; match_val2 may be NULL; workspace:
 * spr_match_workspace_bytes. */
int spr_match_sinkhorn(const float* feat, int d, const float* xyz, const int* cu, const int* cu_host,
                       int npairs, const float* alpha, const float* beta, int n_iters, float* match_val,
                       float* match_val2, int* match_ind, float* out_w, float* out_that, void* ws,
                       size_t ws_bytes, void* stream);

/* ---- small helpers ----------------------------------------------------------*/
/* out[i] = x[idx[i]] rows of width c (idx i32, rows >= n_src read zeros). */
int spr_gather_rows(const float* x, int n_src, int c, const int* idx, int n,
                    float* out, void* stream);

/* ---- losses (forward) of RegTR.compute_loss -- SURVEY 8f row 1 --------------
 * models/qk_regtr_full.py:313-368.  Deterministic reductions (fixed partition,
 * float64 accumulation).  Scalars are written to device memory (out[0]).
 *
 * spr_overlap_pool: one level of compute_overlaps (backbone_kpconv/kpconv.py:
 *   552-578): out[q] = clamp(mean over the valid pool entries of ov_prev, 0, 1).
 * spr_bce_logits_mean: nn.BCEWithLogitsLoss(reduction='mean') (:90, :329).
 * spr_infonce_pair: InfoNCELossFull.compute_infonce (models/losses/
 *   feature_loss.py:268-296) for ONE pair; anchor_xyz are the source keypoints,
 *   transformed by pose_gt [3,4] inside (se3_transform, :341-345).
 * spr_transform_l1_pair: mean |T_gt x - T_pred x| over one pair's keypoints
 *   (:349-355).   spr_sum_scaled: out[0] = scale * sum(values[0..n)).
 * ws: spr_loss_workspace_bytes(max anchors, max positives, d) covers all four.
 */
size_t spr_loss_workspace_bytes(int n_max, int m_max, int d);
int spr_overlap_pool(const float* ov_prev, int ns_prev, const int* pool,
                     int pool_stride, int w, int nq, float* out, void* stream);
int spr_bce_logits_mean(const float* x, const float* y, int n, float* out,
                        void* ws, size_t ws_bytes, void* stream);
int spr_infonce_pair(const float* anchor_feat, int n, const float* positive_feat,
                     int m, int d, const float* anchor_xyz, const float* pose_gt,
                     const float* positive_xyz, const float* W, float r_p,
                     float r_n, float* out, void* ws, size_t ws_bytes,
                     void* stream);
int spr_transform_l1_pair(const float* pose_gt, const float* pose_pred,
                          const float* xyz, int n, float* out, void* ws,
                          size_t ws_bytes, void* stream);
int spr_sum_scaled(const float* values, int n, float scale, float* out,
                   void* stream);

/* ==== backward ===============================================
 * The reference trains through the path with torch autograd
 * (models/generic_reg_model.py:82-84 training_step -> trainer.py:107-124
 * backward / clip / step).  These entry points are the explicit gradients of the
 * operators above; autograd.py wires them into torch.autograd.Function objects.
 *
 * spr_bgemm: batched strided exact-f32 GEMM, the matrix-product workhorse
 *   C_b(i,j) = alpha * sum_k A_b(i,k) B_b(k,j) + beta * C_b(i,j),
 *   X_b(p,q) = X[x_off_b + p * sx_p + q * sx_q]; desc: device array of nbatch records
 *   {int64 a_off, b_off, c_off; int32 m, n, k, pad} (element offsets / sizes per batch).
 *   Gives dX = dY W, dW = dY^T X (nn.Linear), the KPConv weight / feature gradients
 *   and every product of the attention backward.
 * spr_reduce_parts: out[j] (+)= scale * sum_p parts[p][j] in fixed order (deterministic split-K).
 * spr_act_bwd: dy * act'(y) for SPR_ACT_RELU / SPR_ACT_SIGMOID.   spr_colsum: bias gradients.
 * spr_layernorm_bwd / spr_instnorm_bwd / spr_maxpool_bwd / spr_scatter_rows_add: gradients of
 *   spr_layernorm (both outputs), spr_instnorm (incl. fused add + LeakyReLU), spr_maxpool_gather,
 *   spr_gather_rows.
 * spr_kpconv_weighted_features: recomputes wf[n,p,c] = sum_k infl[n,p,k] x[idx[n,k],c]
 *   (kpconv_blocks.py:394) and the neighbour count (:409-411); spr_kpconv_bwd_dx scatters
 *   d wf back to d x.  Together with two spr_bgemm calls = KPConv backward.
 * spr_softmax_rows / spr_softmax_bwd_rows: row softmax of per-batch matrices (located by
 *   c_off, m rows, n columns of the same descriptor records) and its backward -- the
 *   non-GEMM steps of the attention backward.
 */
int spr_bgemm(const float* A, const float* B, float* C, const void* desc_dev, int nbatch,
              int max_m, int max_n, long sa_i, long sa_k, long sb_k, long sb_j, long sc_i,
              long sc_j, float alpha, float beta, void* stream);
/* out[nl, nr] = L[rows, nl]^T R[rows, nr] (row-major, contiguous) with float64 accumulation and a
 * fixed-order reduction: weight gradients that are long, nearly cancelling sums (the first KPConv's
 * dW = wf^T g over every point of the batch, kpconv_blocks.py:401-406 differentiated).  nl * nr <= 65536. */
/* dW = dY^T X in the forward's arithmetic (range-scaled split-fp16 MFMA, fp32 accumulation): per-batch partial
 * products parts[b][nl][nr] over `chunk` rows each, summed by spr_reduce_parts (fixed order).  L [rows, nl],
 * R [rows, nr] row-major; nl, nr multiples of 4, chunk a multiple of 16; ranges as in spr_linear_r. */
size_t spr_tn_product_split_workspace_bytes(void);
int spr_tn_product_split(const float* L, const float* R, long rows, int nl, int nr, int chunk,
                         const float* l_range, int l_range_n, const float* r_range, int r_range_n,
                         float* parts, void* ws, size_t ws_bytes, void* stream);
size_t spr_tn_product_f64_workspace_bytes(long rows, int nl, int nr);
int spr_tn_product_f64(const float* L, const float* R, long rows, int nl, int nr, float* out,
                       void* ws, size_t ws_bytes, void* stream);
int spr_reduce_parts(const float* parts, int nparts, long n, float scale, float* out,
                     int accumulate, void* stream);
int spr_act_bwd(const float* y, const float* dy, int act, long n, float* out, void* stream);
size_t spr_colsum_workspace_bytes(int n);
int spr_colsum(const float* x, long m, int n, float* out, void* ws, size_t ws_bytes, void* stream);
size_t spr_layernorm_bwd_workspace_bytes(int c);
int spr_layernorm_bwd(const float* x, int m, int c, const float* gamma, float eps,
                      const float* dy_norm, const float* dy_pos, float* dx, float* dgamma,
                      float* dbeta, void* ws, size_t ws_bytes, void* stream);
size_t spr_instnorm_bwd_workspace_bytes(int max_len_host, int nb, int c);
int spr_instnorm_bwd(const float* x, const float* out, const float* dout, const int* cu, int n,
                     int nb, int max_len_host, int c, float eps, int norm, float slope,
                     float* dx, float* dadd, void* ws, size_t ws_bytes, void* stream);
/* The three scatter-adds of the backward (max-pool, row gather, KPConv neighbour gather) sum in 64-bit
 * fixed point with integer atomics -- order independent, bitwise reproducible -- and WRITE dx in full
 * (no pre-zeroing); ws: spr_scatter_workspace_bytes(rows of dx, channels). */
size_t spr_scatter_workspace_bytes(long rows, int c);
int spr_maxpool_bwd(const float* x, int ns, int c, const int* idx, int nq, int idx_stride, int k,
                    const float* dy, float* dx, void* ws, size_t ws_bytes, void* stream);
int spr_scatter_rows_add(const float* dy, const int* idx, int n, int c, int n_src, float* dx,
                         void* ws, size_t ws_bytes, void* stream);
int spr_kpconv_weighted_features(const float* q_xyz, int nq, const float* s_xyz, int ns,
                                 const int* nbr, int nbr_stride, int kmax, const float* x, int cin,
                                 const float* kernel_points, int n_kp, float kp_extent,
                                 float* wf, float* cnt, void* stream);
int spr_kpconv_bwd_dx(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                      int nbr_stride, int kmax, int cin, const float* kernel_points, int n_kp,
                      float kp_extent, const float* dwf, float* dx, void* ws, size_t ws_bytes,
                      void* stream);
/* dwf_range: the max |dwf| partials published by the product that wrote dwf (spr_linear_r out_range); NULL = measured */
int spr_kpconv_bwd_dx_r(const float* q_xyz, int nq, const float* s_xyz, int ns, const int* nbr,
                        int nbr_stride, int kmax, int cin, const float* kernel_points, int n_kp,
                        float kp_extent, const float* dwf, const float* dwf_range, int dwf_range_n, float* dx,
                        void* ws, size_t ws_bytes, void* stream);
int spr_softmax_rows(float* mat, const void* desc_dev, int nbatch, int max_m, void* stream);
int spr_softmax_bwd_rows(const float* p, float* dp, const void* desc_dev, int nbatch, int max_m,
                         void* stream);

/* Loss and pose-head gradients.
 * spr_bce_logits_mean_bwd: dx = gout[0] (sigmoid(x) - y) / n.
 * spr_infonce_pair_dlogits: recomputes the pair's logits and writes the un-normalised
 *   d logits [n, m] + row mask [n] (divide by sum(mask)); also returns W_sym [d, d] and
 *   t = A W_sym [n, d] for the caller's GEMM chain; spr_wsym_bwd maps d W_sym to d W.
 * spr_transform_l1_pair_bwd: d pose_pred [3,4] of spr_transform_l1_pair.
 * spr_weighted_procrustes_bwd: gradient of the Kabsch solve w.r.t. b, w (and a if da != NULL)
 *   by implicit differentiation of the SVD-based rotation (se3_torch.py:141-162; the reference
 *   uses torch.svd's autograd).  da / db / dw packed like a / b / w.
 * spr_sinkhorn_bwd: gradient of spr_sinkhorn_correspondences w.r.t. feat, alpha, beta
 *   (unrolled over the n_iters slack-Sinkhorn iterations; se3_torch.py:166-239,
 *   qk_regtr_full.py:525-536).
 */
int spr_bce_logits_mean_bwd(const float* x, const float* y, int n, const float* gout, float* dx,
                            void* stream);
int spr_infonce_pair_dlogits(const float* anchor_feat, int n, const float* positive_feat, int m,
                             int d, const float* anchor_xyz, const float* pose_gt,
                             const float* positive_xyz, const float* W, float r_p, float r_n,
                             float* dlogits, float* row_mask, float* wsym_out, float* t_out,
                             void* ws, size_t ws_bytes, void* stream);
int spr_wsym_bwd(const float* dwsym, int d, float* dW, void* stream);
int spr_transform_l1_pair_bwd(const float* pose_gt, const float* pose_pred, const float* xyz, int n,
                              const float* gout, float* dpose_pred, void* stream);
int spr_weighted_procrustes_bwd(const float* a, const float* b, const float* w,
                                const int* pair_cu, int npairs, const float* dpose, float* da,
                                float* db, float* dw, void* stream);
size_t spr_sinkhorn_bwd_workspace_bytes(const int* cu_host, int npairs, int n_iters);
int spr_sinkhorn_bwd(const float* feat, int d, const float* xyz, const int* cu,
                     const int* cu_host, int npairs, const float* alpha, const float* beta,
                     int n_iters, const float* dw, const float* dthat, float* dfeat,
                     float* dalpha, float* dbeta, void* ws, size_t ws_bytes, void* stream);

/* Per-launch timing of the fused KPConv kernel and of the attention core kernel
 * with HIP events recorded on the launch stream (used by bench.py for the
 * roofline figures; off by default).
 * spr_prof_enable(1) clears the log and starts recording, spr_prof_enable(0)
 * clears and stops.  spr_prof_read synchronises on the recorded events, CONSUMES them and
 * returns up to max_records entries (all arrays HOST memory):
 *   codes[i] = cin * 100000 + cout (KPConv) or -1 (attention core), nqs[i] =
 *   query / token count, ms[i] = duration. */
int spr_prof_enable(int on);
int spr_prof_read(int max_records, int* codes_host, int* nqs_host, float* ms_host);

/* MFMA / wave layout self test (writes 0 to *status_host on success). */
int spr_selftest(int* status_host);

#ifdef __cplusplus
}
#endif
#endif /* SPR_H_ */
