/*
 * include/efg_hip.h -- C ABI of libefg_hip.so, the MI355X (gfx950) implementation of the EFG
 * Voxel-DETR / ConQueR hot path.  This is the drop-in boundary: the entry points are what a
 * replacement of the reference's pybind11 module `efg._C` (efg/operators/src/vision.cpp:70-122)
 * and of the third-party `spconv` ops binds to.  INTEGRATION.md shows the reference-side stubs.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in `_host`;
 *  - tensors are dense row-major ("contiguous"), fp32 / int32 / int64 as named;
 *  - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream); every kernel
 *    of a call is enqueued on it and nothing synchronises the device unless stated;
 *  - temporaries live in a caller-owned workspace (`ws`, `ws_bytes`), sized by the matching
 *    `*_workspace_bytes` query, so the caller's allocator (PyTorch's caching allocator) owns
 *    all memory and calls are re-entrant (no global mutable state);
 *  - return value: 0 = OK, negative = error (EFG_E_*); efg_last_error() returns a thread-local
 *    message for the last failure on the calling thread.
 */
#ifndef EFG_HIP_H
#define EFG_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EFG_OK 0
#define EFG_E_INVALID (-1)   /* bad argument (shape, size, unsupported configuration) */
#define EFG_E_WORKSPACE (-2) /* workspace too small */
#define EFG_E_HIP (-3)       /* a HIP runtime call / kernel launch failed */

const char* efg_last_error(void);
/* "efg_hip <version> gfx950" */
const char* efg_version(void);

/*
 * A non-blocking stream of the caller's own, on the calling thread's current device, for hipGraph captures.  On ROCm 7
 * a capture that is invalidated (an illegal call while capturing) leaves its stream invalidated for good --
 * hipStreamEndCapture reports the error without ending the capture -- so a capture must not run on a stream that will be
 * handed to anything else later (PyTorch's pool streams are).  Create one, capture on it, keep it for the next capture;
 * destroy it if its capture failed.  (No counterpart in the reference: CUDA ends an invalidated capture.)
 */
int efg_capture_stream_create(void** stream_out);
int efg_capture_stream_destroy(void* stream);

/* ------------------------------------------------------------------------------------------
 * Voxelization.  Replaces efg::dynamic_voxelize / efg::hard_voxelize
 * (efg/operators/src/voxelize/voxelization.h:51-83; CPU semantics voxelization_cpu.cpp:7-99).
 * Two implementations behind efg_hard_voxelize_f32 (csrc/voxelize_bins.hip, the default: points binned into
 * BEV supercells, per-bin tables in LDS; csrc/voxelize_hash.hip: one global hash table, taken when the
 * environment says EFG_VOX_IMPL=hash or when the grid has far more supercells than points -- that path needs
 * grid volume x batch < 2^32 - 1 cells).
 * ---------------------------------------------------------------------------------------- */

/* coors[n,3] int32 (z,y,x); (-1,-1,-1) for points outside coors_range (CPU encoding,
 * voxelization_cpu.cpp:32-37).  NaN coordinates count as outside. */
int efg_dynamic_voxelize_f32(const float* points, int64_t n, int f, const float* voxel_size_host,
                             const float* coors_range_host, int32_t* coors, void* stream);

/* Bytes of scratch efg_hard_voxelize_f32 needs for this call shape (0 for invalid arguments).  The geometry
 * matters: the binned path keeps per-supercell counters. */
size_t efg_hard_voxelize_workspace_bytes(int64_t n_total, int batch, int f, int max_points, int max_voxels,
                                         const float* voxel_size_host, const float* coors_range_host);

/*
 * Batched hard voxelization of `batch` scenes in one call.  Scene b owns the point rows
 * [point_offsets_host[b], point_offsets_host[b+1]).  Per scene the result is bit-identical to
 * the reference loop (voxelization_cpu.cpp:43-99): voxels in first-occurrence order, at most
 * max_points points per voxel in point order, processing stops at the first point that would
 * open voxel number max_voxels.
 *
 * Outputs are CONCATENATED over scenes (the layout efg/data/datasets/waymo/waymo.py:143-183
 * `collate` produces): rows [base_b, base_b + M_b) belong to scene b, base_b = sum_{b'<b} M_b'.
 *   voxels   f32 [batch*max_voxels, max_points, f]  rows < sum M fully written (zero padded)
 *   coors    i32 [batch*max_voxels, coors_cols]     coors_cols = 3: (z,y,x); 4: (b,z,y,x)
 *   npv      i32 [batch*max_voxels]
 *   voxel_num i32 [batch]                           M_b (device; read it after the stream syncs)
 *   mean     f32 [batch*max_voxels, f] or NULL      fused VoxelMeanFeatureExtractor
 *                                                   (efg/modeling/readers/voxel_reader.py:14-19)
 * With batch == 1 and coors_cols == 3 this is exactly efg::hard_voxelize.
 *
 * Besides the caller's workspace the binned implementation owns a little device memory of its own, per (device, stream):
 * its per-XCD counters and look-back records (a few MB: 8 words per BEV supercell + 192 KB), allocated with hipMalloc +
 * hipMemset on first use, left ZERO by every call for the next one (so a call has no clear launch) and kept for the life
 * of the process.  Calls on ONE stream must be issued by one thread at a time.  A call issued while its stream is being
 * captured into a HIP graph takes those words from the workspace and clears them with a kernel instead (a replay must
 * not depend on what eager calls left behind).
 */
int efg_hard_voxelize_f32(const float* points, const int64_t* point_offsets_host, int batch, int f,
                          const float* voxel_size_host, const float* coors_range_host, int max_points,
                          int max_voxels, float* voxels, int32_t* coors, int coors_cols, int32_t* npv,
                          int32_t* voxel_num, float* mean, void* ws, size_t ws_bytes, void* stream);

/* Development aid (scripts/vox_timeline.py): with a device buffer of the returned number of uint64 words set, thread
 * 0 of every workgroup of the binned voxelizer's kernels stores the 100 MHz wall clock at its phase markers
 * ([kernel 0..3][marker 0..7][workgroup < 32768]); NULL (the default) switches it off.  Returns the word count. */
size_t efg_hard_voxelize_debug_timeline(void* device_buf_u64);

/* ------------------------------------------------------------------------------------------
 * Dynamic scatter.  Replaces efg::dynamic_point_to_voxel_forward / _backward
 * (voxelization.h:96-128; scatter_points_cuda.cu:209-352).  reduce: 0 sum, 1 mean, 2 max.
 * Two-phase forward because M (number of distinct voxels) sizes the outputs:
 *   efg_scatter_index   builds point2voxel[n] (voxel id = rank of the linearised coordinate,
 *                       -1 for rows with a negative coordinate) and writes M to *m_dev;
 *   efg_scatter_reduce  fills voxel_feats[M,c], voxel_coors[M,ndim], count[M].
 * ---------------------------------------------------------------------------------------- */
size_t efg_scatter_workspace_bytes(int64_t n, int ndim, const int32_t* dims_host);
/* dims_host[ndim] = per-column max + 1 (the caller computes coors.max(0)+1 as the reference
 * does at scatter_points_cuda.cu:220). */
int efg_scatter_index(const int32_t* coors, int64_t n, int ndim, const int32_t* dims_host,
                      int32_t* point2voxel, int32_t* m_dev, void* ws, size_t ws_bytes, void* stream);
/* Sum / mean are accumulated as exact 64-bit fixed-point integers (scale from the call's largest |x| and n): the
 * result is the correctly rounded true sum and bit-identical run to run -- the reference's float atomics
 * (scatter_points_cuda.cu:101-133) depend on arrival order.  ws: efg_scatter_reduce_workspace_bytes(m, c). */
size_t efg_scatter_reduce_workspace_bytes(int64_t m, int c);
int efg_scatter_reduce_f32(const float* feats, const int32_t* coors, const int32_t* point2voxel, int64_t n,
                           int c, int ndim, int reduce, int64_t m, float* voxel_feats, int32_t* voxel_coors,
                           int32_t* count, void* ws, size_t ws_bytes, void* stream);
/* grad_feats[n,c] is fully written.  ws: m*c int32 (max only). */
int efg_scatter_backward_f32(float* grad_feats, const float* grad_voxel_feats, const float* feats,
                             const float* voxel_feats, const int32_t* point2voxel, const int32_t* count,
                             int64_t n, int64_t m, int c, int reduce, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sparse convolution.  Replaces the spconv ops reached from
 * efg/modeling/backbones/sparse_net.py:79-98,120-165,273-309,485-545 (SparseConvTensor,
 * SubMConv3d, SparseConv3d, .dense()).  indices are int32 (b,z,y,x) rows.
 *
 * Geometry is held in an "index" = bitmap of active cells + exclusive popcount prefix, packed
 * as uint2 {bits, prefix} per 32 cells of the linearised grid ((b*D+z)*H+y)*W+x: the rank of a
 * set bit is the row of that site in canonical (ascending linear index) order.
 * ---------------------------------------------------------------------------------------- */
size_t efg_spconv_index_bytes(int batch, const int* shape_host);       /* bytes of one index    */
size_t efg_spconv_index_workspace_bytes(int batch, const int* shape_host);

/* Build the index of a given set of sites.  perm (i32[m]) receives canonical-rank -> row of
 * `indices` (identity when the rows are already in canonical order). */
int efg_spconv_index_from_indices(const int32_t* indices, int64_t m, int batch, const int* shape_host,
                                  void* index, int32_t* perm, void* ws, size_t ws_bytes, void* stream);

/* Regular SparseConv3d geometry, step 1: mark every output site touched by an input site and COUNT them.
 * out_shape_host receives (in + 2p - k)/s + 1 per axis.  *m_out_dev receives the site count as soon as the one
 * marking kernel is done; the caller reads it back and allocates out_indices[m_out,4].  The index only becomes
 * rankable (usable by _emit / _build_nbr) after efg_spconv_index_rank -- split off so that the count read-back
 * does not wait for the three ranking kernels. */
int efg_spconv_index_downsample(const int32_t* in_indices, int64_t m_in, int batch, const int* in_shape_host,
                                const int* ksize_host, const int* stride_host, const int* pad_host,
                                void* out_index, int* out_shape_host, int32_t* m_out_dev, void* ws,
                                size_t ws_bytes, void* stream);
/* step 2: popcount prefix over the bitmap (m_dev may be NULL; otherwise it receives the site count again). */
int efg_spconv_index_rank(void* index, int batch, const int* shape_host, int32_t* m_dev, void* ws, size_t ws_bytes,
                          void* stream);
int efg_spconv_index_emit(const void* index, int batch, const int* shape_host, int32_t* out_indices,
                          void* stream);

/* nbr[kvol][m_out] int32: row (in the INPUT tensor's row order) of the active input at
 * out*stride - pad + k, or -1.  in_perm may be NULL when input rows are in canonical order. */
int efg_spconv_build_nbr(const void* in_index, const int32_t* in_perm, int batch, const int* in_shape_host,
                         const int32_t* out_indices, int64_t m_out, const int* ksize_host,
                         const int* stride_host, const int* pad_host, int32_t* nbr, void* stream);
/* Reverse table for dgrad: rnbr[kvol][m_in] = output row o with nbr[k][o] == i, or -1. */
int efg_spconv_build_rnbr(const int32_t* nbr, int64_t m_out, int kvol, int64_t m_in, int32_t* rnbr,
                          void* stream);

/* Weights arrive in the spconv 2.x parameter layout f32 [cout][kvol][cin] ([Cout,kd,kh,kw,Cin]) and
 * are re-packed once per call site into MFMA B-operand order (one 16-byte load per lane feeds four
 * v_mfma_f32_16x16x4_f32):  for_dgrad = 0: packed[k][cin/16][cout_pad][16]  (reduce over cin)
 *                           for_dgrad = 1: packed[k][cout/16][cin_pad][16]  (reduce over cout).
 * Within a 16-channel group the lane kk's fragment holds channels {kk, kk+4, kk+8, kk+12}; for_dgrad | 2 ("natural
 * order": {4kk .. 4kk+3}) is the layout of the 16-byte-gather tile kernel that was retired in round 6 -- the packer still
 * writes it, no convolution entry point takes it.
 * for_dgrad | 4: the split-precision (bf16 x 3) layout of the tile kernel's A/B arm (flip_offsets | 4 there; reduction
 * width a multiple of 32, output width a multiple of 64; same size): per (offset, 32-channel step, n-tile of 16) the 64
 * lanes' 8 bf16 of W_hi, then of W_lo.  efg_spconv_tile_bf16x3_ok says whether a layer is covered. */
size_t efg_spconv_packed_weight_bytes(int cout, int kvol, int cin, int for_dgrad);
int efg_spconv_pack_weight_f32(const float* weight, int cout, int kvol, int cin, int for_dgrad, float* packed,
                               void* stream);
/* The same for n weights in ONE launch.  items_dev: device array of n 32-byte records
 *   { const float* weight; float* packed; int32 cout, kvol, cin, for_dgrad; }
 * (the model's layers change together at the optimizer step: their packed copies are refreshed together). */
int efg_spconv_pack_weights_multi(const void* items_dev, int n, void* stream);

/* out[o][:] = bias + sum_k W[:,k,:] . in[nbr[k][o]][:]   (bias may be NULL; packed: for_dgrad = 0) */
int efg_spconv_forward_f32(const float* in_feat, int64_t m_in, int cin, const float* packed_weight,
                           const float* bias, int cout, int kvol, const int32_t* nbr, int64_t m_out,
                           float* out_feat, void* stream);
/* grad_in[i][:] = sum_k W[:,k,:]^T . grad_out[rnbr[k][i]][:]   (packed: for_dgrad = 1).
 * row_order: NULL, or a permutation of the m_in rows (efg_spconv_parity_order): workgroup tiles take their 16 rows
 * in that order; the result does not depend on it. */
int efg_spconv_dgrad_f32(const float* grad_out, int64_t m_out, int cout, const float* packed_weight, int cin,
                         int kvol, const int32_t* rnbr, int64_t m_in, const int32_t* row_order, float* grad_in,
                         void* stream);
/* ---- mask-sorted row tiles (csrc/spconv_tiles.hip) ------------------------------------------------------------
 * A tile plan of a neighbour table `nbr` [kvol][m]: the rows re-ordered inside chunks of 1024 consecutive rows by
 * their kernel-offset mask, cut into 16-row tiles, with the neighbour columns in tile-major order and per-offset
 * validity masks.  Geometry only (no features, no weights): build once per table, reuse for every convolution that
 * walks it.  kvol <= 31.  Replaces spconv's "mask sort" of the implicit-GEMM rulebook (third-party, not in the
 * reference tree; call sites efg/modeling/backbones/sparse_net.py:85-95,125-147). */
size_t efg_spconv_tile_plan_bytes(int64_t m, int kvol);
int efg_spconv_tile_plan(const int32_t* nbr, int64_t m, int kvol, void* plan, size_t plan_bytes, void* stream);
/* out[o][:] = bias + sum_k W[:,k,:] . in[nbr[k][o]][:] over the plan of nbr.  flip_offsets = 1 pairs table column c
 * with weight offset kvol-1-c: the dgrad of a submanifold conv on the FORWARD table's plan (packed: for_dgrad = 1,
 * in = grad_out, cin/cout swapped by the caller), since the transposed table of a symmetric window is the table
 * with its offsets reversed.  The dgrad of a strided conv passes the plan of its transposed table and flip 0.
 * flip_offsets | 2 (natural-order weights, the 16-byte-gather kernel variant: 3-13 % slower on every res18 layer) is refused
 * since round 6.
 * Stream-K (default; EFG_TILE_STREAMK=0 turns it off): on the 64-output-channel split-K shapes of submanifold tables
 * (and of strided tables with >= 128 channels on both sides) the launch is as many workgroups as the device holds and
 * the (row tile, active offset) items of the plan -- its prefix sums are part of the plan buffer -- are cut into equal
 * shares; same sums in a fixed order (reproducible run to run).  This is the one entry point that owns device memory:
 * 32 MB of share scratch + 16 KB of flags per (device, stream), allocated with hipMalloc on first use and kept for the
 * life of the process; calls on ONE stream must be issued by one thread at a time (they are: PyTorch's forward and
 * autograd threads never overlap on a stream). */
/* The launch shape efg_spconv_forward_tiled_f32 uses for these sizes: n-tiles (of 16 output channels) per wave, row
 * sub-tiles per wave, split-K waves -- i.e. the conv_tile_kernel<NT, R, KS> instantiation; the host labels its timings
 * with it (one definition shared with the launcher, so labels cannot drift from what ran). */
int efg_spconv_tile_shape(int cin, int cout, int kvol, int64_t m_in, int64_t m_out, int* nt, int* r, int* ks);
int efg_spconv_forward_tiled_f32(const float* in_feat, int64_t m_in, int cin, const float* packed_weight,
                                 const float* bias, int cout, int kvol, const void* plan, int64_t m_out,
                                 int flip_offsets, float* out_feat, void* stream);
/* TWO convolutions over ONE table in one launch -- the main and the shortcut SparseConv3d of a residual stage's first block
 * (efg/modeling/backbones/sparse_net.py:125-165: both read the block's input through the same strided rulebook), which as two
 * launches are two ramps and two tails on a chip neither fills:
 *   in_b == NULL, out_b != NULL ("N pair", the forward):  out_a = in_a (x) W_a,  out_b = in_a (x) W_b; the unit grid holds the
 *     n-slices of both, each computed exactly as a launch of its own would (bit-identical to two efg_spconv_forward_tiled_f32
 *     calls wherever those do not run stream-K; with stream-K the shares are cut over the joint item list);
 *   in_b != NULL, out_b == NULL ("K pair", their data gradient):  out_a = in_a (x) W_a + in_b (x) W_b accumulated in ONE pass
 *     over the reduction channels of in_a, then of in_b (both cin wide; packed_a / packed_b in the data-gradient layout, plan =
 *     the plan of the transposed table): the sum two efg_spconv_forward_tiled_f32 calls and an addition kernel produce, to
 *     fp32 rounding.
 * Both operands [m_in][cin], both results [m_out][cout], default weight order, no bias.  flip_offsets as above (bit 0 only). */
int efg_spconv_tiled_pair_f32(const float* in_a, const float* in_b, int64_t m_in, int cin, const float* packed_a,
                              const float* packed_b, int cout, int kvol, const void* plan, int64_t m_out, int flip_offsets,
                              float* out_a, float* out_b, void* stream);
/* Stream-K's bounded wait (a share that does not arrive within EFG_TILE_SK_POLLS polls, ~20 ms: two processes
 * spinning on one device, a preempted queue) makes the owner RECOMPUTE the unit -- same sum, different order, twice the
 * work.  Every such event is counted on the device; this returns the count summed over the current device's streams
 * (it synchronises: a measurement / test call, not a step call) and optionally clears it.  A launch issued while its
 * stream is being captured into a HIP graph never uses stream-K (the launch epoch is a kernel argument: a replay would
 * read stale shares). */
int efg_spconv_streamk_fallbacks(int64_t* count_out, int reset);
/* 1 when efg_spconv_forward_tiled_f32 runs a (cin -> cout, kvol) layer on conv_small_kernel<C16, NT, VEC> (the narrow layers
 * of a stem: 16 or 32 reduction channels, or at most 8; at most 32 output channels and 28 offsets; 16-byte aligned features and
 * packed weights, default weight order): a lane loads 16 bytes of ITS neighbour row, a 4 x 4 register transpose makes them
 * the MFMA operands, the layer's packed weights sit in LDS -- and the sums are conv_tile_kernel's bit for bit.  c16 / nt
 * (may be null) receive the instantiation: 16-channel groups of the reduction, 16-column output tiles.  EFG_CONV_SMALL=0
 * (read per call) keeps every layer on conv_tile_kernel. */
int efg_spconv_small_ok(int cin, int cout, int kvol, int* c16, int* nt);
/* 1 when the split-precision arm of the tile kernel covers a (cin -> cout, kvol) convolution of these table sizes. */
int efg_spconv_tile_bf16x3_ok(int cin, int cout, int kvol, int64_t m_in, int64_t m_out);

/* order[m]: the rows of `indices` (int32 [m][4] = b, z, y, x) grouped by the parity of (z, y, x) -- the rows of a
 * stride-2 layer's dgrad that share their set of reachable kernel offsets.  ws: 64 bytes. */
int efg_spconv_parity_order(const int32_t* indices, int64_t m, int32_t* order, void* ws, size_t ws_bytes,
                            void* stream);
size_t efg_spconv_wgrad_workspace_bytes(int64_t m_out, int cin, int cout, int kvol);
/* grad_w[cout][kvol][cin] = sum_o grad_out[o]^T (x) in[nbr[k][o]]  (deterministic two-pass) */
int efg_spconv_wgrad_f32(const float* in_feat, int64_t m_in, int cin, const float* grad_out, int64_t m_out,
                         int cout, int kvol, const int32_t* nbr, float* grad_w, void* ws, size_t ws_bytes,
                         void* stream);
/* The same weight gradient over the tile plan of `nbr` (efg_spconv_tile_plan; the plan the forward pass of the layer
 * uses): a (16-row tile, offset) unit of the plan is four K-steps of the fp32 MFMA whose operands are loaded straight
 * from the feature rows in fragment layout -- no pair compaction, no LDS staging, no workgroup barrier (csrc/spconv_wgt.hip).
 * Covered (efg_spconv_wgrad_tiled_ok): cout 16, 32 or a multiple of 64; cin 1..16, 32 or a multiple of 64; kvol <= 31.  Deterministic
 * two-pass like efg_spconv_wgrad_f32; the grouping of a weight's partial sums follows the plan's tiles, so the two
 * entry points agree to fp32 rounding, not bit for bit.  Replaces the same spconv call (indice-conv backward, weight
 * part) as efg_spconv_wgrad_f32: efg/modeling/backbones/sparse_net.py:85-95 via spconv.SparseConv3d / SubMConv3d. */
int efg_spconv_wgrad_tiled_ok(int cin, int cout, int kvol);
/* The launch schedule of the plan-walking weight gradient for a (cin -> cout) layer over `plan`: a device-side table
 * that cuts the plan's (tile, offset) units into slots of EQUAL unit counts (the offsets of a window differ 5x in active
 * tiles); the number of slots is such that slots x blocks of the layer is a whole number of device fills.  Computed from
 * the plan on the device (no host round trip), a pure function of the plan and the layer shape; layers of equal shape
 * share it.  efg_spconv_wgrad_tiled_f32 must be given the schedule built for ITS (m_out, cin, cout, kvol). */
size_t efg_spconv_wgrad_sched_bytes(int64_t m_out, int cin, int cout, int kvol);
int efg_spconv_wgrad_sched(const void* plan, int64_t m_out, int cin, int cout, int kvol, void* sched, size_t sched_bytes,
                           void* stream);
size_t efg_spconv_wgrad_tiled_workspace_bytes(int64_t m_out, int cin, int cout, int kvol);
int efg_spconv_wgrad_tiled_f32(const float* in_feat, int64_t m_in, int cin, const float* grad_out, int64_t m_out,
                               int cout, int kvol, const void* plan, const void* sched, float* grad_w, void* ws,
                               size_t ws_bytes, void* stream);
/* The weight gradients of TWO layers of one shape over the same input rows, plan and schedule (the pair above) in one launch +
 * one fold: grad_w_a / grad_w_b are bit-identical to two efg_spconv_wgrad_tiled_f32 calls (same slots, same order).
 * Workspace: 2 x efg_spconv_wgrad_tiled_workspace_bytes. */
int efg_spconv_wgrad_tiled_pair_f32(const float* in_feat, int64_t m_in, int cin, const float* grad_out_a,
                                    const float* grad_out_b, int64_t m_out, int cout, int kvol, const void* plan,
                                    const void* sched, float* grad_w_a, float* grad_w_b, void* ws, size_t ws_bytes,
                                    void* stream);

/* SparseConvTensor.dense(): dense f32 [batch, c, D, H, W], fully written (zeros where inactive).
 * feat rows must be in canonical order (perm == NULL) or mapped through perm. */
int efg_sparse_to_dense_f32(const float* feat, int c, const void* index, const int32_t* perm, int batch,
                            const int* shape_host, float* dense, void* stream);
/* backward of dense(): grad_feat[m,c] gathered from grad_dense. */
int efg_dense_to_sparse_f32(const float* grad_dense, int c, const int32_t* indices, int64_t m, int batch,
                            const int* shape_host, float* grad_feat, void* stream);

/* BEV flatten fused with dense(): out f32 [batch, H, W, c*D] with channel index c*D + d -- the values of
 * dense().view(N, C*D, H, W) (sparse_net.py:304-306) in channels-last order, fully written. D <= 16. */
int efg_sparse_to_bev_f32(const float* feat, int c, const void* index, const int32_t* perm, int batch,
                          const int* shape_host, float* out, void* stream);
int efg_bev_to_sparse_f32(const float* grad_out, int c, const int32_t* indices, int64_t m, int batch,
                          const int* shape_host, float* grad_feat, void* stream);

/* ------------------------------------------------------------------------------------------
 * Box / multi-scale deformable attention.  One kernel family behind both
 * efg::box_attn_forward/backward (efg/operators/src/box_attn/box_attn.h:29-83) and
 * efg::ms_deform_attn_forward/backward (efg/operators/src/deform_attn/ms_deform_attn.h:22-63).
 *   value f32 [b,s,h,d]; shapes i64 [l,2] (H,W); level_start i64 [l];
 *   loc f32 [b,lq,h,l,p,2] (x,y in [0,1]); attn f32 [b,lq,h,l,p]; out f32 [b,lq,h*d].
 * d must be a multiple of 4 and <= 256.
 * ---------------------------------------------------------------------------------------- */
int efg_msda_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                         const float* loc, const float* attn, int b, int s, int h, int d, int l, int lq, int p,
                         float* out, void* stream);
/* grad_value must be zero-filled by the caller (it is accumulated into); grad_loc / grad_attn
 * are fully written. */
int efg_msda_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                          const float* loc, const float* attn, const float* grad_out, int b, int s, int h,
                          int d, int l, int lq, int p, float* grad_value, float* grad_loc, float* grad_attn,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused Box3dAttention sampling ($CQ/modules/box_attention.py:62-115): box geometry (centre / relu'd
 * size / rotation from the reference window + raw offsets), softmax over the L*P logits and the
 * bilinear sampling of efg_msda_* in one kernel -- the [B,Lq,H,L,P,2] grid and the softmaxed weights
 * are never materialised.
 *   ref_windows f32 [b,lq,7] (x,y,z,l,w,h,angle; only 0,1,3,4,6 are read)
 *   offsets     f32 [b,lq,h,l,v]  raw linear output, v = 4 (no rotation) | 5 (angle offset)
 *   logits      f32 [b,lq,h,l*p]  raw linear output (pre-softmax)
 *   kernel_indices f32 [p,2]      the k x k lattice (x,y)
 * backward: grad_value must be zero-filled; grad_offsets / grad_logits are fully written; d = 32 only.
 * ---------------------------------------------------------------------------------------- */
int efg_box_attn_fused_forward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                   const float* ref_windows, const float* offsets, const float* logits,
                                   const float* kernel_indices, int b, int s, int h, int d, int l, int lq, int p,
                                   int v, float* out, void* stream);
/* ws (optional, may be NULL): scratch of efg_box_attn_fused_backward_workspace_bytes(...) bytes.  With it, large
 * free-position (decoder) launches sum their grad_value contributions through sorted 8-byte entries instead of
 * d float atomics per corner; without it, or for small launches, atomics are used.  Same results either way.
 * The encoder's tile kernel (queries = the cells of one map) uses the same scratch for the corners that leave the
 * 16 x 16 window of their query tile; without it those corners are written with d float atomics each (a cliff once the
 * boxes grow with training).
 * The first int32 of ws is an overflow counter the call zeroes and the kernels bump when an entry does not fit the
 * bin the counting pass sized for it (the entry is dropped, never written into a neighbour's bin).  It is 0 unless the
 * counting and the writing kernel disagree about a corner's cell; the tests read it after every call. */
size_t efg_box_attn_fused_backward_workspace_bytes(int b, int s, int h, int l, int lq, int p);
int efg_box_attn_fused_backward_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                    const float* ref_windows, const float* offsets, const float* logits,
                                    const float* kernel_indices, const float* grad_out, int b, int s, int h, int d,
                                    int l, int lq, int p, int v, float* grad_value, float* grad_offsets,
                                    float* grad_logits, void* ws, size_t ws_bytes, void* stream);

/* Column sums / split focal sums end inside their partial kernel: the block that draws the last ticket of a self-resetting
 * counter sums the partial rows.  A counter found beyond its launch's block count (a slot that was not at rest) is counted
 * on the device; this call synchronises the CURRENT device, adds the number of ring words that are not zero at rest (a
 * counter left at 0 < k < blocks, which no kernel can see) and optionally puts ring and count back to rest.  Call it
 * between steps, with the device whose streams ran the sums current.  0 in a healthy process. */
int efg_ticket_ring_errors(int64_t* count_out, int reset);

/* The same two calls with ROW STRIDES (floats; 0 = dense) for the offsets and logits matrices and their gradients:
 * Box3dAttention computes both with ONE projection [b, lq, h*l*p + h*l*v] (reference $CQ/modules/box_attention.py:97-104
 * runs two Linears on the same query) and hands the kernels the two column ranges of that matrix; grad_offsets /
 * grad_logits are then the matching column ranges of one gradient matrix. */
int efg_box_attn_fused_forward_strided_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                           const float* ref_windows, const float* offsets, int off_row_stride,
                                           const float* logits, int logit_row_stride, const float* kernel_indices, int b,
                                           int s, int h, int d, int l, int lq, int p, int v, float* out, void* stream);
int efg_box_attn_fused_backward_strided_f32(const float* value, const int64_t* shapes, const int64_t* level_start,
                                            const float* ref_windows, const float* offsets, int off_row_stride,
                                            const float* logits, int logit_row_stride, const float* kernel_indices,
                                            const float* grad_out, int b, int s, int h, int d, int l, int lq, int p, int v,
                                            float* grad_value, float* grad_offsets, float* grad_logits, void* ws,
                                            size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Rotated BEV overlap / IoU and NMS.  Replaces
 * efg::boxes_overlap_bev_gpu / boxes_iou_bev_gpu / nms_gpu / nms_normal_gpu
 * (efg/operators/src/iou3d_nms/iou3d_nms.h:7-16, iou3d_nms.cpp:41-160) and the torch composition
 * boxes_iou3d_gpu (efg/operators/iou3d_nms.py:54-87).
 *   boxes f32 [n,7] = (x, y, z, dx, dy, dz, heading), device memory.
 * ---------------------------------------------------------------------------------------- */
/* out f32 [na,nb], fully written.  mode 0: BEV overlap area, 1: BEV IoU, 2: 3-D IoU. */
int efg_boxes_bev_f32(const float* boxes_a, int na, const float* boxes_b, int nb, int mode, float* out,
                      void* stream);
size_t efg_nms_workspace_bytes(int n);
/* boxes_sorted: already ordered by descending score (the reference sorts in Python, iou3d_nms.py:98-104).
 * Greedy suppression of every later box whose IoU with a kept box is > thresh (rotated != 0: rotated BEV
 * IoU, 0: axis-aligned "normal" IoU).  keep i64 [n] (device) receives the kept row numbers in ascending
 * order, *num_keep (device int32) their count -- the reference returns the count as the host int
 * num_to_keep and fills a CPU LongTensor; here nothing leaves the GPU until the caller asks. */
int efg_nms_f32(const float* boxes_sorted, int n, float thresh, int rotated, int64_t* keep, int* num_keep,
                void* ws, size_t ws_bytes, void* stream);
/* The same over several independent sets in one launch: segment i32 [n] = set id of every box, ascending; inside
 * a set the boxes are ordered by descending score; boxes of different sets never suppress each other.  Replaces a
 * Python loop of nms_gpu calls (TrajectoryFormer runs 11 per sample per step, $TF/trajectoryformer.py:666-676);
 * the kept row numbers come back ascending, i.e. set by set in score order.  Workspace: efg_nms_workspace_bytes(n). */
int efg_nms_segmented_f32(const float* boxes_sorted, const int32_t* segment, int n, float thresh, int rotated,
                          int64_t* keep, int* num_keep, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Points inside vertical cylinders, per cylinder in cloud order (csrc/crop.hip).  Replaces the [boxes x points]
 * distance matrix + per-box Python loop of TrajectoryFormer's crop_current_frame_points
 * (playground/tracking.3d/waymo/trajectoryformer/trajectoryformer.centerpoint/modules/utils.py:361-431).
 * points f32 [n_points][f] (several scenes concatenated); cylinder c tests the rows point_range[c] = {lo, hi} against
 * centre_radius[c] = {x, y, r}: inside <=> sqrt(dx^2 + dy^2) <= r (and points[time_col] < max_time unless
 * time_col < 0).  n_cyl must be a multiple of 16 and every aligned group of 16 must share one point range (pad a
 * scene's list with r = -1).  Call 1: starts = index = NULL, counts i32 [n_cyl] <- number of points inside.
 * Call 2: starts i64 [n_cyl] (exclusive prefix of the counts), index i32 [sum counts] <- the rows (relative to lo) of
 * cylinder c's points, ascending, at index[starts[c] ...].
 * n_chunks (1..64): every scene's point range is cut into that many slices, one workgroup per (16 cylinders, slice), so
 * that a few hundred cylinders fill the chip.  With n_chunks > 1: chunk_counts i32 [n_cyl][n_chunks] is written by call
 * 1 and read by call 2 (same n_chunks), and call 1 ACCUMULATES into counts, which must be zero on entry. */
int efg_cylinder_select_f32(const float* points, int64_t n_points, int f, int time_col, float max_time,
                            const int64_t* point_range, const float* centre_radius, int64_t n_cyl, const int64_t* starts,
                            int32_t* counts, int32_t* index, int n_chunks, int32_t* chunk_counts, void* stream);

/* ------------------------------------------------------------------------------------------
 * Attention over short sequences in exact fp32 (csrc/attention.hip): softmax(scale * Q K^T) V for 1 <= seq_q, seq_k <=
 * 128 tokens and 64-wide heads, one workgroup per (sequence, head), probabilities kept in MFMA accumulators.  Replaces
 * the nn.MultiheadAttention cores of TrajectoryFormer's point encoder (.../trajectoryformer.centerpoint/modules/
 * transformer.py:44-92: self_attn over the 128 points of every trajectory hypothesis, 1232 x 4 sequences per layer, and
 * point_attn of the summary token against them).
 *   element (b, row, h, d) of Q at q + b * q_batch_stride + row * q_row_stride + h * 64 + d, of K / V at
 *   k|v + b * kv_batch_stride + row * kv_row_stride + h * 64 + d (strides in floats, multiples of 4; 16-byte aligned
 *   bases): the fused in-projection outputs [B, S, 3, H, 64] (q | k | v along the channel axis) or [B, Sq, H, 64] +
 *   [B, Sk, 2, H, 64] are read in place.
 *   out f32 [batch, seq_q, heads, 64]; lse f32 [batch, heads, seq_q] = log sum_k exp(scale * <q, k>) (for the backward)
 * Backward: dout f32 [batch, seq_q, heads, 64] -> dq / dk / dv addressed with the strides of q / k / v (so one tensor of
 * the in-projection's layout receives all three); the probabilities are recomputed from lse. */
int efg_attention_fwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k, const float* v,
                          int64_t kv_batch_stride, int64_t kv_row_stride, int64_t batch, int seq_q, int seq_k, int heads,
                          float scale, float* out, float* lse, void* stream);
int efg_attention_bwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k, const float* v,
                          int64_t kv_batch_stride, int64_t kv_row_stride, const float* out, const float* lse,
                          const float* dout, int64_t batch, int seq_q, int seq_k, int heads, float scale, float* dq, float* dk,
                          float* dv, void* stream);

/* The same for sequences of any length with 32-wide heads and an optional boolean mask, blocked over 128 keys with the
 * online softmax (csrc/attention.hip, second half).  Replaces the core of the decoder's nn.MultiheadAttention self-attention
 * in ConQueR / Voxel-DETR (projects/ConQueR/.../transformer.py:258-317: 1000 queries + denoising groups, 8 heads of 32, bool
 * attn_mask).  q / k / v each with their own (batch, row) strides in floats; mask_bits u32 [s, mask_words]: bit (key & 31) of
 * word (key >> 5) in row `query` set = that key is NOT attended (torch's boolean attn_mask, shared by all sequences and
 * heads), NULL = no mask.  A query whose keys are all masked yields zeros (PyTorch: NaN).
 * out f32 [batch, s, heads, 32], lse f32 [batch, heads, s]; backward: dq / dk / dv addressed like q / k / v. */
int efg_attention_long_fwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k,
                               int64_t k_batch_stride, int64_t k_row_stride, const float* v, int64_t v_batch_stride,
                               int64_t v_row_stride, const uint32_t* mask_bits, int mask_words, int64_t batch, int s, int heads,
                               float scale, float* out, float* lse, void* stream);
int efg_attention_long_bwd_f32(const float* q, int64_t q_batch_stride, int64_t q_row_stride, const float* k,
                               int64_t k_batch_stride, int64_t k_row_stride, const float* v, int64_t v_batch_stride,
                               int64_t v_row_stride, const uint32_t* mask_bits, int mask_words, const float* out,
                               const float* lse, const float* dout, int64_t batch, int s, int heads, float scale, float* dq,
                               float* dk, float* dv, float* delta_ws /* scratch f32 [batch, heads, s] */, void* stream);

/* ------------------------------------------------------------------------------------------
 * Linear sum assignment on the device (SURVEY.md section 8(f) "GPU matcher").  Replaces the
 * device->host transfer + scipy.optimize.linear_sum_assignment(C[b]) of $CQ/modules/matcher.py:86-91.
 *   cost f32 [p, nq, g_stride]: p independent problems (layers x scenes), nq queries (rows) x GT
 *        boxes (columns); problem i uses its first ng[i] columns.  ng i32 [p], device memory.
 *   query_of_gt i64 [p, g_stride]: the query assigned to each GT column, -1 for padded columns (and
 *        for unassigned GTs when ng[i] > nq).  The reference's (row_ind, col_ind) pairs are
 *        {(query_of_gt[g], g)}; the assignment equals scipy's for every input, ties included.
 *   status i32 [p] or NULL: 0 ok, 1 infeasible / non-finite costs (scipy raises ValueError there).
 * ---------------------------------------------------------------------------------------- */
int efg_lsap_f32(const float* cost, int n_problems, int nq, int g_stride, const int32_t* ng,
                 int64_t* query_of_gt, int32_t* status, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused residual-add + LayerNorm (the post-norm steps of $CQ/transformer.py:231-243,296-317, which the
 * reference runs as torch add + nn.LayerNorm).  Rows of c floats (c % 4 == 0, c <= 1024).
 *   forward : z = x + residual (residual may be NULL: z = x and z_out is not written), y = LN(z) * gamma + beta,
 *             mean / rstd [rows] saved for backward.
 *   backward: dz (the gradient of x AND of residual), dgamma, dbeta [c]; deterministic two-stage reduction.
 * ---------------------------------------------------------------------------------------- */
int efg_add_layernorm_forward_f32(const float* x, const float* residual, const float* gamma, const float* beta,
                                  float eps, int64_t rows, int c, float* z_out, float* y, float* mean, float* rstd,
                                  void* stream);
size_t efg_add_layernorm_backward_workspace_bytes(int64_t rows, int c);
int efg_add_layernorm_backward_f32(const float* dy, const float* z, const float* mean, const float* rstd,
                                   const float* gamma, int64_t rows, int c, float* dz, float* dgamma, float* dbeta,
                                   void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Point-cloud augmentation + range filter (the step before the path):
 * the per-point work of RandomFlip3D / GlobalRotation / GlobalScaling / GlobalTranslation / FilterByRange /
 * PointShuffle (efg/data/augmentations/extend_3d.py:108-236,286-315) on device clouds.  The random draws stay
 * with the caller (the host mirror draws them with the reference's numpy calls).
 * ---------------------------------------------------------------------------------------- */
enum { EFG_PT_NEG_Y = 0, /* "flip along x axis": y = -y */
       EFG_PT_NEG_X = 1, /* "flip along y axis": x = -x */
       EFG_PT_ROT_Z = 2, /* a = cos, b = sin: (x, y) <- (x*a + y*(-b), x*b + y*a) */
       EFG_PT_SCALE = 3, /* x, y, z *= a */
       EFG_PT_TRANSLATE = 4 /* x += a, y += b, z += c */ };
typedef struct { int kind; float a, b, c; } efg_point_op;
size_t efg_points_transform_filter_workspace_bytes(int64_t n);
/* points f32 [n,f] (x, y, z, features...) -> out [<= n, f]: ops applied in order (<= 8), then -- when range_host
 * (6 floats: min xyz, max xyz, bounds inclusive) is not NULL -- rows outside the range are dropped, order
 * preserved (== points[mask]).  *count (device int32) receives the number of rows written. */
int efg_points_transform_filter_f32(const float* points, int64_t n, int f, const efg_point_op* ops_host, int n_ops,
                                    const float* range_host, float* out, int32_t* count, void* ws, size_t ws_bytes,
                                    void* stream);
/* out[i] = points[index[i]] (row gather; index i64 [m] on the device) -- PointShuffle with a given permutation */
int efg_points_gather_f32(const float* points, const int64_t* index, int64_t m, int f, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused detection losses ($CQ/modules/matcher.py:40-80, $CQ/losses.py:26-108), all layers and scenes per launch.
 *   logits f32 [layers, b, q, c]; boxes f32 [layers, b, q, 7] (cx cy cz l w h rad, normalised);
 *   tgt_labels i64 [b, g], tgt_boxes f32 [b, g, 7] (zero padded); denom: device float (num_boxes).
 * ---------------------------------------------------------------------------------------- */
/* cost f32 [p = layers*b, q, g] = w_bbox*L1(6) + w_class*(focal pos - neg cost at the GT label) + w_giou*(-GIoU3D)
 * + w_rad*|d rad|  (no gradient; feeds efg_lsap_f32) */
int efg_match_cost_f32(const float* logits, const float* boxes, const int64_t* tgt_labels, const float* tgt_boxes,
                       int p, int b, int q, int c, int g, float w_class, float w_bbox, float w_giou, float w_rad,
                       float alpha, float gamma, float* cost, void* stream);
/* out[l] = sum over the n = b*q rows and c classes of sigmoid_focal_loss(logit, class == target_class[l, row]) / denom;
 * target_class i32 [layers, n], -1 = background row */
int efg_focal_loss_forward_f32(const float* logits, const int32_t* target_class, int layers, int64_t n, int c,
                               float alpha, float gamma, const float* denom, float* out, void* stream);
int efg_focal_loss_backward_f32(const float* logits, const int32_t* target_class, int layers, int64_t n, int c,
                                float alpha, float gamma, const float* denom, const float* grad_out,
                                float* grad_logits, void* stream);
/* matched pair i: prediction boxes[l_idx[i], b_idx[i], q_idx[i]] vs tgt_boxes[b_idx[i], g_idx[i]];
 * out f32 [layers, 3] = (sum L1 of x y z l w h, sum (1 - GIoU3D), sum |d rad|) / denom.
 * backward: grad_boxes [layers, b, q, 7] must be zero-filled; a prediction is matched at most once. */
int efg_box_loss_forward_f32(const float* boxes, const float* tgt_boxes, const int64_t* l_idx, const int64_t* b_idx,
                             const int64_t* q_idx, const int64_t* g_idx, int64_t n, int layers, int b, int q, int g,
                             const float* denom, float* out, void* stream);
int efg_box_loss_backward_f32(const float* boxes, const float* tgt_boxes, const int64_t* l_idx, const int64_t* b_idx,
                              const int64_t* q_idx, const int64_t* g_idx, int64_t n, int layers, int b, int q, int g,
                              const float* denom, const float* grad_out, float* grad_boxes, void* stream);

/* Unsorted top-k per row of x [rows, n] (the proposal selection, $CQ/transformer.py:65: torch.topk(sorted=False)):
 * values [rows, k], indices int64 [rows, k], in ascending index order; of the elements equal to the k-th largest value
 * the lowest indices are taken (a fixed rule: two runs pick the same proposals).  One launch, one workgroup per row. */
int efg_topk_unsorted_f32(const float* x, int64_t rows, int n, int k, float* values, int64_t* indices, void* stream);

/* Iterative box refinement of the detection heads ($CQ/heads.py:76-79, $CQ/transformer.py:60-81):
 *   out = sigmoid(delta + inverse_sigmoid(anchor)), inverse_sigmoid as $CQ/modules/utils.py:83-87 (eps 1e-5);
 * n elements, any shape.  Backward: grad_delta = grad * out * (1 - out); with `grad_anchor` (NULL where the anchors are
 * detached reference windows; the heads' aux outputs of layers >= 1 pass anchors with a graph, $CQ/voxel_detr.py:171-180)
 * also grad_anchor = grad_delta * d inverse_sigmoid(anchor), the clamps differentiated as autograd does. */
int efg_box_refine_forward_f32(const float* delta, const float* anchor, int64_t n, float eps, float* out, void* stream);
int efg_box_refine_backward_f32(const float* grad, const float* out, const float* anchor, int64_t n, float eps,
                                float* grad_delta, float* grad_anchor, void* stream);

/* ------------------------------------------------------------------------------------------
 * BatchNorm1d (training statistics) + optional residual + optional ReLU over sparse features [m, c]
 * (the norm / activation steps of efg/modeling/backbones/sparse_net.py:85-95,120-165, which the reference runs as
 * nn.BatchNorm1d, an add and nn.ReLU).  c % 4 == 0, c <= 1024.
 *   forward : y = relu?((x - mean) * invstd * weight + bias + residual?); mean / invstd [c] saved for backward;
 *             running_mean / running_var (NULL or both) updated with `momentum` and the unbiased variance,
 *             *num_batches_tracked (i64, NULL ok) incremented.
 *   backward: dx, dresidual (= dy masked by y > 0; NULL ok), dweight / dbias [c].
 *   ws: efg_bn_workspace_bytes(c).
 * ---------------------------------------------------------------------------------------- */
size_t efg_bn_workspace_bytes(int c);
int efg_bn_forward_f32(const float* x, const float* residual, const float* weight, const float* bias,
                       float* running_mean, float* running_var, int64_t* num_batches_tracked, float momentum, float eps,
                       int64_t m, int c, int relu, float* y, float* mean, float* invstd, void* ws, size_t ws_bytes,
                       void* stream);
int efg_bn_backward_f32(const float* dy, const float* x, const float* y, const float* weight, const float* mean,
                        const float* invstd, int64_t m, int c, int relu, float* dx, float* dresidual, float* dweight,
                        float* dbias, void* ws, size_t ws_bytes, void* stream);

/* ---- GroupNorm over a channels-last map [batch][rows = H*W][c] ---------------------------------------------
 * The input projection in front of the transformer ($CQ/voxel_detr.py:43-51: Conv2d 1x1 + nn.GroupNorm(32, 256)),
 * computed in the token layout the encoder reads instead of NCHW (two 72 MB transposing copies each way).
 * Biased variance, eps inside the square root (torch.nn.functional.group_norm).  c / groups must be a multiple of 4.
 *   forward : y, mean / rstd [batch * groups];   backward: dx, dweight / dbias [c].   ws: efg_gn_workspace_bytes. */
size_t efg_gn_workspace_bytes(int batch, int c);
int efg_gn_forward_f32(const float* x, const float* weight, const float* bias, float eps, int batch, int64_t rows,
                       int c, int groups, float* y, float* mean, float* rstd, void* ws, size_t ws_bytes, void* stream);
int efg_gn_backward_f32(const float* dy, const float* x, const float* weight, const float* mean, const float* rstd,
                        int batch, int64_t rows, int c, int groups, float* dx, float* dweight, float* dbias, void* ws,
                        size_t ws_bytes, void* stream);

/* ---- column sums: the bias gradient of the path's Linear layers -------------------------------------------
 * Replaces autograd's grad_output.sum(0) for every nn.Linear of the transformer ($CQ/transformer.py:215-243,
 * 273-317; $CQ/modules/blocks.py:5-17; $CQ/modules/box_attention.py:31-40).  x: rows x cols fp32, row-major with
 * `row_stride` floats between rows; out[cols].  Deterministic two-pass reduction; rows == 0 zero-fills. */
size_t efg_colsum_workspace_bytes(int64_t rows, int cols);
int efg_colsum_f32(const float* x, int64_t rows, int cols, int64_t row_stride, float* out, void* ws, size_t ws_bytes,
                   void* stream);
/* The ReLU backward of a Linear + ReLU folded into the same pass (operators/linear.py LinearFunction.backward): g_out =
 * (y > 0) ? g : 0, what autograd's threshold_backward writes, and out[c] = column sums of g_out in efg_colsum_f32's order.
 * Contiguous [rows, cols] matrices, cols % 4 == 0, 16-byte aligned; workspace efg_colsum_workspace_bytes(rows, cols);
 * g_out may alias g. */
int efg_relu_bwd_colsum_f32(const float* g, const float* y, int64_t rows, int cols, float* g_out, float* out, void* ws,
                            size_t ws_bytes, void* stream);

/* ---- split-precision (bf16 x 3) GEMM: the A/B arm of the bench, never the default path ---------------------
 * C[m, n] = A[m, k] . B[k, n] (+ bias[n]) (ReLU if relu != 0), fp32 in and out, every operand split into two bf16 terms
 * (x = hi + lo) and hi.hi + hi.lo + lo.hi accumulated in fp32 on the bf16 MFMA: ~2^-16 relative per product instead of
 * fp32's 2^-24, at 3/16 of the fp32 MFMA time (gemm_bf16x3.hip).  Stands where the encoder's nn.Linear products
 * ($CQ/transformer.py:215-243, $CQ/modules/box_attention.py:31-40) call hipBLASLt in fp32 when EFG_GEMM_ARM=bf16x3.
 *   pack: B(kk, nn) = w[kk * stride_k + nn * stride_n] (so a Linear weight [out, in] packs as B = W^T with
 *         stride_k = 1, stride_n = in, and as B = W with stride_k = in, stride_n = 1), split and laid out in the
 *         order the MFMA lanes read it; `packed` holds efg_gemm_bf16x3_pack_bytes(k, n) bytes.
 *   gemm: A rows 16-byte aligned, lda >= k, k % 4 == 0, lda % 4 == 0; C rows ldc floats apart. */
size_t efg_gemm_bf16x3_pack_bytes(int k, int n);
int efg_gemm_bf16x3_pack_f32(const float* w, int64_t stride_k, int64_t stride_n, int k, int n, void* packed,
                             void* stream);
/* Both layouts of an nn.Linear weight w[n_out, n_in] in one launch: packed_fwd = pack(B = w^T; efg_gemm_bf16x3_pack_bytes(n_in,
 * n_out) bytes) for y = x w^T, packed_dgrad = pack(B = w; ..._pack_bytes(n_out, n_in) bytes) for dx = dy w. */
int efg_gemm_bf16x3_pack_linear_f32(const float* w, int n_out, int n_in, void* packed_fwd, void* packed_dgrad, void* stream);
int efg_gemm_bf16x3_f32(const float* a, int64_t m, int k, int64_t lda, const void* packed_b, int n, const float* bias,
                        int relu, float* c, int64_t ldc, void* stream);
/* The weight gradient of the same arm: dw[n, k] = sum over the m rows of g[m, n] * x[m, k] (g = grad_output, x = the
 * layer's input, both row-major fp32 with 16-byte aligned rows, n and k multiples of 4), split products as above, the row
 * range summed in a fixed chunk order (deterministic).  ws: efg_gemm_bf16x3_wgrad_workspace_bytes(m, n, k). */
size_t efg_gemm_bf16x3_wgrad_workspace_bytes(int64_t m, int n, int k);
int efg_gemm_bf16x3_wgrad_f32(const float* g, int64_t ldg, const float* x, int64_t ldx, int64_t m, int n, int k, float* dw,
                              void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
