/* gsplat_hip.h -- C ABI of the MI355X (gfx950) differentiable Gaussian-splat rasterizer.
 *
 * This is the drop-in boundary underneath the three Python calls splat-trainer makes into taichi_splatting
 * (the reference has no C/FFI layer of its own):
 *
 *     project_to_image(gaussians, camera_params, config)            splat_trainer/scene/mlp_scene.py:375,415
 *     render_projected(indexes, gaussians2d, features, depth, ...)  splat_trainer/scene/mlp_scene.py:377-378,418-419
 *     evaluate_sh_at(sh_features, positions, indexes, camera_pos)   splat_trainer/scene/transfer_sh.py:49
 *     loss.backward()  (autograd node of the above)                 splat_trainer/trainer/trainer.py:512
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (torch ``data_ptr()``) unless the name ends in ``_host``;
 *   - all floating-point data is float32, row-major, contiguous; index tensors at the boundary are int64;
 *   - nothing here allocates, frees or synchronises: callers pass workspaces sized by the *_workspace_bytes
 *     queries and a ``hipStream_t`` (as ``void*``; NULL = default stream); calls only enqueue work;
 *   - camera: ``T_camera_world`` = 16 floats (4x4 row-major world->camera), ``projection`` = fx, fy, cx, cy in
 *     pixels (splat_trainer/trainer/trainer.py:291-301); both stay on the device, so no host sync is needed;
 *   - return value: 0 (GSR_OK) or a negative GSR_ERR_* code; the library never throws and never prints;
 *   - M = 0 and O = 0 are valid inputs everywhere (the reference treats "no visible points" as a caller-level
 *     error, splat_trainer/trainer/trainer.py:507-509, not a boundary error).
 *
 * The maths each entry point implements is specified in oracle/torch_oracle.py (header comment).
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSR_OK 0
#define GSR_ERR_INVALID_ARGUMENT -1
#define GSR_ERR_WORKSPACE_TOO_SMALL -2
#define GSR_ERR_LAUNCH_FAILED -3
#define GSR_ERR_UNSUPPORTED -4

#define GSR_ROW_FLOATS 16      /* packed per-splat row, 64 bytes, splat order:  u v A B | C op qlim f0 | f1 f2 depth log2(op) | 0 0 0 0
                                  (qlim = min(q_max, 2 ln(op / alpha_threshold)): a pixel contributes iff q <= qlim);
                                  the gradient rows use the same pitch:  mx my mxx mxy | myy dop prune split | df0 df1 df2 vis | 0 0 0 0 */
#define GSR_PARTIAL_FLOATS 12  /* per-(tile,splat) gradient partial: mx my mxx mxy | myy m0 prune split | df0 df1 df2 -
                                  (m* = moments of G dL/dG about the splat's mean, m0 the zeroth; the per-splat sweep turns
                                  their sums into du = A mx + B my, dv = B mx + C my, dA = -mxx/2, dB = -mxy, dC = -myy/2,
                                  dopacity = m0 / opacity) */

#ifndef GSR_HAVE_RASTER_PARAMS
#define GSR_HAVE_RASTER_PARAMS
/* Mirrors the RasterConfig fields the reference sets (trainer.py:305-310) plus this build's constants. */
typedef struct GsrRasterParamsC {
  float alpha_threshold;  /* 1/255 */
  float clamp_max_alpha;  /* 0.99 */
  float T_eps;            /* 1 - saturate_threshold (1e-4) */
  float q_max;            /* gaussian_scale^2 (9) */
  float blur;             /* blur_cov (+ aa_blur when antialias) */
  int32_t antialias;      /* 0 / 1 */
  int32_t tile_size;      /* must be 16 */
  float margin_px;        /* margin_tiles * tile_size */
} GsrRasterParamsC;
#endif

/* Heavy-tile list segmentation (optional; pass NULL where a GsrSegmentsC* is taken to composite every tile with one
 * wave).  All pointers are device buffers owned by the caller; tile_seg / seg_desc / seg_total are filled by
 * gsr_segment_plan, the seg_* pixel buffers are scratch written by the forward pass and read by the backward pass. */
typedef struct GsrSegmentsC {
  const uint32_t* tile_seg;   /* [num_tiles,2]: first segment, number of segments (0 = light tile); followed by
                                 [heavy_capacity]: the segments of heavy tiles, listed compactly by the plan */
  const uint32_t* seg_desc;   /* [capacity,4]: tile, list start, list end, index within the tile */
  const uint32_t* seg_total;  /* GSR_SEG_TOTAL_WORDS device words: [0] segments of this frame (<= capacity), [1] of them in
                                 heavy tiles, [16 + 32 x + c]: tiles of XCD band x in length class c (the forward pass's launch order) */
  int64_t capacity;           /* from gsr_segment_capacity */
  int64_t heavy_capacity;     /* from gsr_segment_heavy_capacity (<= capacity) */
  float* seg_P;               /* [capacity,256] */
  float* seg_TC;              /* [capacity,256,4], 16-byte aligned: (T, c0, c1, c2) per pixel slot */
  int32_t* seg_last;          /* [capacity,256] */
  float* seg_median;          /* [capacity,256]; NULL unless a median depth image is requested */
  const uint32_t* tile_order; /* [GSR_TILE_ORDER_WORDS(num_tiles)]: the forward pass's launch order (tiles by XCD band and
                                 length class, filled by gsr_segment_plan), or NULL: tiles in image order */
} GsrSegmentsC;
#define GSR_SEG_TOTAL_WORDS 272
#define GSR_TILE_ORDER_WORDS(num_tiles) (8 * 32 * ((((num_tiles) + 127) / 128) * 16))

/* sizeof of the ABI's structs as the library was compiled: 0 GsrRasterParamsC, 1 GsrSegmentsC, 2 GsrFrameC,
 * 3 GsrFramePlanC, 4 GsrFrameResultC, 5 GsrFrameBackwardC (-1 otherwise) -- for a binding to check its own layout. */
int64_t gsr_struct_bytes(int32_t which);
int gsr_abi_version(void);                 /* bumped on any signature change (currently 30) */
const char* gsr_error_string(int code);

/* ---- device-wide primitives (K5: radix bin + depth sort) ------------------------------------------------ */
size_t gsr_scan_workspace_bytes(int64_t n);
int gsr_exclusive_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_dev, void* workspace,
                           size_t workspace_bytes, void* stream);
/* Same scan with an overflow guard: *overflow_dev (zero on entry) is set to 1 when a prefix or the total reaches 2^31
 * (beyond what the 32-bit list positions downstream can address; also raised before any u32 wrap can occur). */
int gsr_exclusive_scan_u32_checked(const uint32_t* in, uint32_t* out, int64_t n, uint32_t* total_dev,
                                   uint32_t* overflow_dev, void* workspace, size_t workspace_bytes, void* stream);
size_t gsr_sort_workspace_bytes(int64_t n);
/* Stable LSD radix sort of (key, value) pairs by key bits [begin_bit, end_bit); ping-pongs a<->b.
 * Returns 0 if the result ends in (keys_a, vals_a), 1 if in (keys_b, vals_b), negative on error.
 * n_dev (may be NULL): device word with the element count when n is only the capacity of the arrays (workspace sized
 * for n): the sort can then be enqueued before the count has reached the host; min(n, *n_dev) elements are sorted. */
int gsr_sort_pairs_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, int64_t n,
                       int vals_are_iota, int begin_bit, int end_bit, void* workspace, size_t workspace_bytes,
                       const uint32_t* n_dev, void* stream);
/* Same, carrying a second value array with every key (the tile sort moves instance id + depth rank). */
int gsr_sort_pairs2_u32(uint32_t* keys_a, uint32_t* vals_a, uint32_t* vals2_a, uint32_t* keys_b, uint32_t* vals_b,
                        uint32_t* vals2_b, int64_t n, int vals_are_iota, int begin_bit, int end_bit, void* workspace,
                        size_t workspace_bytes, const uint32_t* n_dev, void* stream);

/* ---- K1 frustum cull + compaction  (project_to_image, first half) --------------------------------------- */
size_t gsr_cull_workspace_bytes(int64_t N);
/* indexes_out: int64[N] capacity, ascending point order; count_dev: uint32 = M. */
int gsr_frustum_cull(const float* position, int64_t N, const float* T_camera_world, const float* projection,
                     int32_t W, int32_t H, float near_plane, float far_plane, float margin_px,
                     int64_t* indexes_out, uint32_t* count_dev, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K2 3D->2D projection forward / backward  (project_to_image, second half) --------------------------- */
/* gaussians2d_out: [M,6] = u v A B C opacity; depth_out: [M].  count_dev (may be NULL): device word holding the
 * true number of valid entries of ``indexes`` (<= M); lets the call be enqueued right behind gsr_frustum_cull,
 * before the host has read the count back.  depth_keys_out (may be NULL): [M] the depth sort's keys, exactly what
 * gsr_depth_keys(depth_out, bias, max_key) would give (saves that launch when the caller rasterizes next). */
int gsr_project_forward(const float* position, const float* log_scaling, const float* rotation_xyzw,
                        const float* alpha_logit, const int64_t* indexes, int64_t M, const float* T_camera_world,
                        const float* projection, const GsrRasterParamsC* params_host, float* gaussians2d_out,
                        float* depth_out, const uint32_t* count_dev, uint32_t* depth_keys_out, uint32_t depth_key_bias,
                        uint32_t depth_key_max, void* stream);
/* Rows ``indexes`` of the N-sized gradient tensors are written (accumulate = 0: other rows untouched, pass
 * zeros) or added to (accumulate = 1: "+=" straight into the caller's .grad buffers; rows are unique, no atomics). */
int gsr_project_backward(const float* position, const float* log_scaling, const float* rotation_xyzw,
                         const float* alpha_logit, const int64_t* indexes, int64_t M, const float* T_camera_world,
                         const float* projection, const GsrRasterParamsC* params_host, const float* dL_dgaussians2d,
                         const float* dL_ddepth, float* d_position, float* d_log_scaling, float* d_rotation,
                         float* d_alpha_logit, int32_t accumulate, void* stream);

/* ---- K3 spherical-harmonics colour forward / backward  (evaluate_sh_at) --------------------------------- */
/* sh_features [N,3,K], K in {1,4,9,16}; colour = 0.5 + sum_k sh[c][k] Y_k(normalize(p - camera_pos)). */
/* jacobian_out [M,9] or NULL: d colour / d position (row-major 3x3 per splat), saved for the backward pass.
 * count_dev (may be NULL): device word holding the true row count when M is only an upper bound -- lets the caller
 * enqueue the colours right behind K1/K2, before the visible count has reached the host (as gsr_project_forward). */
int gsr_sh_forward(const float* sh_features, const float* positions, const int64_t* indexes, int64_t M, int32_t K,
                   const float* camera_pos, float* colors_out, float* jacobian_out, const uint32_t* count_dev,
                   void* stream);
/* d_sh_features [N,3,K] and d_positions [N,3] (may be NULL): rows ``indexes`` are written (accumulate = 0, pass
 * zeros) or added to (accumulate = 1).
 * d_positions is the gradient through the view direction normalize(p - camera_pos). */
int gsr_sh_backward(const float* dL_dcolors, const float* sh_features, const float* positions, const int64_t* indexes,
                    int64_t M, int32_t K, const float* camera_pos, const float* jacobian /* [M,9] or NULL */,
                    float* d_sh_features, float* d_positions, int32_t accumulate, void* stream);

/* inverse_out [N] int32: rank m of row r in the ascending list `indexes` [M], or -1 when the row is not in the list. */
int gsr_inverse_map(const int64_t* indexes, int64_t M, int64_t N, int32_t* inverse_out, void* stream);
/* Same gradient as gsr_sh_backward, but d_sh_features [N,3,K] is OVERWRITTEN for every scene row (zeros where the
 * camera saw nothing): the caller needs neither a zero-filled buffer nor a read-modify-write.  inverse [N] from
 * gsr_inverse_map, or NULL when M == N (every row visible, indexes = 0..N-1).  d_positions [N,3] (may be NULL) is
 * accumulated ("+="). */
int gsr_sh_backward_dense(const float* dL_dcolors, const float* sh_features, const float* positions,
                          const int32_t* inverse, int64_t M, int64_t N, int32_t K, const float* camera_pos,
                          const float* jacobian /* [M,9] or NULL */, float* d_sh_features, float* d_positions,
                          void* stream);

/* Multi-camera form for the data-parallel path.  Camera c's colour gradients, scattered to scene rows (all-zero rows
 * = not visible), start at dL_dcolors_dense + c * dense_stride ([N,3] floats each, dense_stride >= 3N); its position
 * (3 floats) at camera_positions + c * camera_stride.  (The strides let one all-gathered block per camera carry both:
 * [N+1,3] with the position in the last row.)  Adds sum_c g_c (x) Y(dir_c) to d_sh_features [N,3,K] -- or overwrites
 * every row when accumulate = 0 -- and adds the view-direction term to d_positions [N,3] (may be NULL), cameras in
 * index order.  Lets ranks exchange the 16x smaller colour gradients instead of all-reducing d_sh. */
int gsr_sh_backward_multi(const float* dL_dcolors_dense, int64_t dense_stride, const float* camera_positions,
                          int64_t camera_stride, int32_t num_cameras, const float* sh_features, const float* positions,
                          int64_t N, int32_t K, float* d_sh_features, float* d_positions, int32_t accumulate,
                          void* stream);

/* The depth sort's keys of the visible splats from their positions alone (what gsr_project_forward /
 * gsr_project_sh_forward write as depth_keys_out, bit for bit): lets a caller start the depth sort right behind the cull,
 * on a second stream, while the projection / colour sweep runs on the first (the frame driver does). */
int gsr_depth_keys_from_positions(const float* position, const int64_t* indexes, int64_t M, const uint32_t* count_dev,
                                  const float* T_camera_world, const float* projection, uint32_t depth_key_bias,
                                  uint32_t depth_key_max, uint32_t* keys_out, void* stream);

/* ---- K2 + K3 fused  (render_gaussians: project_to_image + evaluate_sh_at in one sweep) ------------------------ */
/* One [M,16] row per visible splat (GSR_ROW_FLOATS), written whole:  u v A B | C opacity qlim f0 | f1 f2 depth 0 | 0 0 0 0,
 * exactly the values gsr_project_forward + gsr_sh_forward return (same device code).  screen_scale_out [M,2];
 * jacobian_out [M,9] or NULL, depth_keys_out [M] or NULL, count_dev or NULL as in the two single calls. */
int gsr_project_sh_forward(const float* position, const float* log_scaling, const float* rotation_xyzw,
                           const float* alpha_logit, const float* sh_features, int32_t K, const int64_t* indexes, int64_t M,
                           const float* T_camera_world, const float* projection, const float* camera_pos,
                           const GsrRasterParamsC* params_host, float* rows_out, float* screen_scale_out,
                           float* jacobian_out, const uint32_t* count_dev, uint32_t* depth_keys_out,
                           uint32_t depth_key_bias, uint32_t depth_key_max, void* stream);
/* Backward of the geometry half from the packed gradient rows of gsr_reduce_gradients, one sequential sweep in splat
 * order: K2 backward into the N-sized gradient tensors (all four or none -- d_position NULL skips the geometry), plus
 * the position term of the colour gradient when ``jacobian`` [M,9] is given.  mode 0: rows ``indexes`` are written (the
 * others untouched: pass zeros); 1: added to ("+="); 2: EVERY scene row is written, zeros where the camera saw nothing
 * (no zero-fill by the caller, no read-modify-write) -- ``inverse`` [N] from gsr_inverse_map, or NULL when M == N.
 * dL_dgaussians2d_extra [M,6] / dL_ddepth [M] (may be NULL): gradients that reached gaussians2d / depth from outside the
 * rasterizer.  The rows' scalar columns are copied out where asked: d_colors_out [M,3] (input of the gsr_sh_backward*
 * calls), prune_cost_out / split_score_out / visibility_out [M]. */
int gsr_project_backward_rows(const float* position, const float* log_scaling, const float* rotation_xyzw,
                              const float* alpha_logit, const int64_t* indexes, int64_t M, const int32_t* inverse,
                              int64_t N, const float* T_camera_world, const float* projection,
                              const GsrRasterParamsC* params_host, const float* rows /* unused (the conic is recomputed); may be NULL */,
                              const float* grad_rows, const float* dL_dgaussians2d_extra, const float* dL_ddepth,
                              const float* jacobian,
                              float* d_position, float* d_log_scaling, float* d_rotation, float* d_alpha_logit,
                              int32_t mode, float* d_colors_out, float* prune_cost_out, float* split_score_out,
                              float* visibility_out, void* stream);

/* ---- K4 tile overlap count / key emit, tile ranges  (render_projected, binning) ------------------------- */
/* depth -> sortable u32 keys, key = min(bits(depth) - bias, max_key), monotone in depth.  gsr_depth_key_range gives
 * (bias, max_key) for a camera's near / far planes: the keys of a frame then span only bits(far) - bits(near) -- 27
 * bits for 0.1 .. 100 -- and gsr_sort_pairs_u32(0, bit length of max_key) needs three passes instead of four (9-bit
 * digits).  bias 0 / max_key 0xFFFFFFFF (what the range call returns for a non-positive or infinite range): plain keys. */
int gsr_depth_key_range(float near_plane, float far_plane, uint32_t* bias_out, uint32_t* max_key_out);
int gsr_depth_keys(const float* depth, int64_t M, uint32_t bias, uint32_t max_key, uint32_t* keys_out, void* stream);
/* The three-call form's (M,6) gaussians2d + (M) depth + (M,C) features packed into the [M,16] rows everything downstream
 * reads (GSR_ROW_FLOATS; written whole, 64 bytes per splat, splat order); also screen_scale_out [M,2] =
 * (sigma_major, sigma_minor) in pixels, sqrt of the eigenvalues of the blurred 2D covariance.  The one-call form gets
 * the same rows straight from gsr_project_sh_forward. */
int gsr_pack_rows(const float* gaussians2d, const float* depth, const float* features, int64_t M, int32_t C,
                  const GsrRasterParamsC* params_host, float* rows_out, float* screen_scale_out, void* stream);
/* For rank k in depth order (order[k] = splat id): gathers the splat's row -- the one crossing of the depth-order
 * permutation on the forward side, one 64-byte line per splat -- and writes the number of tiles its support touches (a
 * tile counts when the support reaches the pixel centres of its upper or lower half) and tile_hits_out [M,4] uint32:
 * which tiles of the splat's extent were counted and which halves of each, for gsr_tile_emit (opaque to the caller). */
int gsr_tile_count(const float* rows, const uint32_t* order, int64_t M, int32_t W, int32_t H,
                   const GsrRasterParamsC* params_host, uint32_t* count_out, uint32_t* tile_hits_out,
                   const uint32_t* M_dev /* NULL, or the device word with the visible count when M is a capacity: ranks at
                                            or beyond it get count 0 */, void* stream);
/* gsr_tile_count followed by the exclusive scan of the counts in TWO launches instead of three (the count pass leaves
 * per-block totals in the workspace, the scan pass adds them up itself): offsets_out [M], total_dev = the pair count O,
 * overflow_flag raised when O reaches 2^31 (as gsr_exclusive_scan_u32_checked does).  M <= GSR_TILE_COUNT_OFFSETS_MAX
 * (beyond that the totals in front of a block are too many to add up per block: use gsr_tile_count + the scan). */
#define GSR_TILE_COUNT_OFFSETS_MAX 4194304
size_t gsr_tile_count_offsets_workspace_bytes(int64_t M);
int gsr_tile_count_offsets(const float* rows, const uint32_t* order, int64_t M, int32_t W, int32_t H,
                           const GsrRasterParamsC* params_host, uint32_t* count_out, uint32_t* tile_hits_out,
                           const uint32_t* M_dev, uint32_t* offsets_out, uint32_t* total_dev, uint32_t* overflow_flag,
                           void* workspace, size_t workspace_bytes, void* stream);
/* offsets = exclusive scan of count.  Instance i of rank k gets keys[offsets[k]+i] = tile id and
 * inst2splat[...] = order[k] | (half mask << 30): instances are emitted rank-major (already depth-sorted), the value they
 * carry is the splat id the composite kernels fetch the row by.
 * capacity = number of entries the two output arrays hold: instances at or beyond it are dropped, so the call may be
 * enqueued into buffers sized from a guess while the exact total is still on its way to the host (the caller compares
 * the total with the capacity afterwards and emits again if it was too small). */
int gsr_tile_emit(const float* rows, const uint32_t* order, const uint32_t* offsets, const uint32_t* tile_hits, int64_t M,
                  int32_t W, int32_t H, const GsrRasterParamsC* params_host, uint32_t* keys_out, uint32_t* inst2splat_out,
                  int64_t capacity, const uint32_t* M_dev /* as gsr_tile_count */, void* stream);
/* From the tile-sorted keys: per-tile [start, end).  tile_range must be zero-filled by the caller:
 * [num_tiles, 2] uint32. */
int gsr_tile_ranges(const uint32_t* sorted_keys, int64_t O, int32_t num_tiles, uint32_t* tile_range,
                    const uint32_t* O_dev /* NULL, or the device word with the count when O is a capacity */, void* stream);

/* ---- heavy-tile list segmentation ----------------------------------------------------------------------- */
/* One wave walks one tile's depth-sorted list serially.  A tile with more than seg_pairs pairs is cut into segments of
 * max(seg_pairs, ~len / 32) pairs: its forward walk (still one wave) leaves a checkpoint at every segment end and its
 * backward pass runs one wave per segment; a tile with more than heavy_min (>= seg_pairs) pairs is composited forward by
 * one wave per segment as well.
 *
 * seg_pairs_cfg / heavy_min_cfg > 0 fix the two thresholds; <= 0 selects them from the frame's pair count O (and
 * needs_grad): gsr_segment_thresholds returns the values a frame with O pairs is cut with.  The plan kernel evaluates
 * the same rule, from O or -- O_dev != NULL -- from the device word holding the count, so the plan can be enqueued
 * before the count has reached the host.
 * gsr_segment_capacity: host-side bound on the number of segments (sizes the buffers) of a frame with exactly O pairs
 * (O_is_bound = 0) or with at most O pairs (O_is_bound = 1). */
int gsr_segment_thresholds(int32_t seg_pairs_cfg, int32_t heavy_min_cfg, int64_t O, int32_t num_tiles, int32_t needs_grad,
                           int32_t* seg_pairs_out, int32_t* heavy_min_out);
int64_t gsr_segment_capacity(int64_t O, int32_t O_is_bound, int32_t seg_pairs_cfg, int32_t heavy_min_cfg, int32_t num_tiles,
                             int32_t needs_grad);
/* ... and on the number of those that belong to heavy tiles (the forward passes A and C launch one block each). */
int64_t gsr_segment_heavy_capacity(int64_t O, int32_t O_is_bound, int32_t seg_pairs_cfg, int32_t heavy_min_cfg,
                                   int32_t num_tiles, int32_t needs_grad);
/* tile_seg_out [2 num_tiles + heavy_capacity], seg_desc_out [capacity,4], seg_total_out [GSR_SEG_TOTAL_WORDS],
 * tile_order_out [GSR_TILE_ORDER_WORDS(num_tiles)] or NULL (see GsrSegmentsC); seg_total_out must be ZERO on entry
 * (tiles reserve their slots with integer atomics on it). */
int gsr_segment_plan(const uint32_t* tile_range, int32_t num_tiles, int32_t seg_pairs_cfg, int32_t heavy_min_cfg,
                     int32_t needs_grad, int64_t O, const uint32_t* O_dev, int64_t capacity, int64_t heavy_capacity,
                     uint32_t* tile_seg_out, uint32_t* seg_desc_out, uint32_t* seg_total_out, uint32_t* tile_order_out,
                     void* stream);

/* ---- K6 alpha-composite forward ------------------------------------------------------------------------- */
/* image [H,W,C]; final_T [H,W]; last [H,W] int32 = 1 + list position of the last contributing splat;
 * median_depth [H,W] or NULL; vis_partial [O] (indexed by instance id; must be zero-filled) and pair_vis [O]
 * (the same per-(tile,splat) visibility sum_px T*alpha, indexed by sorted list position) or both NULL. */
int gsr_composite_forward(const float* rows /* [M,16] */, const uint32_t* sorted_splat, const uint32_t* sorted_inst,
                          const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                          const GsrRasterParamsC* params_host, float* image_out, float* final_T_out,
                          int32_t* last_out, float* median_depth_out, float* vis_partial_out, float* pair_vis_out,
                          const GsrSegmentsC* segments_host /* or NULL */,
                          int32_t prefetch_rows /* speed only: touch the rows 32-64 pairs ahead of the walk (pays once the
                                                   row table has outgrown the caches: GSR_PREFETCH_MIN_ROWS).  With it
                                                   the walk fetches sorted_splat four words at a time and may read up
                                                   to THREE WORDS PAST the O-th entry (values unused): the buffer must be
                                                   readable that far (gsr_frame_plan sizes it so) */,
                          void* stream);
#ifndef GSR_PREFETCH_MIN_ROWS
#define GSR_PREFETCH_MIN_ROWS 1000000
#endif

/* ---- K7 alpha-composite backward (per-pixel reverse walk) ----------------------------------------------- */
/* partial_out [O,12]: written only for pairs with pair_vis > 0 (the others are never read). */
int gsr_composite_backward(const float* rows /* [M,16] */, const uint32_t* sorted_splat, const uint32_t* sorted_inst,
                           const float* pair_vis, const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                           const GsrRasterParamsC* params_host, const float* final_T, const int32_t* last,
                           const float* dL_dimage, const float* image /* the forward output; needed with segments */,
                           float* partial_out, const GsrSegmentsC* segments_host /* the forward pass's, or NULL */,
                           void* stream);

/* ---- deterministic per-splat reductions of the per-(tile,splat) partials -------------------------------- */
/* visibility_out [M] indexed by splat (not rank).  capacity = slots vis_partial holds (the pair count, or the bound
 * the buffers were sized with: slots at or beyond it are not read). */
int gsr_reduce_visibility(const float* vis_partial, const uint32_t* offsets, const uint32_t* count,
                          const uint32_t* order, int64_t M, float* visibility_out, int64_t capacity,
                          const uint32_t* M_dev /* as gsr_tile_count */, void* stream);
/* grad_rows_out [M,16] indexed by splat, each row written whole (the one crossing of the depth-order permutation on the
 * backward side):  mx my mxx mxy | myy dop prune_cost split_score | df0 df1 df2 visibility | 0 0 0 0  (m*: the summed
 * moments of GSR_PARTIAL_FLOATS; gsr_project_backward_rows / gsr_unpack_grad_rows turn them into d(u, v, A, B, C) with
 * the conic of the forward row).  The visibility column holds the same bits gsr_reduce_visibility returns. */
int gsr_reduce_gradients(const float* partial, const float* vis_partial, const uint32_t* offsets,
                         const uint32_t* count, const uint32_t* order, int64_t M, float* grad_rows_out, void* stream);
/* The rows taken apart for the three-call form: d_gaussians2d [M,6], d_features [M,C]; prune_cost / split_score /
 * visibility [M] (each may be NULL). */
int gsr_unpack_grad_rows(const float* rows /* the forward rows [M,16] */, const float* grad_rows, int64_t M, int32_t C,
                         float* d_gaussians2d, float* d_features, float* prune_cost_out, float* split_score_out,
                         float* visibility_out, void* stream);

/* ---- frame driver: the forward half of render_gaussians(use_sh = True) behind one call ---------------------
 * (scene.render -> project_to_image + evaluate_sh_at + render_projected, splat_trainer/scene/mlp_scene.py:410-427 /
 * scripts/test_split.py:30.)  The same launches as the single entry points above, chained with the two data-dependent
 * sizes of a frame left on the device: kernels run over the N scene rows and stop at the visible count M (counts[0]),
 * pair buffers hold `pair_capacity` pairs and every kernel reads the pair count O (counts[1]) on the device.  The host
 * receives counts[0..2] = [M, O, overflow flag] from the middle of the chain; when O > pair_capacity the frame is run again with a
 * larger capacity (nothing in the results depends on the capacity or on N as a bound). */
typedef struct GsrFrameC {
  const float* position;        /* [N,3] */
  const float* log_scaling;     /* [N,3] */
  const float* rotation_xyzw;   /* [N,4] */
  const float* alpha_logit;     /* [N]   */
  const float* sh_features;     /* [N,3,K] */
  int64_t N;
  int32_t K;                    /* 1, 4, 9 or 16 */
  int32_t W, H;
  const float* T_camera_world;  /* 16 floats */
  const float* projection;      /* fx fy cx cy */
  const float* camera_pos;      /* 3 floats */
  float near_plane, far_plane;
  GsrRasterParamsC params;
  int32_t want_jacobian;        /* save d colour / d position [M,9] for the backward pass */
  int32_t want_median;
  int32_t compute_visibility;
  int32_t needs_grad;           /* a backward pass will follow: per-pair visibility + checkpoints are kept */
  int32_t seg_pairs, seg_min_pairs;   /* RasterConfig.segment_pairs / segment_min_pairs */
  int64_t pair_capacity;
  /* Projected mode (position == NULL): the K4..K6 half of the three-call form, render_projected(indexes, gaussians2d,
   * features, depth, ...) (mlp_scene.py:418-419) -- the rows are packed from the caller's tensors, N is the exact number
   * of splats (nothing is culled here), C their feature channels, counts[0] is not used. */
  const float* gaussians2d;     /* [N,6] */
  const float* depth;           /* [N]   */
  const float* features;        /* [N,C] */
  int32_t C;                    /* 1..3 (3 in the one-call form) */
  const uint32_t* depth_order;  /* [N] or NULL: the depth order when the caller has it already (skips the depth sort) */
  /* One-call form only, all three or none: a second stream on which the depth sort (keys from the positions) runs
   * while the projection / colour sweep occupies `stream`; event_fork is recorded on `stream` behind the cull,
   * event_join on side_stream behind the sort (hipEvent_t, caller-owned, timing disabled). */
  void* side_stream;
  void* event_fork;
  void* event_join;
} GsrFrameC;
/* Byte offsets of the frame's buffers inside the two caller-owned arenas (-1: not present in this frame).  `out`:
 * everything the Rendering or the backward pass still needs after the forward pass; the first zero_bytes bytes are
 * zero-filled by the driver.  `work`: scratch that may be freed as soon as the forward pass has run. */
typedef struct GsrFramePlanC {
  int64_t out_bytes, work_bytes;
  int64_t zero_begin, zero_bytes;
  /* out arena */
  int64_t prune_cost, split_score, counts, tile_range, vis_partial, indexes, rows, screen_scale, jacobian, visibility,
      image, final_T, last, median, count, offsets, vals_a, vals_b, tvals_a, tvals_b, trank_a, trank_b, pair_vis,
      seg_tables, seg_pix, seg_last;
  int64_t seg_capacity, seg_heavy_capacity;
  /* work arena */
  int64_t cull_ws, sort_ws, scan_ws, tsort_ws, keys_a, keys_b, tile_hits, tkeys_a, tkeys_b;
  int64_t cull_ws_bytes, sort_ws_bytes, scan_ws_bytes, tsort_ws_bytes;
} GsrFramePlanC;
/* Where the ping-pong sorts left their results (byte offsets into `out`) and the segment tables of the frame. */
typedef struct GsrFrameResultC {
  int64_t order;          /* [N] u32: splat id by depth rank (first M valid) */
  int64_t sorted_inst;    /* [pair_capacity] u32, or -1 when pair_capacity == 0 */
  int64_t sorted_splat;   /* [pair_capacity] u32 */
  GsrSegmentsC segments;  /* device pointers into `out`; valid when has_segments */
  int32_t has_segments;
} GsrFrameResultC;
int gsr_frame_plan(const GsrFrameC* frame_host, GsrFramePlanC* plan_out_host);
/* counts_host (pinned host memory, 3 words; may be NULL) receives [M, O, overflow] by an asynchronous copy issued right
 * behind the scan that finalises them -- the rest of the chain is enqueued behind it -- and event_counts (hipEvent_t, may
 * be NULL) is recorded after that copy: the caller waits on the event, not on the stream.
 * event_k6_begin / event_k6_end: hipEvent_t recorded around the composite launch (bench.py's roofline leg), or NULL. */
int gsr_frame_forward(const GsrFrameC* frame_host, const GsrFramePlanC* plan_host, void* out_arena, void* work_arena,
                      GsrFrameResultC* result_out_host, uint32_t* counts_host, void* event_counts, void* event_k6_begin,
                      void* event_k6_end, void* stream);

/* The backward half of the same frame behind one call (loss.backward() through the node of scene.render,
 * splat_trainer/trainer/trainer.py:512): K7 -> gsr_reduce_gradients -> (gsr_inverse_map) -> gsr_project_backward_rows ->
 * gsr_sh_backward(_dense), with the arguments those entry points take.  All buffers are the caller's. */
typedef struct GsrFrameBackwardC {
  /* scene and camera of the forward call */
  const float* position; const float* log_scaling; const float* rotation_xyzw; const float* alpha_logit;
  const float* sh_features;       /* [N,3,K]; only read when sh_mode != 0 */
  int64_t N; int32_t K; int32_t W, H, C;
  const float* T_camera_world; const float* projection; const float* camera_pos;
  GsrRasterParamsC params;
  /* what the forward pass left (gsr_frame_forward's out arena) */
  int64_t M, O;
  const int64_t* indexes; const float* rows; const uint32_t* order; const uint32_t* count; const uint32_t* offsets;
  const uint32_t* sorted_splat; const uint32_t* sorted_inst; const float* pair_vis; const float* vis_partial;
  const uint32_t* tile_range; const float* final_T; const int32_t* last; const float* image;
  const float* jacobian;          /* [M,9] or NULL */
  const GsrSegmentsC* segments;   /* host pointer, or NULL */
  /* incoming gradients: d_image [H,W,C] (NULL: nothing reached the image), and what reached gaussians2d / depth */
  const float* d_image; const float* d_gaussians2d; const float* d_depth;
  /* scratch */
  float* partial;                 /* [O, GSR_PARTIAL_FLOATS] */
  float* grad_rows;               /* [M, GSR_ROW_FLOATS] */
  int32_t* inverse;               /* [N]; needed when M < N and (mode == 2 or sh_mode == 1) */
  float* d_colors;                /* [M,3]; needed when sh_mode != 0 (may be given otherwise: the colour gradient) */
  /* outputs: geometry gradients [N,*] with gsr_project_backward_rows' mode; the SH coefficient gradient [N,3,K] with
   * sh_mode 0: none, 1: every row overwritten (gsr_sh_backward_dense), 2: rows of `indexes` accumulated; the per-point
   * outputs [M] (each may be NULL) */
  float* d_position; float* d_log_scaling; float* d_rotation; float* d_alpha_logit; int32_t mode;
  float* d_sh; int32_t sh_mode;
  float* prune_cost; float* split_score; float* visibility;
} GsrFrameBackwardC;
/* event_k7_begin / event_k7_end: hipEvent_t recorded around the composite backward launch, or NULL. */
int gsr_frame_backward(const GsrFrameBackwardC* backward_host, void* event_k7_begin, void* event_k7_end, void* stream);
/* The same in two halves: stages bit 0 = K7 + per-splat reduction (grad_rows complete behind it), bit 1 = the rest; 3 = all.
 * A data-parallel caller packs and starts exchanging the colour-gradient factors (grad_rows columns 8..10) in between. */
int gsr_frame_backward_stages(const GsrFrameBackwardC* backward_host, int32_t stages, void* event_k7_begin,
                              void* event_k7_end, void* stream);

/* ---- loss stage next to the path: fused SSIM, replaces the CUDA-only fused_ssim package
 *      the reference imports (splat_trainer/trainer/trainer.py:17,112,450-462; trainer/evaluation.py:7,42) ------------ */
size_t gsr_ssim_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);
/* Mean SSIM (11x11 Gaussian window, sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) of img1 vs img2, both
 * [B,C,H,W] with arbitrary element strides (4 x int64 on the host: batch, channel, row, column -- NCHW,
 * channels_last and (H,W,C) views all work).  crop = 0: mean over the whole map ("same"); crop = 5: mean over the map
 * without its 5-pixel border ("valid", what the reference uses).  mean_out: device float.  dm_* ([B,C,H,W]
 * contiguous, all three or none): derivative maps consumed by gsr_ssim_backward. */
int gsr_ssim_forward(const float* img1, const float* img2, const int64_t* strides1_host, const int64_t* strides2_host,
                     int32_t B, int32_t C, int32_t H, int32_t W, int32_t crop, float* mean_out, float* dm_dmu1,
                     float* dm_dm11, float* dm_dm12, void* workspace, size_t workspace_bytes, void* stream);
/* d_img1 (element strides strides_out_host) = grad_scale * d mean / d img1; grad_scale_dev: device float. */
int gsr_ssim_backward(const float* img1, const float* img2, const int64_t* strides1_host, const int64_t* strides2_host,
                      const int64_t* strides_out_host, int32_t B, int32_t C, int32_t H, int32_t W, const float* dm_dmu1,
                      const float* dm_dm11, const float* dm_dm12, const float* grad_scale_dev, float* d_img1,
                      void* stream);

/* Pixel losses of the same stage (splat_trainer/trainer/trainer.py:465-488: l1 / mse on the image clamped to [0, 1]):
 * loss_out[0] = mean(f(clamp(image, lo, hi) - target)), f = square (kind 0) or abs (kind 1), over n contiguous floats
 * (16-byte aligned); fixed-order sums.  The backward writes grad_scale_dev[0] * d loss / d image, zero where the clamp
 * is active (torch.clamp's rule). */
size_t gsr_pixel_loss_workspace_bytes(int64_t n);
int gsr_pixel_loss_forward(const float* image, const float* target, int64_t n, int32_t kind, float lo, float hi,
                           float* loss_out, void* workspace, size_t workspace_bytes, void* stream);
int gsr_pixel_loss_backward(const float* image, const float* target, int64_t n, int32_t kind, float lo, float hi,
                            const float* grad_scale_dev, float* d_image, void* stream);

/* ---- optimizer stage next to the path: sparse visibility-aware Adam / LaProp step on the
 *      rows a batch has seen; replaces the Taichi kernels behind taichi_splatting.optim.ParameterClass.step
 *      (splat_trainer/scene/mlp_scene.py:214-230, options :58-60, groups config/scene/mlp.yaml:8-14).  Arithmetic:
 *      oracle/optim_oracle.py. ------------------------------------------------------------------------------------ */
/* Once per step.  indexes [M] int64, unique; visibility [M] or NULL (weight 1, no normalisation; vis_avg unused);
 * step [N] and vis_avg [N] are per-point state updated in place; row_scale_out [M,4] feeds gsr_opt_step. */
int gsr_opt_point_weights(const int64_t* indexes, const float* visibility, int64_t M, float* step, float* vis_avg,
                          float beta1, float beta2, float vis_beta, float vis_smooth, int32_t bias_correction,
                          float* row_scale_out, void* stream);
/* Once per parameter group, in place on rows `indexes` of param / exp_avg [N,D] and exp_avg_sq ([N,D] for type 0,
 * [N] otherwise).  type: 0 scalar, 1 vector (one second moment per row), 2 local_vector (D = 3, basis [M,3,3]
 * row-major: gradient and update expressed in the splat's own axes).  algo: 0 Adam, 1 LaProp.  grad_clip <= 0: off. */
int gsr_opt_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* indexes,
                 const float* row_scale, const float* basis, int64_t M, int32_t D, int32_t type, int32_t algo, float lr,
                 float beta1, float beta2, float eps, float grad_clip, void* stream);
/* basis_out [M,3,3] row-major = R(normalize(rotation[i])) * diag(max(exp(log_scaling[i]), eps)), i = indexes[m] (NULL:
 * i = m): the frame local_vector groups step in (gaussians/split.py:16-20, mlp_scene.py:219-225). */
int gsr_point_basis(const float* log_scaling, const float* rotation_xyzw, const int64_t* indexes, int64_t M, float eps,
                    float* basis_out, void* stream);

/* ---- densify / prune support next to the path ------------------------------------------
 *      deterministic top-n mask replacing take_n = argsort(t)[:n] -> mask (splat_trainer/controller/target_controller.py:
 *      150-160) and fused keep-mask compaction + append of all per-point columns (scene.split_and_prune,
 *      splat_trainer/scene/mlp_scene.py:301-310; controller/point_state.py:76-110). -------------------------------- */
size_t gsr_select_workspace_bytes(int64_t N);
/* mask_out [N] bytes (0/1): the n smallest values (descending = 0) or the n largest (descending = 1); among equal
 * values the lowest indexes are taken (what a stable argsort gives); NaN orders above +inf, -0 equals +0. */
int gsr_select_n(const float* values, int64_t N, int64_t n, int32_t descending, uint8_t* mask_out, void* workspace,
                 size_t workspace_bytes, void* stream);

#define GSR_MAX_COLUMNS 32
typedef struct GsrColumnC {
  const void* src;       /* [N, width] 4-byte words: the column before compaction */
  void* dst;             /* [kept + n_tail, width]: kept rows in order, then the appended rows */
  const void* tail;      /* [n_tail, width] appended rows, or NULL: the appended rows are zero-filled */
  int32_t width_dwords;
} GsrColumnC;
size_t gsr_compact_workspace_bytes(int64_t N);
/* keep_mask [N] bytes.  block_offsets_out [ceil(N/256)] = kept rows before each block of 256 source rows;
 * kept_total_dev = number of kept rows (read it back to size the destination columns). */
int gsr_compact_offsets(const uint8_t* keep_mask, int64_t N, uint32_t* block_offsets_out, uint32_t* kept_total_dev,
                        void* workspace, size_t workspace_bytes, void* stream);
/* One launch moves every column: kept rows to dst[0, kept), tail rows (or zeros) to dst[kept, kept + n_tail). */
int gsr_compact_columns(const uint8_t* keep_mask, int64_t N, const uint32_t* block_offsets, int64_t kept_total,
                        int64_t n_tail, const GsrColumnC* columns_host, int32_t n_columns, void* stream);

/* PointState.add_rendering (splat_trainer/controller/point_state.py:34-50) for one camera, fused: for every row m of
 * the camera's list, with i = idx[m]:  max_scale_px[i] = max(., max over the scale_cols (1 or 2) columns of
 * screen_scale[m]);  points_in_view[i] += visibility[m] > 0;  visibility[i] += visibility[m];
 * split_score[i] = exp_lerp(split_alpha, ., split_score[m]);  prune_cost[i] = exp_lerp(prune_alpha, ., prune_cost[m]).
 * idx rows are unique (one camera; NULL = rows are the points 0..M-1); the state arrays have N entries (points_in_view
 * int16).  Each input group is optional: a NULL screen_scale / visibility / split_score / prune_cost leaves the
 * corresponding state untouched (the data-parallel exchange replays only the two EMAs per camera and reduces the rest).
 * visible_sum (may be NULL): a second N-sized accumulator that also receives += visibility on the same rows -- the
 * scene's own `visible` statistic (mlp_scene.py:244), so that it does not cost an index_add launch of its own. */
int gsr_point_state_add(const int64_t* idx, const float* screen_scale, int32_t scale_cols, const float* visibility,
                        const float* split_score, const float* prune_cost, int64_t M, float split_alpha,
                        float prune_alpha, float* state_prune_cost, float* state_split_score, float* state_max_scale_px,
                        int16_t* state_points_in_view, float* state_visibility, float* visible_sum, void* stream);

/* The reference's per-camera loss mix (trainer.py:448-488 without reg_loss) behind one call per direction:
 *   loss = w_l1 mean|x - t| + w_mse mean (x - t)^2 + w_ssim / levels * sum_l (1 - ssim_valid(pool^l x, pool^l t)),
 * x = clamp(image, lo, hi), pool = 2 x 2 average pooling (floor sizes), image / target [H, W, C] contiguous, C <= 4,
 * levels 1..4 (the coarsest level must be more than 10 pixels per side).  metrics_out (device floats): [loss, l1, mse,
 * ssim_0 .. ssim_{levels-1}].  The workspace carries the forward pass's state to gsr_msloss_backward, which writes
 * d_image = grad_scale_dev[0] * d loss / d image (zero where the clamp is active). */
size_t gsr_msloss_workspace_bytes(int32_t H, int32_t W, int32_t C, int32_t levels);
int gsr_msloss_forward(const float* image, const float* target, int32_t H, int32_t W, int32_t C, int32_t levels,
                       float w_l1, float w_mse, float w_ssim, float lo, float hi, float* metrics_out, void* workspace,
                       size_t workspace_bytes, void* stream);
int gsr_msloss_backward(const float* image, const float* target, int32_t H, int32_t W, int32_t C, int32_t levels,
                        float w_l1, float w_mse, float w_ssim, float lo, float hi, const float* grad_scale_dev,
                        void* workspace, size_t workspace_bytes, float* d_image, void* stream);

/* ---- data-parallel exchange helpers (no reference counterpart: the reference is single-GPU) ------------- */
/* One fixed-size block per camera, GSR_DP_BLOCK_FLOATS(N) = 6N + 3 floats: [0,3N) colour-gradient rows (0 where the
 * camera saw nothing), [3N,3N+3) camera position, [3N+3,4N+3) split_score and [4N+3,5N+3) prune_cost (NaN where
 * unseen), [5N+3,6N+3) larger screen-space sigma (0 where unseen).  The first 3N+3 floats are the per-camera input of
 * gsr_sh_backward_multi.  gsr_dp_pack fills one block from a camera's rows (idx NULL: rows are the points 0..M-1);
 * gsr_dp_replay applies the two order-dependent exp_lerp EMAs of PointState.add_rendering (point_state.py:49-50) for
 * all cameras in camera order (slots[c] = block index of camera c) and folds the screen-scale maximum (:37).
 * The two order-independent SUMS of the update ride along: gsr_dp_pack(visibility, visibility_sum, views_sum) adds this
 * camera's visibility and its "saw the point" count to two N-sized accumulators (the extra columns of the gradient
 * all-reduce); gsr_dp_replay(visibility_sum, views_sum, state_visibility, state_points_in_view) adds the all-reduced
 * sums to the state (:40,43).  Pass NULL for the group to leave it out. */
#define GSR_DP_BLOCK_FLOATS(N) (6 * (int64_t)(N) + 3)
int gsr_dp_pack(const int64_t* idx, const float* dL_dcolors, const float* split_score, const float* prune_cost,
                const float* screen_scale, int32_t scale_cols, const float* camera_pos, int64_t M, int64_t N,
                float* block_out, const float* visibility, float* visibility_sum, float* views_sum, void* stream);
int gsr_dp_replay(const float* blocks, int64_t stride, const int32_t* slots, int32_t num_cameras, int64_t N,
                  float split_alpha, float prune_alpha, float* state_split_score, float* state_prune_cost,
                  float* state_max_scale_px, const float* visibility_sum, const float* views_sum,
                  float* state_visibility, int16_t* state_points_in_view, void* stream);

/* Sharded form of the exchange (default since round 4).  L = ceil(N / num_ranks); rank r owns the points [r L, (r+1) L).
 * gsr_dp_pack_sharded fills, for ONE camera of this rank (slot of slots_per_rank): factors_out (3N + 3 floats: the
 * camera's all-gather block, colour-gradient rows + camera position -- the input of gsr_sh_backward_multi); its two
 * controller scores in the all-to-all SEND layout scores_out[dest rank][slot][field][L] (field 0 split_score, 1
 * prune_cost; NaN where unseen); the running maximum of the larger screen-space sigma over this rank's cameras in
 * scale_max[N] (finished by a MAX all-reduce); and the two sums as gsr_dp_pack does.  factors_out may be NULL (the block
 * was packed earlier by gsr_dp_pack_factors_rows).
 * gsr_dp_replay_slice applies the two exp_lerp EMAs (point_state.py:49-50) of ALL cameras of the batch in camera order
 * (camera c arrived from rank c % num_ranks in slot c / num_ranks: recv[src rank][slot][field][L]) to this rank's slice
 * of the state and leaves the slice's new values in slice_out[field][L], the all-gather send buffer.
 * gsr_dp_finish writes the gathered slices (gathered[rank][field][L]) back into the N-sized state and folds the reduced
 * maximum and sums (any of the two groups may be NULL). */
int gsr_dp_pack_sharded(const int64_t* idx, const float* dL_dcolors, const float* split_score, const float* prune_cost,
                        const float* screen_scale, int32_t scale_cols, const float* camera_pos, int64_t M, int64_t N,
                        int32_t num_ranks, int32_t slots_per_rank, int32_t slot, float* factors_out, float* scores_out,
                        float* scale_max, const float* visibility, float* visibility_sum, float* views_sum, void* stream);
/* factors_out (3N + 3) of one camera straight from its packed gradient rows [M,16] (columns 8..10 = d colour), i.e. before
 * the backward sweep has copied them out: what lets the factor all-gather start behind gsr_frame_backward_stages(.., 1, ..).
 * gsr_dp_pack_sharded(factors_out = NULL) then leaves the block alone. */
int gsr_dp_pack_factors_rows(const int64_t* idx, const float* grad_rows, const float* camera_pos, int64_t M, int64_t N,
                             float* factors_out, void* stream);
int gsr_dp_replay_slice(const float* recv, int32_t num_ranks, int32_t slots_per_rank, int64_t N, int32_t rank,
                        int32_t num_cameras, float split_alpha, float prune_alpha, const float* state_split_score,
                        const float* state_prune_cost, float* slice_out, void* stream);
int gsr_dp_finish(const float* gathered, int32_t num_ranks, int64_t N, float* state_split_score, float* state_prune_cost,
                  const float* scale_max, float* state_max_scale_px, const float* visibility_sum, const float* views_sum,
                  float* state_visibility, int16_t* state_points_in_view, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
