// Frame driver: the forward half of one render_gaussians(use_sh=True) call behind ONE entry point.
//
//   K1 cull -> fused K2 + K3 -> depth sort -> K4 count -> scan -> K4 emit -> tile sort -> tile ranges -> segment plan
//   -> K6 (+ heavy-tile passes) [-> per-splat visibility]
//
// The chain is the same sequence of launches the Python host used to make one ctypes call at a time
// (splat-trainer_amd/renderer.py), with the two data-dependent sizes of a frame left ON THE DEVICE: the visible count M
// (kernels are launched for the N scene rows and stop at the device word) and the pair count O (buffers hold
// `pair_capacity` pairs; every kernel that needs O reads it from the device).  The host reads [M, O, overflow] back once,
// after everything has been enqueued -- nothing in the chain waits for a round trip -- and runs the frame again with a
// larger capacity when O turned out to exceed it.  Results do not depend on the capacity or on the bound N.
//
// No allocation here either: gsr_frame_plan lays the frame's buffers out in two caller-owned arenas (outputs that outlive
// the frame's backward pass / scratch) and returns their offsets; gsr_frame_forward enqueues.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "../../include/gsplat_hip.h"

namespace {

inline int64_t align_up(int64_t x, int64_t a = 256) { return (x + a - 1) / a * a; }

struct Cursor {
  int64_t at = 0;
  int64_t take(int64_t bytes) {
    const int64_t o = at;
    at = align_up(at + (bytes > 0 ? bytes : 1));     // an empty buffer still gets a range of its own
    return o;
  }
};

inline int tile_bits_of(int num_tiles) {
  int b = 1;
  while ((1 << b) < num_tiles) ++b;
  return b;
}

inline int bit_length(uint32_t v) {
  int b = 0;
  while (v) { ++b; v >>= 1; }
  return b < 1 ? 1 : b;
}

template <class T>
inline T* at(void* base, int64_t off) {
  return off < 0 ? nullptr : reinterpret_cast<T*>(reinterpret_cast<uint8_t*>(base) + off);
}

}  // namespace

extern "C" {

int gsr_frame_plan(const GsrFrameC* f, GsrFramePlanC* p) {
  if (!f || !p) return GSR_ERR_INVALID_ARGUMENT;
  if (f->N < 0 || f->N >= (1ll << 30) || f->W <= 0 || f->H <= 0 || f->pair_capacity < 0 || f->pair_capacity > 0x7FFFFFFFll)
    return GSR_ERR_INVALID_ARGUMENT;
  const bool projected = f->position == nullptr;
  if (!projected && f->K != 1 && f->K != 4 && f->K != 9 && f->K != 16) return GSR_ERR_UNSUPPORTED;
  if (f->C < 1 || f->C > 3 || (!projected && f->C != 3)) return GSR_ERR_UNSUPPORTED;
  if (f->params.tile_size != 16) return GSR_ERR_UNSUPPORTED;
  memset(p, 0, sizeof(*p));
  const int64_t N = f->N, cap = f->pair_capacity;
  const int64_t tx = (f->W + 15) / 16, ty = (f->H + 15) / 16, T = tx * ty, P = (int64_t)f->W * f->H;
  const bool vis_partial = f->compute_visibility || f->needs_grad;

  Cursor out;
  // one zero-filled region at the head of the output arena: everything that must start at zero
  p->zero_begin = 0;
  p->prune_cost = out.take(4 * N);
  p->split_score = out.take(4 * N);
  p->counts = out.take(4 * (3 + GSR_SEG_TOTAL_WORDS));   // [M, O, overflow flag, then the segment plan's counters]
  p->tile_range = out.take(4 * 2 * T);
  p->vis_partial = vis_partial ? out.take(4 * cap) : -1;
  p->zero_bytes = out.at;
  p->indexes = projected ? -1 : out.take(8 * N);
  p->rows = out.take(4 * GSR_ROW_FLOATS * N);
  p->screen_scale = out.take(4 * 2 * N);
  p->jacobian = (!projected && f->want_jacobian && f->K > 1) ? out.take(4 * 9 * N) : -1;
  p->visibility = out.take(4 * N);
  p->image = out.take(4 * f->C * P);
  p->final_T = out.take(4 * P);
  p->last = out.take(4 * P);
  p->median = f->want_median ? out.take(4 * P) : -1;
  // what the backward pass reads
  p->count = out.take(4 * N);
  p->offsets = out.take(4 * N);
  p->vals_a = out.take(4 * N);                 // depth order ends up in vals_a or vals_b
  p->vals_b = out.take(4 * N);
  p->tvals_a = out.take(4 * cap);              // sorted instance ids / splat ids end up in the a or the b set
  p->tvals_b = out.take(4 * cap);
  p->trank_a = out.take(4 * cap + 16);         // (+ 3 words: K6's large-frame walk fetches list words four at a time)
  p->trank_b = out.take(4 * cap + 16);
  p->pair_vis = vis_partial ? out.take(4 * cap) : -1;
  // segment tables and pixel slots (forward checkpoints read by the backward pass)
  p->seg_capacity = 0;
  p->seg_heavy_capacity = 0;
  p->seg_tables = p->seg_pix = p->seg_last = -1;
  if (f->seg_pairs != 0 && cap > 0) {
    const int64_t sc = gsr_segment_capacity(cap, 1, f->seg_pairs, f->seg_min_pairs, (int32_t)T, f->needs_grad ? 1 : 0);
    if (sc > 0) {
      int64_t hc = gsr_segment_heavy_capacity(cap, 1, f->seg_pairs, f->seg_min_pairs, (int32_t)T, f->needs_grad ? 1 : 0);
      if (hc > sc) hc = sc;
      p->seg_capacity = sc;
      p->seg_heavy_capacity = hc;
      p->seg_tables = out.take(4 * (2 * T + hc + 4 * sc + GSR_TILE_ORDER_WORDS(T)));
      const int64_t planes = 5 + (f->want_median ? 1 : 0);
      p->seg_pix = out.take(4 * planes * sc * 256);
      p->seg_last = out.take(4 * sc * 256);
    }
  }
  p->out_bytes = out.at;

  Cursor work;
  p->cull_ws_bytes = projected ? 0 : (int64_t)gsr_cull_workspace_bytes(N);
  p->sort_ws_bytes = (int64_t)gsr_sort_workspace_bytes(N);
  p->scan_ws_bytes = (int64_t)gsr_scan_workspace_bytes(N);
  if (N <= GSR_TILE_COUNT_OFFSETS_MAX && (int64_t)gsr_tile_count_offsets_workspace_bytes(N) > p->scan_ws_bytes)
    p->scan_ws_bytes = (int64_t)gsr_tile_count_offsets_workspace_bytes(N);
  p->tsort_ws_bytes = (int64_t)gsr_sort_workspace_bytes(cap);
  p->cull_ws = projected ? -1 : work.take(p->cull_ws_bytes);
  p->sort_ws = work.take(p->sort_ws_bytes);
  p->scan_ws = work.take(p->scan_ws_bytes);
  p->tsort_ws = work.take(p->tsort_ws_bytes);
  p->keys_a = work.take(4 * N);
  p->keys_b = work.take(4 * N);
  p->tile_hits = work.take(16 * N);
  p->tkeys_a = work.take(4 * cap);
  p->tkeys_b = work.take(4 * cap);
  p->work_bytes = work.at;
  return GSR_OK;
}

int gsr_frame_forward(const GsrFrameC* f, const GsrFramePlanC* p, void* out, void* work, GsrFrameResultC* res,
                      uint32_t* counts_host, void* event_counts, void* event_k6_begin, void* event_k6_end,
                      void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!f || !p || !out || !work || !res) return GSR_ERR_INVALID_ARGUMENT;
  if (f->N <= 0) return GSR_ERR_INVALID_ARGUMENT;          // an empty scene is the caller's blank frame
  const bool projected = f->position == nullptr;
  if (projected) {
    if (!f->gaussians2d || !f->depth || !f->features) return GSR_ERR_INVALID_ARGUMENT;
  } else if (!f->log_scaling || !f->rotation_xyzw || !f->alpha_logit || !f->sh_features || !f->T_camera_world ||
             !f->projection || !f->camera_pos) {
    return GSR_ERR_INVALID_ARGUMENT;
  }
  memset(res, 0, sizeof(*res));
  const int64_t N = f->N, cap = f->pair_capacity;
  const int tx = (f->W + 15) / 16, ty = (f->H + 15) / 16, T = tx * ty;
  uint32_t* counts = at<uint32_t>(out, p->counts);
  uint32_t* M_dev = projected ? nullptr : counts;          // projected mode: N is exact
  uint32_t* O_dev = counts + 1;
  int rc;
  bool forked = false;
#define GSR_TRY(call)          \
  do {                         \
    rc = (call);               \
    if (rc < 0) return rc;     \
  } while (0)

  if (hipMemsetAsync(at<uint8_t>(out, p->zero_begin), 0, (size_t)p->zero_bytes, stream) != hipSuccess)
    return GSR_ERR_LAUNCH_FAILED;
  float* rows = at<float>(out, p->rows);
  uint32_t key_bias = 0, key_max = 0;
  GSR_TRY(gsr_depth_key_range(f->near_plane, f->far_plane, &key_bias, &key_max));
  uint32_t* keys_a = at<uint32_t>(work, p->keys_a);
  if (projected) {
    GSR_TRY(gsr_pack_rows(f->gaussians2d, f->depth, f->features, N, f->C, &f->params, rows,
                          at<float>(out, p->screen_scale), stream_));
    if (!f->depth_order) GSR_TRY(gsr_depth_keys(f->depth, N, key_bias, key_max, keys_a, stream_));
  } else {
    int64_t* indexes = at<int64_t>(out, p->indexes);
    GSR_TRY(gsr_frustum_cull(f->position, N, f->T_camera_world, f->projection, f->W, f->H, f->near_plane, f->far_plane,
                             f->params.margin_px, indexes, M_dev, at<uint8_t>(work, p->cull_ws),
                             (size_t)p->cull_ws_bytes, stream_));
    forked = f->side_stream && f->event_fork && f->event_join;
    if (forked) {
      // the depth sort -- nine small latency-bound launches -- goes to the side stream with keys formed from the
      // positions alone, and runs while the fused K2 + K3 sweep streams the coefficient rows on this one
      hipStream_t side = reinterpret_cast<hipStream_t>(f->side_stream);
      if (hipEventRecord(reinterpret_cast<hipEvent_t>(f->event_fork), stream) != hipSuccess ||
          hipStreamWaitEvent(side, reinterpret_cast<hipEvent_t>(f->event_fork), 0) != hipSuccess)
        return GSR_ERR_LAUNCH_FAILED;
      GSR_TRY(gsr_depth_keys_from_positions(f->position, indexes, N, M_dev, f->T_camera_world, f->projection, key_bias,
                                            key_max, keys_a, f->side_stream));
    }
    GSR_TRY(gsr_project_sh_forward(f->position, f->log_scaling, f->rotation_xyzw, f->alpha_logit, f->sh_features, f->K,
                                   indexes, N, f->T_camera_world, f->projection, f->camera_pos, &f->params, rows,
                                   at<float>(out, p->screen_scale), at<float>(out, p->jacobian), M_dev,
                                   forked ? nullptr : keys_a, key_bias, key_max, stream_));
  }
  // depth order of the visible splats (stable: ties keep ascending index), keys written by the projection
  uint32_t* vals_a = at<uint32_t>(out, p->vals_a);
  uint32_t* vals_b = at<uint32_t>(out, p->vals_b);
  const uint32_t* order = f->depth_order;
  res->order = -1;                                          // the caller's own array
  if (!order) {
    GSR_TRY(gsr_sort_pairs_u32(keys_a, vals_a, at<uint32_t>(work, p->keys_b), vals_b, N, 1, 0, bit_length(key_max),
                               at<uint8_t>(work, p->sort_ws), (size_t)p->sort_ws_bytes, M_dev,
                               forked ? f->side_stream : stream_));
    res->order = rc == 1 ? p->vals_b : p->vals_a;
    order = rc == 1 ? vals_b : vals_a;
    if (forked && (hipEventRecord(reinterpret_cast<hipEvent_t>(f->event_join), reinterpret_cast<hipStream_t>(f->side_stream)) != hipSuccess ||
                   hipStreamWaitEvent(stream, reinterpret_cast<hipEvent_t>(f->event_join), 0) != hipSuccess))
      return GSR_ERR_LAUNCH_FAILED;
  }
  uint32_t* count = at<uint32_t>(out, p->count);
  uint32_t* offsets = at<uint32_t>(out, p->offsets);
  uint32_t* hits = at<uint32_t>(work, p->tile_hits);
  if (N <= GSR_TILE_COUNT_OFFSETS_MAX) {
    GSR_TRY(gsr_tile_count_offsets(rows, order, N, f->W, f->H, &f->params, count, hits, M_dev, offsets, O_dev, counts + 2,
                                   at<uint8_t>(work, p->scan_ws), (size_t)p->scan_ws_bytes, stream_));
  } else {
    GSR_TRY(gsr_tile_count(rows, order, N, f->W, f->H, &f->params, count, hits, M_dev, stream_));
    GSR_TRY(gsr_exclusive_scan_u32_checked(count, offsets, N, O_dev, counts + 2, at<uint8_t>(work, p->scan_ws),
                                           (size_t)p->scan_ws_bytes, stream_));
  }
  // [M, O, overflow] are final here: their copy to the host goes in NOW, in the middle of the chain, so that the host --
  // which needs them to shape the frame's tensors and goes on to enqueue the loss and the backward pass -- gets them
  // while the device still has the emit, the tile sort and the composite ahead of it
  if (counts_host) {
    if (hipMemcpyAsync(counts_host, counts, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
    if (event_counts && hipEventRecord(reinterpret_cast<hipEvent_t>(event_counts), stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
  }
  res->sorted_inst = res->sorted_splat = -1;
  float* image = at<float>(out, p->image);
  if (cap > 0) {
    uint32_t* tkeys_a = at<uint32_t>(work, p->tkeys_a);
    uint32_t* tkeys_b = at<uint32_t>(work, p->tkeys_b);
    uint32_t* tvals_a = at<uint32_t>(out, p->tvals_a);
    uint32_t* tvals_b = at<uint32_t>(out, p->tvals_b);
    uint32_t* trank_a = at<uint32_t>(out, p->trank_a);
    uint32_t* trank_b = at<uint32_t>(out, p->trank_b);
    GSR_TRY(gsr_tile_emit(rows, order, offsets, hits, N, f->W, f->H, &f->params, tkeys_a, trank_a, cap, M_dev, stream_));
    // values: instance id (implicit 0..O-1) and splat id (+ half mask) travel with the tile key
    GSR_TRY(gsr_sort_pairs2_u32(tkeys_a, tvals_a, trank_a, tkeys_b, tvals_b, trank_b, cap, 1, 0, tile_bits_of(T),
                                at<uint8_t>(work, p->tsort_ws), (size_t)p->tsort_ws_bytes, O_dev, stream_));
    const bool in_b = rc == 1;
    res->sorted_inst = in_b ? p->tvals_b : p->tvals_a;
    res->sorted_splat = in_b ? p->trank_b : p->trank_a;
    const uint32_t* sorted_keys = in_b ? tkeys_b : tkeys_a;
    const uint32_t* sorted_inst = in_b ? tvals_b : tvals_a;
    const uint32_t* sorted_splat = in_b ? trank_b : trank_a;
    uint32_t* tile_range = at<uint32_t>(out, p->tile_range);
    GSR_TRY(gsr_tile_ranges(sorted_keys, cap, T, tile_range, O_dev, stream_));
    const GsrSegmentsC* seg = nullptr;
    if (p->seg_capacity > 0) {
      uint32_t* tables = at<uint32_t>(out, p->seg_tables);
      uint32_t* tile_seg = tables;
      uint32_t* seg_desc = tables + 2 * (int64_t)T + p->seg_heavy_capacity;
      uint32_t* seg_total = counts + 3;
      uint32_t* tile_order = seg_desc + 4 * p->seg_capacity;
      GSR_TRY(gsr_segment_plan(tile_range, T, f->seg_pairs, f->seg_min_pairs, f->needs_grad ? 1 : 0, cap, O_dev,
                               p->seg_capacity, p->seg_heavy_capacity, tile_seg, seg_desc, seg_total, tile_order, stream_));
      float* pix = at<float>(out, p->seg_pix);
      res->segments.tile_seg = tile_seg;
      res->segments.seg_desc = seg_desc;
      res->segments.seg_total = seg_total;
      res->segments.tile_order = tile_order;
      res->segments.capacity = p->seg_capacity;
      res->segments.heavy_capacity = p->seg_heavy_capacity;
      res->segments.seg_TC = pix;
      res->segments.seg_P = pix + 4 * p->seg_capacity * 256;
      res->segments.seg_median = f->want_median ? pix + 5 * p->seg_capacity * 256 : nullptr;
      res->segments.seg_last = at<int32_t>(out, p->seg_last);
      res->has_segments = 1;
      seg = &res->segments;
    }
    if (event_k6_begin && hipEventRecord(reinterpret_cast<hipEvent_t>(event_k6_begin), stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
    GSR_TRY(gsr_composite_forward(rows, sorted_splat, sorted_inst, tile_range, f->W, f->H, f->C, &f->params, image,
                                  at<float>(out, p->final_T), at<int32_t>(out, p->last), at<float>(out, p->median),
                                  at<float>(out, p->vis_partial), at<float>(out, p->pair_vis), seg,
                                  N >= GSR_PREFETCH_MIN_ROWS ? 1 : 0, stream_));
    if (event_k6_end && hipEventRecord(reinterpret_cast<hipEvent_t>(event_k6_end), stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
    if (f->compute_visibility && !f->needs_grad)
      GSR_TRY(gsr_reduce_visibility(at<float>(out, p->vis_partial), offsets, count, order, N,
                                    at<float>(out, p->visibility), cap, M_dev, stream_));
  }
#undef GSR_TRY
  return GSR_OK;
}

// sizeof of the ABI's structs as this library was compiled (0 GsrRasterParamsC, 1 GsrSegmentsC, 2 GsrFrameC, 3 GsrFramePlanC,
// 4 GsrFrameResultC, 5 GsrFrameBackwardC; -1 otherwise): lets a binding check its own layout before the first call.
int64_t gsr_struct_bytes(int32_t which) {
  switch (which) {
    case 0: return (int64_t)sizeof(GsrRasterParamsC);
    case 1: return (int64_t)sizeof(GsrSegmentsC);
    case 2: return (int64_t)sizeof(GsrFrameC);
    case 3: return (int64_t)sizeof(GsrFramePlanC);
    case 4: return (int64_t)sizeof(GsrFrameResultC);
    case 5: return (int64_t)sizeof(GsrFrameBackwardC);
    default: return -1;
  }
}

// The backward half behind one call: K7 -> per-splat reduction -> (inverse map) -> geometry sweep -> SH coefficient
// gradient, the launches renderer._FrameFn.backward used to make one ctypes call at a time.
int gsr_frame_backward(const GsrFrameBackwardC* b, void* event_k7_begin, void* event_k7_end, void* stream_) {
  return gsr_frame_backward_stages(b, 3, event_k7_begin, event_k7_end, stream_);
}

// stages: bit 0 = K7 + the per-splat reduction (the packed gradient rows are complete behind it), bit 1 = everything
// after (inverse map, geometry sweep, SH coefficient gradient).  A data-parallel caller runs the two halves as two calls and
// starts its exchange of the colour-gradient factors -- columns 8..10 of the rows -- in between, so that the transfer runs
// next to the sweep (distributed.CameraShardedStep: early factor gather).  3 = gsr_frame_backward.
int gsr_frame_backward_stages(const GsrFrameBackwardC* b, int32_t stages, void* event_k7_begin, void* event_k7_end,
                              void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (stages < 1 || stages > 3) return GSR_ERR_INVALID_ARGUMENT;
  if (!b || b->N < 0 || b->M < 0 || b->M > b->N || b->O < 0 || b->mode < 0 || b->mode > 2 || b->sh_mode < 0 ||
      b->sh_mode > 2)
    return GSR_ERR_INVALID_ARGUMENT;
  const int64_t M = b->M, N = b->N;
  if (M > 0 && (!b->grad_rows || !b->rows || !b->indexes)) return GSR_ERR_INVALID_ARGUMENT;
  if (b->sh_mode && (!b->d_sh || !b->sh_features || !b->camera_pos || (M > 0 && !b->d_colors))) return GSR_ERR_INVALID_ARGUMENT;
  const bool needs_inverse = M < N && (b->mode == 2 || b->sh_mode == 1);
  if (needs_inverse && !b->inverse) return GSR_ERR_INVALID_ARGUMENT;
#define GSR_TRY(call)                 \
  do {                                \
    const int rc_ = (call);           \
    if (rc_ < 0) return rc_;          \
  } while (0)
  const bool live = M > 0 && b->O > 0 && b->d_image != nullptr;
  if (!(stages & 1)) {
    // (the rows were formed by an earlier call)
  } else if (live) {
    if (!b->partial || !b->vis_partial) return GSR_ERR_INVALID_ARGUMENT;
    if (event_k7_begin && hipEventRecord(reinterpret_cast<hipEvent_t>(event_k7_begin), stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
    GSR_TRY(gsr_composite_backward(b->rows, b->sorted_splat, b->sorted_inst, b->pair_vis, b->tile_range, b->W, b->H, b->C,
                                   &b->params, b->final_T, b->last, b->d_image, b->image, b->partial, b->segments, stream_));
    if (event_k7_end && hipEventRecord(reinterpret_cast<hipEvent_t>(event_k7_end), stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
    GSR_TRY(gsr_reduce_gradients(b->partial, b->vis_partial, b->offsets, b->count, b->order, M, b->grad_rows, stream_));
  } else if (M > 0) {
    if (hipMemsetAsync(b->grad_rows, 0, (size_t)M * GSR_ROW_FLOATS * sizeof(float), stream) != hipSuccess)
      return GSR_ERR_LAUNCH_FAILED;
  }
  if (!(stages & 2)) return GSR_OK;
  if (needs_inverse) GSR_TRY(gsr_inverse_map(b->indexes, M, N, b->inverse, stream_));
  if (M > 0 || b->mode == 2)
    GSR_TRY(gsr_project_backward_rows(b->position, b->log_scaling, b->rotation_xyzw, b->alpha_logit, b->indexes, M,
                                      b->mode == 2 ? b->inverse : nullptr, N, b->T_camera_world, b->projection, &b->params,
                                      b->rows, b->grad_rows, b->d_gaussians2d, b->d_depth, b->jacobian, b->d_position,
                                      b->d_log_scaling, b->d_rotation, b->d_alpha_logit, b->mode, b->d_colors,
                                      live ? b->prune_cost : nullptr, live ? b->split_score : nullptr,
                                      live ? b->visibility : nullptr, stream_));
  if (b->sh_mode == 1) {
    GSR_TRY(gsr_sh_backward_dense(b->d_colors, b->sh_features, b->position, b->inverse, M, N, b->K, b->camera_pos, nullptr,
                                  b->d_sh, nullptr, stream_));
  } else if (b->sh_mode == 2 && M > 0) {
    GSR_TRY(gsr_sh_backward(b->d_colors, b->sh_features, b->position, b->indexes, M, b->K, b->camera_pos, nullptr, b->d_sh,
                            nullptr, 1, stream_));
  }
#undef GSR_TRY
  return GSR_OK;
}

}  // extern "C"
