// Densify / prune support next to the path:
//
//   gsr_select_n ........ bool mask of the n smallest (or largest) of N floats, ties broken by ascending index -- the
//                         deterministic form of the reference's take_n = argsort(t)[:n] -> mask
//                         (splat_trainer/controller/target_controller.py:150-160), whose non-stable argsort leaves the
//                         choice among equal values (0 and inf in their thousands: unseen points, min_views masking)
//                         to the sort implementation.  4-pass 8-bit radix SELECT on order-preserving keys (no sort:
//                         5 reads of the array instead of a full argsort), then the equal-to-threshold entries are
//                         ranked in index order with a ballot prefix.
//   gsr_compact_* ....... keep-mask stream compaction of ALL per-point columns at once (parameters, extras, optimizer
//                         state) + append of the split children (scene.split_and_prune, scene/mlp_scene.py:301-310:
//                         points[keep_mask] then append_tensors(splits)): per-block keep counts, one scan, then one
//                         gather kernel that moves every column's kept rows to their final place, copies the appended
//                         rows behind them and zero-fills the columns that have no appended data (new optimizer state).
//
//   gsr_point_state_add . the controller's per-camera statistics update PointState.add_rendering
//                         (splat_trainer/controller/point_state.py:34-50: max / += / count / two exp_lerp EMAs over the
//                         rows a camera saw) as ONE launch instead of ~15 gather / scatter launches: it runs once per
//                         camera inside the training step, and once per camera of the WHOLE batch on every rank of a
//                         data-parallel job (the EMAs are order dependent, so every rank replays every camera).
//
// wave64 ballot / mbcnt ranks, LDS-staged row lists, integer atomics only (histogram counts): bit-reproducible.
#include "gsr_device.h"
#include "../../include/gsplat_hip.h"

namespace {

constexpr int DN_THREADS = 256;

inline unsigned dn_grid(int64_t n, int per_block) { return (unsigned)((n + per_block - 1) / per_block); }

// Order-preserving u32 key of a float: -0 folded onto +0 (equal under comparison), NaN above +inf (torch sorts it last).
__device__ __forceinline__ uint32_t select_key(float v, bool descending) {
  v = v + 0.0f;
  if (v != v) v = __uint_as_float(0x7fc00000u);
  const uint32_t b = __float_as_uint(v);
  const uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return descending ? ~k : k;
}

struct SelectState {      // device words shared by the passes
  uint32_t prefix;        // decided high digits of the threshold key
  uint32_t mask;          // which bits of `prefix` are decided
  uint32_t remaining;     // how many entries are still to be taken inside the current prefix class
  uint32_t pad;
};

__global__ __launch_bounds__(DN_THREADS) void select_hist_kernel(const float* __restrict__ values, int64_t N,
                                                                 int descending, int shift,
                                                                 const SelectState* __restrict__ st,
                                                                 uint32_t* __restrict__ hist) {
  __shared__ uint32_t s_hist[256];
  s_hist[threadIdx.x] = 0u;
  __syncthreads();
  const uint32_t prefix = st->prefix, mask = st->mask;
  for (int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x; i < N; i += (int64_t)gridDim.x * DN_THREADS) {
    const uint32_t k = select_key(values[i], descending != 0);
    if ((k & mask) == prefix) atomicAdd(&s_hist[(k >> shift) & 255u], 1u);
  }
  __syncthreads();
  const uint32_t c = s_hist[threadIdx.x];
  if (c) atomicAdd(&hist[threadIdx.x], c);
}

// One block: the digit whose bucket holds the `remaining`-th entry becomes part of the prefix; the histogram is cleared
// for the next pass.
__global__ __launch_bounds__(256) void select_pick_kernel(SelectState* __restrict__ st, uint32_t* __restrict__ hist,
                                                          int shift) {
  __shared__ uint32_t s_cum[256];
  const int d = (int)threadIdx.x;
  const uint32_t mine = hist[d];
  hist[d] = 0u;
  s_cum[d] = mine;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {                               // inclusive scan over the 256 digits
    const uint32_t add = d >= o ? s_cum[d - o] : 0u;
    __syncthreads();
    s_cum[d] += add;
    __syncthreads();
  }
  const uint32_t incl = s_cum[d], excl = incl - mine;
  const uint32_t want = st->remaining;                              // 1-based rank inside the class
  __syncthreads();
  if (mine != 0u && excl < want && want <= incl) {                  // exactly one digit satisfies this
    st->prefix |= (uint32_t)d << shift;
    st->mask |= 255u << shift;
    st->remaining = want - excl;
  }
}

__global__ __launch_bounds__(DN_THREADS) void select_equal_count_kernel(const float* __restrict__ values, int64_t N,
                                                                        int descending,
                                                                        const SelectState* __restrict__ st,
                                                                        uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t s_w[4];
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  const bool eq = i < N && select_key(values[i], descending != 0) == st->prefix;
  const uint64_t b = __ballot(eq);
  if (gsr_lane() == 0) s_w[threadIdx.x >> 6] = (uint32_t)__builtin_popcountll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

__global__ __launch_bounds__(DN_THREADS) void select_mask_kernel(const float* __restrict__ values, int64_t N,
                                                                 int descending, const SelectState* __restrict__ st,
                                                                 const uint32_t* __restrict__ block_offsets,
                                                                 uint8_t* __restrict__ mask_out) {
  __shared__ uint32_t s_w[4];
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  const uint32_t thr = st->prefix, take = st->remaining;
  const uint32_t k = i < N ? select_key(values[i], descending != 0) : 0xffffffffu;
  const bool eq = i < N && k == thr;
  const uint64_t b = __ballot(eq);
  const int wave = threadIdx.x >> 6;
  if (gsr_lane() == 0) s_w[wave] = (uint32_t)__builtin_popcountll(b);
  __syncthreads();
  uint32_t before = block_offsets[blockIdx.x] + (uint32_t)gsr_mbcnt(b);
  for (int w = 0; w < wave; ++w) before += s_w[w];
  if (i < N) mask_out[i] = (k < thr || (eq && before < take)) ? 1 : 0;   // equal entries: lowest indexes first
}

// ------------------------------------------------------------------------------------------------ compaction
__global__ __launch_bounds__(DN_THREADS) void compact_count_kernel(const uint8_t* __restrict__ keep, int64_t N,
                                                                   uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t s_w[4];
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  const uint64_t b = __ballot(i < N && keep[i] != 0);
  if (gsr_lane() == 0) s_w[threadIdx.x >> 6] = (uint32_t)__builtin_popcountll(b);
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}

struct ColumnSet {
  GsrColumnC col[GSR_MAX_COLUMNS];
  int n;
};

// Blocks [0, row_blocks): 256 source rows each -- kept rows ranked with ballot/mbcnt, their local ids staged in LDS,
// then every column streams its kept rows to dst[offset + rank] (writes contiguous across the block).
// Blocks [row_blocks, ...): 256 appended rows each -- copied from the column's tail, or zero-filled without one.
__global__ __launch_bounds__(DN_THREADS) void compact_gather_kernel(const uint8_t* __restrict__ keep, int64_t N,
                                                                    const uint32_t* __restrict__ block_offsets,
                                                                    unsigned row_blocks, int64_t kept_total,
                                                                    int64_t n_tail, ColumnSet cs) {
  typedef uint32_t word;
  if (blockIdx.x >= row_blocks) {
    const int64_t r0 = (int64_t)(blockIdx.x - row_blocks) * DN_THREADS;
    const int64_t rows = min((int64_t)DN_THREADS, n_tail - r0);
    for (int c = 0; c < cs.n; ++c) {
      const int w = cs.col[c].width_dwords;
      word* dst = reinterpret_cast<word*>(cs.col[c].dst) + (kept_total + r0) * w;
      const word* tail = reinterpret_cast<const word*>(cs.col[c].tail);
      const int64_t total = rows * w;
      if (tail) {
        tail += r0 * w;
        for (int64_t e = threadIdx.x; e < total; e += DN_THREADS) dst[e] = tail[e];
      } else {
        for (int64_t e = threadIdx.x; e < total; e += DN_THREADS) dst[e] = 0u;
      }
    }
    return;
  }
  __shared__ uint32_t s_w[4];
  __shared__ uint16_t s_src[DN_THREADS];
  const int64_t row0 = (int64_t)blockIdx.x * DN_THREADS;
  const int64_t i = row0 + threadIdx.x;
  const bool k = i < N && keep[i] != 0;
  const uint64_t b = __ballot(k);
  const int wave = threadIdx.x >> 6;
  if (gsr_lane() == 0) s_w[wave] = (uint32_t)__builtin_popcountll(b);
  __syncthreads();
  uint32_t rank = (uint32_t)gsr_mbcnt(b);
  for (int w = 0; w < wave; ++w) rank += s_w[w];
  const uint32_t kept = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  if (k) s_src[rank] = (uint16_t)threadIdx.x;
  __syncthreads();
  const int64_t base = block_offsets[blockIdx.x];
  for (int c = 0; c < cs.n; ++c) {
    const int w = cs.col[c].width_dwords;
    const word* src = reinterpret_cast<const word*>(cs.col[c].src) + row0 * w;
    word* dst = reinterpret_cast<word*>(cs.col[c].dst) + base * w;
    const uint32_t total = kept * (uint32_t)w;
    if (w == 1) {
      for (uint32_t e = threadIdx.x; e < total; e += DN_THREADS) dst[e] = src[s_src[e]];
    } else {
      for (uint32_t e = threadIdx.x; e < total; e += DN_THREADS) {
        const uint32_t r = e / (uint32_t)w, d = e - r * (uint32_t)w;
        dst[e] = src[(uint32_t)s_src[r] * (uint32_t)w + d];
      }
    }
  }
}

// exp_lerp(t, a, b) = max + log(lerp(exp(a - max), exp(b - max), t))   (splat_trainer/util/misc.py:57-59)
__device__ __forceinline__ float exp_lerp(float t, float a, float b) {
  const float m = fmaxf(a, b);
  const float ea = expf(a - m), eb = expf(b - m);
  return m + logf(ea + t * (eb - ea));
}

__global__ __launch_bounds__(DN_THREADS) void point_state_add_kernel(const int64_t* __restrict__ idx,
                                                                     const float* __restrict__ scale, int scale_cols,
                                                                     const float* __restrict__ vis,
                                                                     const float* __restrict__ split,
                                                                     const float* __restrict__ prune, int64_t M,
                                                                     float split_alpha, float prune_alpha,
                                                                     float* __restrict__ st_prune,
                                                                     float* __restrict__ st_split,
                                                                     float* __restrict__ st_scale,
                                                                     int16_t* __restrict__ st_views,
                                                                     float* __restrict__ st_vis,
                                                                     float* __restrict__ visible_sum) {
  const int64_t m = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (m >= M) return;
  const int64_t i = idx ? idx[m] : m;             // rows of one camera are unique: no two threads share a point
  if (scale) {                                    // every group of updates is optional (NULL input = leave that state alone)
    float sc = scale[m * scale_cols];
    if (scale_cols == 2) sc = fmaxf(sc, scale[m * 2 + 1]);
    st_scale[i] = fmaxf(st_scale[i], sc);
  }
  if (vis) {
    const float v = vis[m];
    if (v > 0.f) st_views[i] = (int16_t)(st_views[i] + 1);
    st_vis[i] += v;
    if (visible_sum) visible_sum[i] += v;         // the scene's own `visible` accumulator (mlp_scene.py:244), same rows
  }
  if (split) st_split[i] = exp_lerp(split_alpha, st_split[i], split[m]);
  if (prune) st_prune[i] = exp_lerp(prune_alpha, st_prune[i], prune[m]);
}


// ---- data-parallel exchange: one dense block per camera (distributed.py: CameraShardedStep) -----------------------
// Block layout (floats, N = points of the scene):  [0, 3N) colour-gradient rows r,g,b (0 where the camera saw nothing),
// [3N, 3N+3) camera position, [3N+3, 4N+3) split_score and [4N+3, 5N+3) prune_cost (NaN where unseen),
// [5N+3, 6N+3) larger screen-space sigma (0 where unseen).  Fixed size: no counts have to be exchanged first, so the
// exchange needs no host round trip; the first 3N+3 floats are what gsr_sh_backward_multi reads.
__global__ __launch_bounds__(DN_THREADS) void dp_fill_kernel(float* __restrict__ block, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (i >= 6 * N + 3) return;
  const bool score = i >= 3 * N + 3 && i < 5 * N + 3;
  block[i] = score ? __uint_as_float(0x7fc00000u) : 0.f;
}

__global__ __launch_bounds__(DN_THREADS) void dp_pack_kernel(const int64_t* __restrict__ idx,
                                                             const float* __restrict__ dcol,
                                                             const float* __restrict__ split,
                                                             const float* __restrict__ prune,
                                                             const float* __restrict__ scale, int scale_cols,
                                                             const float* __restrict__ campos, int64_t M, int64_t N,
                                                             float* __restrict__ block,
                                                             const float* __restrict__ vis,
                                                             float* __restrict__ vis_sum,
                                                             float* __restrict__ views_sum) {
  const int64_t m = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (m < 3) block[3 * N + m] = campos[m];
  if (m >= M) return;
  const int64_t i = idx ? idx[m] : m;
  if (vis) {                                        // this rank's share of the two per-point SUMS (rows unique per camera)
    const float v = vis[m];
    vis_sum[i] += v;
    if (v > 0.f) views_sum[i] += 1.f;
  }
  block[3 * i] = dcol[3 * m];
  block[3 * i + 1] = dcol[3 * m + 1];
  block[3 * i + 2] = dcol[3 * m + 2];
  block[3 * N + 3 + i] = split[m];
  block[4 * N + 3 + i] = prune[m];
  float sc = scale[m * scale_cols];
  if (scale_cols == 2) sc = fmaxf(sc, scale[m * 2 + 1]);
  block[5 * N + 3 + i] = sc;
}

// Every rank replays the order-dependent EMAs of ALL cameras in camera order (slots[c] = block of camera c) and folds
// the screen-scale maximum: one pass over the points instead of one launch per camera.
__global__ __launch_bounds__(DN_THREADS) void dp_replay_kernel(const float* __restrict__ blocks, int64_t stride,
                                                               const int32_t* __restrict__ slots, int num_cameras,
                                                               int64_t N, float split_alpha, float prune_alpha,
                                                               float* __restrict__ st_split,
                                                               float* __restrict__ st_prune,
                                                               float* __restrict__ st_scale,
                                                               const float* __restrict__ vis_sum,
                                                               const float* __restrict__ views_sum,
                                                               float* __restrict__ st_vis,
                                                               int16_t* __restrict__ st_views) {
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (i >= N) return;
  if (vis_sum) {                                    // the two sums arrive all-reduced over the ranks
    st_vis[i] += vis_sum[i];
    st_views[i] = (int16_t)(st_views[i] + (int16_t)views_sum[i]);
  }
  float s = st_split[i], p = st_prune[i], mx = st_scale[i];
  for (int c = 0; c < num_cameras; ++c) {
    const float* b = blocks + (int64_t)slots[c] * stride;
    const float v = b[3 * N + 3 + i];
    if (v == v) {                                   // NaN: this camera did not see the point
      s = exp_lerp(split_alpha, s, v);
      p = exp_lerp(prune_alpha, p, b[4 * N + 3 + i]);
    }
    mx = fmaxf(mx, b[5 * N + 3 + i]);
  }
  st_split[i] = s;
  st_prune[i] = p;
  st_scale[i] = mx;
}

// ---- sharded exchange (round 4): the colour-gradient factors still reach every rank (every rank rebuilds the SH gradient
// of ALL points), but the two order-dependent controller scores only have to be REPLAYED in camera order, not on every rank:
// rank r replays the points [r L, (r + 1) L) for all cameras and the 2-float state of its slice is all-gathered afterwards.
// dp_pack_sharded_kernel writes this camera's scores straight into the all-to-all send layout [dest rank][slot][field][L].
__global__ __launch_bounds__(DN_THREADS) void dp_fill_sharded_kernel(float* __restrict__ factors, float* __restrict__ scores,
                                                                     int64_t N, int64_t L, int32_t cpr, int32_t slot) {
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (factors && i < 3 * N) factors[i] = 0.f;
  if (i < N) {
    const int64_t r = i / L, o = i - r * L;
    float* cell = scores + ((r * cpr + slot) * 2) * L + o;
    cell[0] = __uint_as_float(0x7fc00000u);
    cell[L] = __uint_as_float(0x7fc00000u);
  }
}

__global__ __launch_bounds__(DN_THREADS) void dp_pack_sharded_kernel(
    const int64_t* __restrict__ idx, const float* __restrict__ dcol, const float* __restrict__ split,
    const float* __restrict__ prune, const float* __restrict__ scale, int scale_cols, const float* __restrict__ campos,
    int64_t M, int64_t N, int64_t L, int32_t cpr, int32_t slot, float* __restrict__ factors, float* __restrict__ scores,
    float* __restrict__ scale_max, const float* __restrict__ vis, float* __restrict__ vis_sum,
    float* __restrict__ views_sum) {
  const int64_t m = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (factors && m < 3) factors[3 * N + m] = campos[m];
  if (m >= M) return;
  const int64_t i = idx ? idx[m] : m;
  if (vis) {                                        // this rank's share of the two per-point SUMS (rows unique per camera)
    const float v = vis[m];
    vis_sum[i] += v;
    if (v > 0.f) views_sum[i] += 1.f;
  }
  if (factors) {
    factors[3 * i] = dcol[3 * m];
    factors[3 * i + 1] = dcol[3 * m + 1];
    factors[3 * i + 2] = dcol[3 * m + 2];
  }
  const int64_t r = i / L, o = i - r * L;
  float* cell = scores + ((r * cpr + slot) * 2) * L + o;
  cell[0] = split[m];
  cell[L] = prune[m];
  float sc = scale[m * scale_cols];
  if (scale_cols == 2) sc = fmaxf(sc, scale[m * 2 + 1]);
  scale_max[i] = fmaxf(scale_max[i], sc);           // over this rank's cameras; a MAX all-reduce finishes it
}

// The factor block of one camera from its packed gradient rows (columns 8..10), before the sweep has copied them out.
__global__ __launch_bounds__(DN_THREADS) void dp_factors_rows_kernel(const int64_t* __restrict__ idx,
                                                                     const float* __restrict__ grows,
                                                                     const float* __restrict__ campos, int64_t M, int64_t N,
                                                                     float* __restrict__ factors, int fill) {
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (fill) {                                       // pass 1 (only when M < N): rows the camera did not see
    if (i < 3 * N) factors[i] = 0.f;
    return;
  }
  if (i < 3) factors[3 * N + i] = campos[i];
  if (i >= M) return;
  const int64_t r = idx ? idx[i] : i;
  const float4 g = *reinterpret_cast<const float4*>(grows + GSR_ROW_FLOATS * i + 8);
  factors[3 * r] = g.x; factors[3 * r + 1] = g.y; factors[3 * r + 2] = g.z;
}

// recv[q][s][f][o]: field f of camera (q + s G) at point lo + o, NaN where that camera did not see the point.
__global__ __launch_bounds__(DN_THREADS) void dp_replay_slice_kernel(const float* __restrict__ recv, int32_t G, int32_t cpr,
                                                                     int64_t L, int32_t num_cameras, int64_t lo,
                                                                     int64_t count, float split_alpha, float prune_alpha,
                                                                     const float* __restrict__ st_split,
                                                                     const float* __restrict__ st_prune,
                                                                     float* __restrict__ slice_out) {
  const int64_t o = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (o >= count) return;
  float s = st_split[lo + o], p = st_prune[lo + o];
  for (int c = 0; c < num_cameras; ++c) {
    const int64_t q = c % G, sl = c / G;
    const float* cell = recv + ((q * cpr + sl) * 2) * L + o;
    const float v = cell[0];
    if (v == v) {
      s = exp_lerp(split_alpha, s, v);
      p = exp_lerp(prune_alpha, p, cell[L]);
    }
  }
  slice_out[o] = s;
  slice_out[L + o] = p;
}

// gathered[r][f][o] -> state; the all-reduced maximum and sums folded in.
__global__ __launch_bounds__(DN_THREADS) void dp_finish_kernel(const float* __restrict__ gathered, int64_t L, int64_t N,
                                                               float* __restrict__ st_split, float* __restrict__ st_prune,
                                                               const float* __restrict__ scale_max,
                                                               float* __restrict__ st_scale,
                                                               const float* __restrict__ vis_sum,
                                                               const float* __restrict__ views_sum,
                                                               float* __restrict__ st_vis, int16_t* __restrict__ st_views) {
  const int64_t i = (int64_t)blockIdx.x * DN_THREADS + threadIdx.x;
  if (i >= N) return;
  const int64_t r = i / L, o = i - r * L;
  const float* cell = gathered + (r * 2) * L + o;
  st_split[i] = cell[0];
  st_prune[i] = cell[L];
  if (scale_max) st_scale[i] = fmaxf(st_scale[i], scale_max[i]);
  if (vis_sum) {
    st_vis[i] += vis_sum[i];
    st_views[i] = (int16_t)(st_views[i] + (int16_t)views_sum[i]);
  }
}

}  // namespace

extern "C" {

int gsr_point_state_add(const int64_t* idx, const float* screen_scale, int32_t scale_cols, const float* visibility,
                        const float* split_score, const float* prune_cost, int64_t M, float split_alpha,
                        float prune_alpha, float* state_prune_cost, float* state_split_score, float* state_max_scale_px,
                        int16_t* state_points_in_view, float* state_visibility, float* visible_sum, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || (scale_cols != 1 && scale_cols != 2)) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if ((screen_scale && !state_max_scale_px) || (visibility && (!state_points_in_view || !state_visibility)) ||
      (split_score && !state_split_score) || (prune_cost && !state_prune_cost))
    return GSR_ERR_INVALID_ARGUMENT;
  point_state_add_kernel<<<dn_grid(M, DN_THREADS), DN_THREADS, 0, stream>>>(
      idx, screen_scale, scale_cols, visibility, split_score, prune_cost, M, split_alpha, prune_alpha, state_prune_cost,
      state_split_score, state_max_scale_px, state_points_in_view, state_visibility, visibility ? visible_sum : nullptr);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_dp_pack(const int64_t* idx, const float* dL_dcolors, const float* split_score, const float* prune_cost,
                const float* screen_scale, int32_t scale_cols, const float* camera_pos, int64_t M, int64_t N,
                float* block_out, const float* visibility, float* visibility_sum, float* views_sum, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 3 || M > N || (scale_cols != 1 && scale_cols != 2) || !block_out || !camera_pos)
    return GSR_ERR_INVALID_ARGUMENT;
  if (visibility && (!visibility_sum || !views_sum)) return GSR_ERR_INVALID_ARGUMENT;
  if (M > 0 && (!dL_dcolors || !split_score || !prune_cost || !screen_scale)) return GSR_ERR_INVALID_ARGUMENT;
  if (M < N) {                                      // rows the camera did not see: zero gradient, NaN scores, zero scale
    dp_fill_kernel<<<dn_grid(6 * N + 3, DN_THREADS), DN_THREADS, 0, stream>>>(block_out, N);
    GSR_CHECK_LAUNCH();
  }
  dp_pack_kernel<<<dn_grid(M > 3 ? M : 3, DN_THREADS), DN_THREADS, 0, stream>>>(
      idx, dL_dcolors, split_score, prune_cost, screen_scale, scale_cols, camera_pos, M, N, block_out, visibility,
      visibility_sum, views_sum);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_dp_replay(const float* blocks, int64_t stride, const int32_t* slots, int32_t num_cameras, int64_t N,
                  float split_alpha, float prune_alpha, float* state_split_score, float* state_prune_cost,
                  float* state_max_scale_px, const float* visibility_sum, const float* views_sum,
                  float* state_visibility, int16_t* state_points_in_view, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || num_cameras < 0 || stride < 6 * N + 3) return GSR_ERR_INVALID_ARGUMENT;
  if (N == 0 || num_cameras == 0) return GSR_OK;
  if (!blocks || !slots || !state_split_score || !state_prune_cost || !state_max_scale_px) return GSR_ERR_INVALID_ARGUMENT;
  if (visibility_sum && (!views_sum || !state_visibility || !state_points_in_view)) return GSR_ERR_INVALID_ARGUMENT;
  dp_replay_kernel<<<dn_grid(N, DN_THREADS), DN_THREADS, 0, stream>>>(blocks, stride, slots, num_cameras, N, split_alpha,
                                                                     prune_alpha, state_split_score, state_prune_cost,
                                                                     state_max_scale_px, visibility_sum, views_sum,
                                                                     state_visibility, state_points_in_view);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_dp_pack_sharded(const int64_t* idx, const float* dL_dcolors, const float* split_score, const float* prune_cost,
                        const float* screen_scale, int32_t scale_cols, const float* camera_pos, int64_t M, int64_t N,
                        int32_t num_ranks, int32_t slots_per_rank, int32_t slot, float* factors_out, float* scores_out,
                        float* scale_max, const float* visibility, float* visibility_sum, float* views_sum, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 3 || M > N || (scale_cols != 1 && scale_cols != 2) || num_ranks < 1 || slots_per_rank < 1 || slot < 0 ||
      slot >= slots_per_rank || !scores_out || !scale_max || (factors_out && !camera_pos))
    return GSR_ERR_INVALID_ARGUMENT;
  if (visibility && (!visibility_sum || !views_sum)) return GSR_ERR_INVALID_ARGUMENT;
  if (M > 0 && ((factors_out && !dL_dcolors) || !split_score || !prune_cost || !screen_scale)) return GSR_ERR_INVALID_ARGUMENT;
  const int64_t L = (N + num_ranks - 1) / num_ranks;
  if (M < N) {                                      // rows the camera did not see: zero gradient, NaN scores
    dp_fill_sharded_kernel<<<dn_grid(factors_out ? 3 * N : N, DN_THREADS), DN_THREADS, 0, stream>>>(factors_out, scores_out, N, L,
                                                                                slots_per_rank, slot);
    GSR_CHECK_LAUNCH();
  }
  dp_pack_sharded_kernel<<<dn_grid(M > 3 ? M : 3, DN_THREADS), DN_THREADS, 0, stream>>>(
      idx, dL_dcolors, split_score, prune_cost, screen_scale, scale_cols, camera_pos, M, N, L, slots_per_rank, slot,
      factors_out, scores_out, scale_max, visibility, visibility_sum, views_sum);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_dp_pack_factors_rows(const int64_t* idx, const float* grad_rows, const float* camera_pos, int64_t M, int64_t N,
                             float* factors_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 3 || M > N || !factors_out || !camera_pos || (M > 0 && !grad_rows)) return GSR_ERR_INVALID_ARGUMENT;
  if (M < N) {
    dp_factors_rows_kernel<<<dn_grid(3 * N, DN_THREADS), DN_THREADS, 0, stream>>>(nullptr, nullptr, nullptr, M, N, factors_out, 1);
    GSR_CHECK_LAUNCH();
  }
  dp_factors_rows_kernel<<<dn_grid(M > 3 ? M : 3, DN_THREADS), DN_THREADS, 0, stream>>>(M < N ? idx : nullptr, grad_rows, camera_pos,
                                                                                      M, N, factors_out, 0);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_dp_replay_slice(const float* recv, int32_t num_ranks, int32_t slots_per_rank, int64_t N, int32_t rank,
                        int32_t num_cameras, float split_alpha, float prune_alpha, const float* state_split_score,
                        const float* state_prune_cost, float* slice_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || num_ranks < 1 || slots_per_rank < 1 || rank < 0 || rank >= num_ranks || num_cameras < 0 ||
      num_cameras > num_ranks * slots_per_rank)
    return GSR_ERR_INVALID_ARGUMENT;
  const int64_t L = (N + num_ranks - 1) / num_ranks;
  const int64_t lo = (int64_t)rank * L, count = lo < N ? (N - lo < L ? N - lo : L) : 0;
  if (count == 0) return GSR_OK;
  if (!recv || !state_split_score || !state_prune_cost || !slice_out) return GSR_ERR_INVALID_ARGUMENT;
  dp_replay_slice_kernel<<<dn_grid(count, DN_THREADS), DN_THREADS, 0, stream>>>(recv, num_ranks, slots_per_rank, L,
                                                                               num_cameras, lo, count, split_alpha,
                                                                               prune_alpha, state_split_score,
                                                                               state_prune_cost, slice_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_dp_finish(const float* gathered, int32_t num_ranks, int64_t N, float* state_split_score, float* state_prune_cost,
                  const float* scale_max, float* state_max_scale_px, const float* visibility_sum, const float* views_sum,
                  float* state_visibility, int16_t* state_points_in_view, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || num_ranks < 1) return GSR_ERR_INVALID_ARGUMENT;
  if (N == 0) return GSR_OK;
  if (!gathered || !state_split_score || !state_prune_cost) return GSR_ERR_INVALID_ARGUMENT;
  if (scale_max && !state_max_scale_px) return GSR_ERR_INVALID_ARGUMENT;
  if (visibility_sum && (!views_sum || !state_visibility || !state_points_in_view)) return GSR_ERR_INVALID_ARGUMENT;
  const int64_t L = (N + num_ranks - 1) / num_ranks;
  dp_finish_kernel<<<dn_grid(N, DN_THREADS), DN_THREADS, 0, stream>>>(gathered, L, N, state_split_score, state_prune_cost,
                                                                     scale_max, state_max_scale_px, visibility_sum,
                                                                     views_sum, state_visibility, state_points_in_view);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

size_t gsr_select_workspace_bytes(int64_t N) {
  const size_t blocks = (size_t)((N > 0 ? N : 1) + DN_THREADS - 1) / DN_THREADS;
  // state (16 B) + histogram (1 KiB) + block counts + the scan's own workspace
  return 256 + 1024 + ((blocks * 4 + 255) / 256) * 256 + gsr_scan_workspace_bytes((int64_t)blocks);
}

int gsr_select_n(const float* values, int64_t N, int64_t n, int32_t descending, uint8_t* mask_out, void* workspace,
                 size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || n < 0 || N >= (1ll << 32)) return GSR_ERR_INVALID_ARGUMENT;
  if (N == 0) return GSR_OK;
  if (!values || !mask_out) return GSR_ERR_INVALID_ARGUMENT;
  if (n == 0 || n >= N) {
    if (hipMemsetAsync(mask_out, n == 0 ? 0 : 1, (size_t)N, stream) != hipSuccess) return GSR_ERR_LAUNCH_FAILED;
    return GSR_OK;
  }
  if (!workspace || workspace_bytes < gsr_select_workspace_bytes(N)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  uint8_t* ws = reinterpret_cast<uint8_t*>(workspace);
  SelectState* st = reinterpret_cast<SelectState*>(ws);
  uint32_t* hist = reinterpret_cast<uint32_t*>(ws + 256);
  uint32_t* block_counts = reinterpret_cast<uint32_t*>(ws + 256 + 1024);
  const unsigned blocks = dn_grid(N, DN_THREADS);
  uint8_t* scan_ws = ws + 256 + 1024 + (((size_t)blocks * 4 + 255) / 256) * 256;
  const size_t scan_bytes = gsr_scan_workspace_bytes((int64_t)blocks);

  if (hipMemsetAsync(ws, 0, 256 + 1024, stream) != hipSuccess) return GSR_ERR_LAUNCH_FAILED;
  const uint32_t first = (uint32_t)n;
  if (hipMemcpyAsync(&st->remaining, &first, sizeof(uint32_t), hipMemcpyHostToDevice, stream) != hipSuccess)
    return GSR_ERR_LAUNCH_FAILED;
  const unsigned hist_grid = blocks < 2048u ? blocks : 2048u;
  for (int shift = 24; shift >= 0; shift -= 8) {
    select_hist_kernel<<<hist_grid, DN_THREADS, 0, stream>>>(values, N, descending, shift, st, hist);
    select_pick_kernel<<<1, 256, 0, stream>>>(st, hist, shift);
  }
  select_equal_count_kernel<<<blocks, DN_THREADS, 0, stream>>>(values, N, descending, st, block_counts);
  GSR_CHECK_LAUNCH();
  int rc = gsr_exclusive_scan_u32(block_counts, block_counts, (int64_t)blocks, nullptr, scan_ws, scan_bytes, stream_);
  if (rc != GSR_OK) return rc;
  select_mask_kernel<<<blocks, DN_THREADS, 0, stream>>>(values, N, descending, st, block_counts, mask_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

size_t gsr_compact_workspace_bytes(int64_t N) {
  const size_t blocks = (size_t)((N > 0 ? N : 1) + DN_THREADS - 1) / DN_THREADS;
  return ((blocks * 4 + 255) / 256) * 256 + gsr_scan_workspace_bytes((int64_t)blocks) + 256;
}

int gsr_compact_offsets(const uint8_t* keep_mask, int64_t N, uint32_t* block_offsets_out, uint32_t* kept_total_dev,
                        void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || N >= (1ll << 32) || !kept_total_dev) return GSR_ERR_INVALID_ARGUMENT;
  if (N == 0) {
    if (hipMemsetAsync(kept_total_dev, 0, sizeof(uint32_t), stream) != hipSuccess) return GSR_ERR_LAUNCH_FAILED;
    return GSR_OK;
  }
  if (!keep_mask || !block_offsets_out) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned blocks = dn_grid(N, DN_THREADS);
  if (!workspace || workspace_bytes < gsr_scan_workspace_bytes((int64_t)blocks)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  compact_count_kernel<<<blocks, DN_THREADS, 0, stream>>>(keep_mask, N, block_offsets_out);
  GSR_CHECK_LAUNCH();
  return gsr_exclusive_scan_u32(block_offsets_out, block_offsets_out, (int64_t)blocks, kept_total_dev, workspace,
                                workspace_bytes, stream_);
}

int gsr_compact_columns(const uint8_t* keep_mask, int64_t N, const uint32_t* block_offsets, int64_t kept_total,
                        int64_t n_tail, const GsrColumnC* columns_host, int32_t n_columns, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || kept_total < 0 || kept_total > N || n_tail < 0 || n_columns < 0 || n_columns > GSR_MAX_COLUMNS)
    return GSR_ERR_INVALID_ARGUMENT;
  if (n_columns == 0 || (N == 0 && n_tail == 0)) return GSR_OK;
  if (!columns_host || (N > 0 && (!keep_mask || !block_offsets))) return GSR_ERR_INVALID_ARGUMENT;
  ColumnSet cs;
  cs.n = n_columns;
  for (int c = 0; c < n_columns; ++c) {
    cs.col[c] = columns_host[c];
    if (cs.col[c].width_dwords <= 0 || !cs.col[c].dst || (N > 0 && !cs.col[c].src)) return GSR_ERR_INVALID_ARGUMENT;
  }
  const unsigned row_blocks = dn_grid(N, DN_THREADS), tail_blocks = dn_grid(n_tail, DN_THREADS);
  if (row_blocks + tail_blocks == 0) return GSR_OK;
  compact_gather_kernel<<<row_blocks + tail_blocks, DN_THREADS, 0, stream>>>(keep_mask, N, block_offsets, row_blocks,
                                                                            kept_total, n_tail, cs);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
