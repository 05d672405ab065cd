// K4 tile overlap count + key emit, tile ranges, and the deterministic per-splat reductions of the
// per-(tile,splat) partials written by the composite kernels.
//
// Splats are visited in DEPTH ORDER (rank k -> splat order[k]); instances are emitted rank-major, so the
// instance stream is already depth-sorted and only a stable sort on the tile id is needed afterwards.
// Per-splat data crosses the depth-order permutation exactly twice per frame, one 64-byte line each way: the gather of
// the packed row in tile_count_kernel and the scatter of the packed gradient row in reduce_grad_kernel.
// The per-tile test is exact for the ellipse {d^T conic d <= qmax} against the rectangles of pixel centres of the
// tile's two halves, with qmax shrunk for faint splats (alpha can only reach 1/255 inside 2 ln(255 opacity)).
#include "gsr_device.h"
#include "../../include/gsplat_hip.h"

namespace {

inline GsrRasterParams to_params(const GsrRasterParamsC* c) {
  GsrRasterParams rp;
  __builtin_memcpy(&rp, c, sizeof(rp));
  return rp;
}
inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

__global__ __launch_bounds__(256) void depth_keys_kernel(const float* __restrict__ depth, int64_t M,
                                                         uint32_t* __restrict__ keys, uint32_t bias, uint32_t max_key) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  keys[m] = gsr_depth_key(depth[m], bias, max_key);
}

// Which tiles of its extent does a splat touch, and which halves of each?  A tile counts when the support reaches the
// pixel centres of its upper (bit 0) or lower (bit 1) half -- the two units the composite kernels evaluate; a support
// that only crosses the centre-free strip between the halves contributes to no pixel and makes no pair.
// K4 count stores the answer for the emit pass: extents of up to 32 tiles (all but splats hundreds of pixels wide) as a
// 64-bit map, 2 bits per tile of the extent in row-major order; larger ones are flagged and tested again by the emit.
struct TileHits {            // 16 bytes per depth rank
  uint32_t origin;           // x0 | y0 << 16
  uint32_t shape;            // nx | ny << 8 | (1 << 16 when the map does not fit: emit recomputes)
  uint32_t lo, hi;           // the map
};

__device__ __forceinline__ uint32_t hits_of_large_extent(float u, float v, float A, float B, float C, const GsrExtent& e,
                                                         uint32_t* keys, uint32_t* ranks, uint32_t o, uint32_t capacity,
                                                         uint32_t k, int tiles_x) {
  uint32_t n = 0;
  for (int ty = e.y0; ty < e.y1; ++ty)
    for (int tx = e.x0; tx < e.x1; ++tx) {
      if (!gsr_tile_hit(u, v, A, B, C, e.qmax, tx, ty)) continue;       // cheap rejection first: these extents are big
      const uint32_t hm = gsr_tile_half_mask(u, v, A, B, C, e.qmax, tx, ty);
      if (!hm) continue;
      if (keys) {
        if (o + n >= capacity) return n;      // speculative launch into buffers sized from a guess: the caller re-emits
        keys[o + n] = (uint32_t)(ty * tiles_x + tx);
        ranks[o + n] = k | (hm << 30);        // splat id in the low 30 bits, the tile halves reached in the top 2
      }
      ++n;
    }
  return n;
}

// One depth rank's tile count and hit record (see TileHits).
__device__ __forceinline__ uint32_t tile_count_one(const float* __restrict__ rows, const uint32_t* __restrict__ order,
                                                   int64_t k, int tiles_x, int tiles_y, const GsrRasterParams& rp,
                                                   TileHits* __restrict__ hits) {
  const int64_t s = order[k];
  const float4* r = reinterpret_cast<const float4*>(rows + GSR_ROW_FLOATS * s);
  const float4 r0 = r[0];
  const float4 r1 = r[1];
  const float2 uv = make_float2(r0.x, r0.y), ab = make_float2(r0.z, r0.w), co = make_float2(r1.x, r1.y);
  const GsrExtent e = gsr_splat_extent(uv.x, uv.y, ab.x, ab.y, co.x, co.y, rp, tiles_x, tiles_y);
  const int nx = e.x1 - e.x0, ny = e.y1 - e.y0;
  TileHits h;
  h.origin = (uint32_t)e.x0 | ((uint32_t)e.y0 << 16);
  h.shape = 0u; h.lo = 0u; h.hi = 0u;
  uint32_t n = 0;
  if (nx > 0 && ny > 0) {
    if (nx * ny <= 32) {
      h.shape = (uint32_t)nx | ((uint32_t)ny << 8);
      uint64_t map = 0ull;
      int t = 0;
      for (int ty = e.y0; ty < e.y1; ++ty)
        for (int tx = e.x0; tx < e.x1; ++tx, ++t) {
          const uint64_t hm = gsr_tile_half_mask(uv.x, uv.y, ab.x, ab.y, co.x, e.qmax, tx, ty);
          map |= hm << (2 * t);
          n += hm ? 1u : 0u;
        }
      h.lo = (uint32_t)map; h.hi = (uint32_t)(map >> 32);
    } else {
      h.shape = 1u << 16;
      n = hits_of_large_extent(uv.x, uv.y, ab.x, ab.y, co.x, e, nullptr, nullptr, 0u, 0u, 0u, tiles_x);
    }
  }
  *reinterpret_cast<uint4*>(hits + k) = make_uint4(h.origin, h.shape, h.lo, h.hi);
  return n;
}

// Thread k = depth rank k.  The only crossing of the depth-order permutation on the forward side: ONE gather of the
// first half of the splat's packed 64-byte row (geometry.hip: project_sh_fwd_kernel / pack_rows_kernel); the composite
// kernels later fetch the rows by splat id through the scalar cache, so no depth-ordered copy of the records exists.
__global__ __launch_bounds__(256) void tile_count_kernel(const float* __restrict__ rows,
                                                         const uint32_t* __restrict__ order, int64_t M, int tiles_x,
                                                         int tiles_y, GsrRasterParams rp, uint32_t* __restrict__ count,
                                                         TileHits* __restrict__ hits,
                                                         const uint32_t* __restrict__ M_dev,
                                                         uint32_t* __restrict__ block_sums) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (block_sums) {
    // also leaves the block's total for offsets_from_sums_kernel (the scan's reduce pass, folded in here)
    __shared__ uint32_t s_wave[4];
    uint32_t n = 0u;
    if (k < M) {
      if (!(M_dev && k >= (int64_t)*M_dev)) n = tile_count_one(rows, order, k, tiles_x, tiles_y, rp, hits);
      count[k] = n;
    }
    n = gsr_wave_sum_u32(n);
    if (gsr_lane() == 0) s_wave[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
    return;
  }
  if (k >= M) return;
  if (M_dev && k >= (int64_t)*M_dev) {        // M is a capacity: the visible count is still on the device
    count[k] = 0u;                            // the scan over the capacity then needs no count of its own
    return;
  }
  count[k] = tile_count_one(rows, order, k, tiles_x, tiles_y, rp, hits);
}

// offsets = exclusive scan of count, from the per-256 totals tile_count_kernel left: a block scans 4096 counts and adds
// up the totals in front of them itself (16 per block of this launch: a few thousand words at millions of splats).
// Raises *overflow when the pair count reaches 2^31 (float shadow sums: range, not precision -- as prims.hip does).
__global__ __launch_bounds__(256) void offsets_from_sums_kernel(const uint32_t* __restrict__ count, uint32_t n,
                                                                const uint32_t* __restrict__ sums256,
                                                                uint32_t* __restrict__ offsets,
                                                                uint32_t* __restrict__ total_out,
                                                                uint32_t* __restrict__ overflow) {
  __shared__ uint32_t s_wave[4], s_before[4];
  __shared__ float s_f[4];
  const int lane = gsr_lane(), w = (int)(threadIdx.x >> 6);
  const uint32_t base = blockIdx.x * 4096u + threadIdx.x * 16u;
  uint32_t v[16], sum = 0u;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    v[i] = (base + i < n) ? count[base + i] : 0u;
    sum += v[i];
  }
  uint32_t before = 0u;
  float fbefore = 0.f;
  for (uint32_t j = threadIdx.x; j < blockIdx.x * 16u; j += 256u) {
    const uint32_t b = sums256[j];
    before += b;
    fbefore += (float)b;
  }
  const uint32_t incl = gsr_wave_scan_incl_u32(sum);
  before = gsr_wave_sum_u32(before);
  fbefore = gsr_wave_sum(fbefore + (float)sum);
  if (lane == 63) s_wave[w] = incl;
  if (lane == 0) { s_before[w] = before; s_f[w] = fbefore; }
  __syncthreads();
  uint32_t prefix = (s_before[0] + s_before[1]) + (s_before[2] + s_before[3]), total = 0u;
  uint32_t in_block = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (j < w) in_block += s_wave[j];
    total += s_wave[j];
  }
  if (threadIdx.x == 0 && (s_f[0] + s_f[1]) + (s_f[2] + s_f[3]) >= 2147000000.f) *overflow = 1u;
  uint32_t run = prefix + in_block + incl - sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    if (base + i < n) offsets[base + i] = run;
    run += v[i];
  }
  if (total_out && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = prefix + total;
}


// Instances are emitted rank-major (thread k = depth rank k writes the slots [offsets[k], offsets[k] + count[k])), so the
// stream is already depth-sorted and a stable sort on the tile id finishes the per-tile lists; the VALUE an instance
// carries is the splat's id (row of the packed table), which is what the composite kernels fetch.
__global__ __launch_bounds__(256) void tile_emit_kernel(const float* __restrict__ rows,
                                                        const uint32_t* __restrict__ order,
                                                        const uint32_t* __restrict__ offsets,
                                                        const TileHits* __restrict__ hits, int64_t M, int tiles_x,
                                                        int tiles_y, GsrRasterParams rp, uint32_t* __restrict__ keys,
                                                        uint32_t* __restrict__ inst2splat, uint32_t capacity,
                                                        const uint32_t* __restrict__ M_dev) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M || (M_dev && k >= (int64_t)*M_dev)) return;
  const uint4 hw = *reinterpret_cast<const uint4*>(hits + k);
  uint32_t o = offsets[k];
  const uint32_t sid = order[k];
  if (hw.y >> 16) {                           // extent too large for the map: test again, exactly as the count pass did
    const float4* r = reinterpret_cast<const float4*>(rows + GSR_ROW_FLOATS * (int64_t)sid);
    const float4 r0 = r[0];
    const float4 r1 = r[1];
    const GsrExtent e = gsr_splat_extent(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rp, tiles_x, tiles_y);
    hits_of_large_extent(r0.x, r0.y, r0.z, r0.w, r1.x, e, keys, inst2splat, o, capacity, sid, tiles_x);
    return;
  }
  const int nx = (int)(hw.y & 0xFFu), ny = (int)((hw.y >> 8) & 0xFFu);
  const int x0 = (int)(hw.x & 0xFFFFu), y0 = (int)(hw.x >> 16);
  uint64_t map = (uint64_t)hw.z | ((uint64_t)hw.w << 32);
  for (int j = 0; j < ny; ++j)
    for (int i = 0; i < nx; ++i, map >>= 2) {
      const uint32_t hm = (uint32_t)(map & 3ull);
      if (!hm) continue;
      if (o >= capacity) return;              // speculative launch into buffers sized from a guess: the caller re-emits
      keys[o] = (uint32_t)((y0 + j) * tiles_x + x0 + i);
      inst2splat[o] = sid | (hm << 30);          // splat id in the low 30 bits, the tile halves reached in the top 2
      ++o;
    }
}

__global__ __launch_bounds__(256) void tile_ranges_kernel(const uint32_t* __restrict__ keys, int64_t O,
                                                          uint32_t* __restrict__ range,
                                                          const uint32_t* __restrict__ O_dev) {
  if (O_dev) O = min(O, (int64_t)*O_dev);     // O is a capacity: the pair count is still on the device
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= O) return;
  const uint32_t t = keys[i];
  if (i == 0 || keys[i - 1] != t) range[2 * t] = (uint32_t)i;
  if (i == O - 1 || keys[i + 1] != t) range[2 * t + 1] = (uint32_t)(i + 1);
}

// Instances of rank k occupy the contiguous pre-sort ids [offsets[k], offsets[k] + count[k]); summing them in id
// order gives a fixed association order -> bit-reproducible results (no float atomics anywhere on this path).
__global__ __launch_bounds__(256) void reduce_vis_kernel(const float* __restrict__ vis_partial,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ count,
                                                         const uint32_t* __restrict__ order, int64_t M,
                                                         float* __restrict__ vis, uint32_t limit,
                                                         const uint32_t* __restrict__ M_dev) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= M || (M_dev && k >= (int64_t)*M_dev)) return;
  const uint32_t b = offsets[k], n = count[k];
  float acc = 0.f;
  // limit = slots vis_partial holds: a speculative launch (buffers sized from a guess that turned out too small) must
  // not read past them; its output is discarded by the caller
  for (uint32_t j = 0; j < n && b + j < limit; ++j) acc += vis_partial[b + j];
  vis[order ? order[k] : (uint32_t)k] = acc;
}

// One block = 256 consecutive depth ranks = one CONTIGUOUS range of instance slots.  The range is streamed through
// LDS in 256-slot chunks with fully coalesced 16-byte loads; each thread then adds the slots of its own rank from
// LDS in ascending id order (fixed association order -> bit-reproducible).
// The per-splat sums leave as ONE packed 64-byte row per splat, written whole by its thread (a full line: no partial
// write, no read-modify-write) at the splat's id -- the only crossing of the depth-order permutation on the backward
// side:   mx my mxx mxy | myy dop prune split | df0 df1 df2 visibility | 0 0 0 0   (m*: moments of G dL/dG about the
// mean, composite.hip K7; the sweep in splat order that reads the rows turns them into d(u, v, A, B, C)).
// The visibility column is the sum of the forward pass's per-pair partials in the same id order reduce_vis_kernel uses
// (same bits), so a frame that is back-propagated needs no separate visibility reduction.
__global__ __launch_bounds__(256) void reduce_grad_kernel(const float* __restrict__ partial,
                                                          const float* __restrict__ vis_partial,
                                                          const uint32_t* __restrict__ offsets,
                                                          const uint32_t* __restrict__ count,
                                                          const uint32_t* __restrict__ order, int64_t M,
                                                          float* __restrict__ grows) {
  constexpr int CH = 256;
  __shared__ float4 s_part[CH * 3];
  __shared__ float s_vis[CH];
  __shared__ uint32_t s_lo, s_hi;
  const int64_t k0 = (int64_t)blockIdx.x * 256;
  const int64_t k = k0 + threadIdx.x;
  const bool have = k < M;
  const uint32_t b = have ? offsets[k] : 0u, n = have ? count[k] : 0u;
  if (threadIdx.x == 0) s_lo = b;
  const int64_t klast = (k0 + 255 < M ? k0 + 255 : M - 1);
  if (k == klast) s_hi = b + n;
  __syncthreads();
  const uint32_t lo = s_lo, hi = s_hi;
  float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0;
  float vsum = 0.f;
  const float4* src = reinterpret_cast<const float4*>(partial);
  for (uint32_t c = lo; c < hi; c += CH) {
    const uint32_t m = min((uint32_t)CH, hi - c);
    const bool mine = threadIdx.x < m;
    const float v = mine ? vis_partial[c + threadIdx.x] : 0.f;
    s_vis[threadIdx.x] = v;
    __syncthreads();
    // slots never written by the backward pass (vis == 0) hold garbage: neither fetched nor summed
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const uint32_t f = threadIdx.x + t * CH;                 // float4 index inside the chunk
      if (f < 3 * m && s_vis[f / 3] > 0.f) s_part[f] = src[(size_t)3 * c + f];
    }
    __syncthreads();
    const uint32_t j0 = max(b, c), j1 = min(b + n, c + m);
    for (uint32_t j = j0; j < j1; ++j) {
      const uint32_t l = j - c;
      const float pv = s_vis[l];
      vsum += pv;
      if (pv > 0.f) {
        const float4 p0 = s_part[3 * l], p1 = s_part[3 * l + 1], p2 = s_part[3 * l + 2];
        a0.x += p0.x; a0.y += p0.y; a0.z += p0.z; a0.w += p0.w;
        a1.x += p1.x; a1.y += p1.y; a1.z += p1.z; a1.w += p1.w;
        a2.x += p2.x; a2.y += p2.y; a2.z += p2.z;
      }
    }
    __syncthreads();
  }
  if (!have) return;
  const int64_t s = order ? (int64_t)order[k] : k;           // order == NULL: ranks are splat ids already
  float4* g = reinterpret_cast<float4*>(grows + GSR_ROW_FLOATS * s);
  g[0] = a0;
  g[1] = a1;
  g[2] = make_float4(a2.x, a2.y, a2.z, vsum);
  g[3] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// The packed gradient rows taken apart again for the three-call form (autograd hands d_gaussians2d / d_features on to the
// caller's own graph): one sequential sweep in splat order.
template <int C>
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float* __restrict__ rows,
                                                          const float* __restrict__ grows, int64_t M,
                                                          float* __restrict__ dg2d, float* __restrict__ dfeat,
                                                          float* __restrict__ prune, float* __restrict__ split,
                                                          float* __restrict__ vis) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float4* gr = reinterpret_cast<const float4*>(grows + GSR_ROW_FLOATS * m);
  const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2];
  // moments of G dL/dG about the mean -> d(u, v, A, B, C), with the conic of the forward row (composite.hip, K7)
  const float4* fr = reinterpret_cast<const float4*>(rows + GSR_ROW_FLOATS * m);
  const float4 f0 = fr[0];
  const float cA = f0.z, cB = f0.w, cC = fr[1].x;
  float* g = dg2d + 6 * m;
  *reinterpret_cast<float2*>(g) = make_float2(cA * g0.x + cB * g0.y, cB * g0.x + cC * g0.y);
  *reinterpret_cast<float2*>(g + 2) = make_float2(-0.5f * g0.z, -g0.w);
  const float op = fr[1].y;
  *reinterpret_cast<float2*>(g + 4) = make_float2(-0.5f * g1.x, op > 0.f ? g1.y / op : 0.f);
  if (prune) prune[m] = g1.z;
  if (split) split[m] = g1.w;
  if (vis) vis[m] = g2.w;
  dfeat[C * m] = g2.x;
  if (C > 1) dfeat[C * m + 1] = g2.y;
  if (C > 2) dfeat[C * m + 2] = g2.z;
}

}  // namespace

extern "C" {

int gsr_depth_key_range(float near_plane, float far_plane, uint32_t* bias_out, uint32_t* max_key_out) {
  if (!bias_out || !max_key_out) return GSR_ERR_INVALID_ARGUMENT;
  *bias_out = 0u;
  *max_key_out = 0xFFFFFFFFu;
  if (near_plane > 0.f && far_plane > near_plane && far_plane < 3.0e38f) {     // finite positive range: keys from 0
    uint32_t lo, hi;
    __builtin_memcpy(&lo, &near_plane, 4);
    __builtin_memcpy(&hi, &far_plane, 4);
    *bias_out = lo | 0x80000000u;
    *max_key_out = hi - lo;
  }
  return GSR_OK;
}

int gsr_depth_keys(const float* depth, int64_t M, uint32_t bias, uint32_t max_key, uint32_t* keys_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!depth || !keys_out) return GSR_ERR_INVALID_ARGUMENT;
  depth_keys_kernel<<<grid_for(M, 256), 256, 0, stream>>>(depth, M, keys_out, bias, max_key);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_tile_count(const float* rows, const uint32_t* order, int64_t M, int32_t W, int32_t H,
                   const GsrRasterParamsC* params_host, uint32_t* count_out, uint32_t* tile_hits_out,
                   const uint32_t* M_dev, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || M >= (1ll << 30) || !params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!rows || !order || !count_out || !tile_hits_out) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16;
  if (tx > 0xFFFF || ty > 0xFFFF) return GSR_ERR_UNSUPPORTED;      // the hit records hold 16-bit tile coordinates
  tile_count_kernel<<<grid_for(M, 256), 256, 0, stream>>>(rows, order, M, tx, ty, to_params(params_host), count_out,
                                                         reinterpret_cast<TileHits*>(tile_hits_out), M_dev, nullptr);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

size_t gsr_tile_count_offsets_workspace_bytes(int64_t M) {
  return M > 0 ? (size_t)((M + 255) / 256) * sizeof(uint32_t) + 256 : 256;
}

int gsr_tile_count_offsets(const float* rows, const uint32_t* order, int64_t M, int32_t W, int32_t H,
                           const GsrRasterParamsC* params_host, uint32_t* count_out, uint32_t* tile_hits_out,
                           const uint32_t* M_dev, uint32_t* offsets_out, uint32_t* total_dev, uint32_t* overflow_flag,
                           void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || M > GSR_TILE_COUNT_OFFSETS_MAX || !params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16) return GSR_ERR_UNSUPPORTED;
  if (!total_dev || !overflow_flag) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) { hipMemsetAsync(total_dev, 0, sizeof(uint32_t), stream); return GSR_OK; }
  if (!rows || !order || !count_out || !tile_hits_out || !offsets_out) return GSR_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < gsr_tile_count_offsets_workspace_bytes(M)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16;
  if (tx > 0xFFFF || ty > 0xFFFF) return GSR_ERR_UNSUPPORTED;
  uint32_t* sums = reinterpret_cast<uint32_t*>(workspace);
  tile_count_kernel<<<grid_for(M, 256), 256, 0, stream>>>(rows, order, M, tx, ty, to_params(params_host), count_out,
                                                         reinterpret_cast<TileHits*>(tile_hits_out), M_dev, sums);
  GSR_CHECK_LAUNCH();
  offsets_from_sums_kernel<<<grid_for(M, 4096), 256, 0, stream>>>(count_out, (uint32_t)M, sums, offsets_out, total_dev,
                                                                 overflow_flag);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_tile_emit(const float* rows, const uint32_t* order, const uint32_t* offsets, const uint32_t* tile_hits, int64_t M,
                  int32_t W, int32_t H, const GsrRasterParamsC* params_host, uint32_t* keys_out, uint32_t* inst2splat_out,
                  int64_t capacity, const uint32_t* M_dev, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || M >= (1ll << 30) || !params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (capacity < 0 || capacity > 0x7FFFFFFFll) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!rows || !order || !offsets || !tile_hits || !keys_out || !inst2splat_out) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16;
  tile_emit_kernel<<<grid_for(M, 256), 256, 0, stream>>>(rows, order, offsets, reinterpret_cast<const TileHits*>(tile_hits),
                                                        M, tx, ty, to_params(params_host), keys_out, inst2splat_out,
                                                        (uint32_t)capacity, M_dev);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_tile_ranges(const uint32_t* sorted_keys, int64_t O, int32_t num_tiles, uint32_t* tile_range,
                    const uint32_t* O_dev, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (O < 0 || num_tiles <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (O == 0) return GSR_OK;
  if (!sorted_keys || !tile_range) return GSR_ERR_INVALID_ARGUMENT;
  tile_ranges_kernel<<<grid_for(O, 256), 256, 0, stream>>>(sorted_keys, O, tile_range, O_dev);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_reduce_visibility(const float* vis_partial, const uint32_t* offsets, const uint32_t* count,
                          const uint32_t* order, int64_t M, float* visibility_out, int64_t capacity,
                          const uint32_t* M_dev, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || capacity < 0 || capacity > 0x7FFFFFFFll) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!offsets || !count || !visibility_out) return GSR_ERR_INVALID_ARGUMENT;
  reduce_vis_kernel<<<grid_for(M, 256), 256, 0, stream>>>(vis_partial, offsets, count, order, M, visibility_out,
                                                         (uint32_t)capacity, M_dev);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_reduce_gradients(const float* partial, const float* vis_partial, const uint32_t* offsets,
                         const uint32_t* count, const uint32_t* order, int64_t M, float* grad_rows_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!offsets || !count || !grad_rows_out) return GSR_ERR_INVALID_ARGUMENT;
  reduce_grad_kernel<<<grid_for(M, 256), 256, 0, stream>>>(partial, vis_partial, offsets, count, order, M, grad_rows_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_unpack_grad_rows(const float* rows, const float* grad_rows, int64_t M, int32_t C, float* d_gaussians2d,
                         float* d_features, float* prune_cost_out, float* split_score_out, float* visibility_out,
                         void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!rows || !grad_rows || !d_gaussians2d || !d_features) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(M, 256);
  if (C == 1) unpack_rows_kernel<1><<<g, 256, 0, stream>>>(rows, grad_rows, M, d_gaussians2d, d_features, prune_cost_out, split_score_out, visibility_out);
  else if (C == 2) unpack_rows_kernel<2><<<g, 256, 0, stream>>>(rows, grad_rows, M, d_gaussians2d, d_features, prune_cost_out, split_score_out, visibility_out);
  else unpack_rows_kernel<3><<<g, 256, 0, stream>>>(rows, grad_rows, M, d_gaussians2d, d_features, prune_cost_out, split_score_out, visibility_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
