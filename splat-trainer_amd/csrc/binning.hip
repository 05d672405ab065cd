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

// A large extent (more than 32 tiles: splats tens to hundreds of pixels wide, which a training run grows sooner or later)
// is tested by the WHOLE WAVE, 64 tiles of the extent at a time in row-major order -- one thread walking the extent of a
// screen-filling splat (8160 tiles at 1080p) would hold its wave, and with it the launch, for a millisecond: measured on
// the c4 loop, tile_count 73 -> 1245 us and tile_emit 32 -> 1550 us per frame once a few splats had grown that large.
// All arguments are wave-uniform.  Returns the number of tiles hit (to every lane); with `keys` the hits are also written
// to the slots [o, o + n) in row-major order (ballot-ranked: exactly the order a sequential walk gives).
__device__ __forceinline__ uint32_t hits_of_large_extent(float u, float v, float A, float B, float C, float qmax, int x0,
                                                         int x1, int y0, int y1, uint32_t* keys, uint32_t* ranks,
                                                         uint32_t o, uint32_t capacity, uint32_t k, int tiles_x,
                                                         int lane) {
  const int nx = x1 - x0, total = nx * (y1 - y0);
  uint32_t n = 0;
  for (int t0 = 0; t0 < total; t0 += 64) {
    const int t = t0 + lane;
    uint32_t hm = 0u;
    int tx = 0, ty = 0;
    if (t < total) {
      ty = y0 + t / nx;
      tx = x0 + t % nx;
      if (gsr_tile_hit(u, v, A, B, C, qmax, tx, ty)) hm = gsr_tile_half_mask(u, v, A, B, C, qmax, tx, ty);
    }
    const uint64_t hit = __ballot(hm != 0u);
    if (keys && hm) {
      const uint32_t at = o + n + (uint32_t)gsr_mbcnt(hit);
      if (at < capacity) {                    // speculative launch into buffers sized from a guess: the caller re-emits
        keys[at] = (uint32_t)(ty * tiles_x + tx);
        ranks[at] = k | (hm << 30);           // splat id in the low 30 bits, the tile halves reached in the top 2
      }
    }
    n += (uint32_t)__builtin_popcountll(hit);
  }
  return n;
}

__device__ __forceinline__ float gsr_bcast(float x, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), l));
}

// The wave's large extents, one after the other: every lane that flagged one (`large`) has its splat's geometry
// broadcast and its tiles counted / emitted by all 64 lanes.  Must be reached by every lane of the wave.
__device__ __forceinline__ uint32_t wave_large_extents(bool large, float u, float v, float A, float B, float C,
                                                       const GsrExtent& e, uint32_t* keys, uint32_t* ranks, uint32_t o,
                                                       uint32_t capacity, uint32_t sid, int tiles_x, int lane) {
  uint32_t mine = 0u;
  uint64_t todo = __ballot(large);
  while (todo) {
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    const uint32_t n = hits_of_large_extent(
        gsr_bcast(u, l), gsr_bcast(v, l), gsr_bcast(A, l), gsr_bcast(B, l), gsr_bcast(C, l), gsr_bcast(e.qmax, l),
        __builtin_amdgcn_readlane(e.x0, l), __builtin_amdgcn_readlane(e.x1, l), __builtin_amdgcn_readlane(e.y0, l),
        __builtin_amdgcn_readlane(e.y1, l), keys, ranks, (uint32_t)__builtin_amdgcn_readlane((int)o, l), capacity,
        (uint32_t)__builtin_amdgcn_readlane((int)sid, l), tiles_x, lane);
    if (lane == l) mine = n;
  }
  return mine;
}

// One depth rank's tile count and hit record (see TileHits).  `live` false: a lane without a rank (it only takes part
// in the wave's cooperative passes).
__device__ __forceinline__ uint32_t tile_count_one(const float* __restrict__ rows, const uint32_t* __restrict__ order,
                                                   int64_t k, bool live, int tiles_x, int tiles_y,
                                                   const GsrRasterParams& rp, TileHits* __restrict__ hits) {
  float2 uv = make_float2(0.f, 0.f), ab = uv, co = uv;
  GsrExtent e;
  e.x0 = e.x1 = e.y0 = e.y1 = 0; e.qmax = 0.f;
  TileHits h;
  h.origin = 0u; h.shape = 0u; h.lo = 0u; h.hi = 0u;
  uint32_t n = 0;
  bool large = false;
  if (live) {
    const int64_t s = order[k];
    const float4* r = reinterpret_cast<const float4*>(rows + GSR_ROW_FLOATS * s);
    const float4 r0 = r[0];
    const float4 r1 = r[1];
    uv = make_float2(r0.x, r0.y); ab = make_float2(r0.z, r0.w); co = make_float2(r1.x, r1.y);
    e = gsr_splat_extent(uv.x, uv.y, ab.x, ab.y, co.x, co.y, rp, tiles_x, tiles_y);
    const int nx = e.x1 - e.x0, ny = e.y1 - e.y0;
    h.origin = (uint32_t)e.x0 | ((uint32_t)e.y0 << 16);
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
        large = true;
      }
    }
  }
  const uint32_t nl = wave_large_extents(large, uv.x, uv.y, ab.x, ab.y, co.x, e, nullptr, nullptr, 0u, 0u, 0u, tiles_x,
                                         gsr_lane());
  if (large) n = nl;
  if (live) *reinterpret_cast<uint4*>(hits + k) = make_uint4(h.origin, h.shape, h.lo, h.hi);
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
  // (M may be a capacity with the visible count still on the device: ranks behind it count zero tiles, so the scan over
  // the capacity needs no count of its own; every lane stays for the wave's cooperative pass over large extents)
  const bool live = k < M && !(M_dev && k >= (int64_t)*M_dev);
  uint32_t n = tile_count_one(rows, order, k, live, tiles_x, tiles_y, rp, hits);
  if (k < M) count[k] = n;
  if (block_sums) {
    // also leaves the block's total for offsets_from_sums_kernel (the scan's reduce pass, folded in here)
    __shared__ uint32_t s_wave[4];
    n = gsr_wave_sum_u32(n);
    if (gsr_lane() == 0) s_wave[threadIdx.x >> 6] = n;
    __syncthreads();
    if (threadIdx.x == 0) block_sums[blockIdx.x] = (s_wave[0] + s_wave[1]) + (s_wave[2] + s_wave[3]);
  }
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
  const bool live = k < M && !(M_dev && k >= (int64_t)*M_dev);
  uint4 hw = make_uint4(0u, 0u, 0u, 0u);
  uint32_t o = 0u, sid = 0u;
  if (live) {
    hw = *reinterpret_cast<const uint4*>(hits + k);
    o = offsets[k];
    sid = order[k];
  }
  // extents too large for the map: tested again, exactly as the count pass did, by the whole wave
  const bool large = (hw.y >> 16) != 0u;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
  GsrExtent e;
  e.x0 = e.x1 = e.y0 = e.y1 = 0; e.qmax = 0.f;
  if (large) {
    const float4* r = reinterpret_cast<const float4*>(rows + GSR_ROW_FLOATS * (int64_t)sid);
    r0 = r[0];
    r1 = r[1];
    e = gsr_splat_extent(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, rp, tiles_x, tiles_y);
  }
  wave_large_extents(large, r0.x, r0.y, r0.z, r0.w, r1.x, e, keys, inst2splat, o, capacity, sid, tiles_x, gsr_lane());
  if (!live || large) return;
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

// Instances of rank k occupy the contiguous pre-sort ids [offsets[k], offsets[k] + count[k]).  Their sum has ONE
// association order, whoever forms it (reduce_vis_kernel, reduce_grad_kernel: same bits; no float atomics anywhere on
// this path -> bit-reproducible results):
//   * up to GSR_REDUCE_SERIAL slots: added one after the other in id order;
//   * more (a splat that reaches more than 64 tiles): the slots are taken in groups of 64 counted from the rank's FIRST
//     slot, a group is summed over the wave by the fixed tree of gsr_wave_sum_to_lane63 (missing slots count 0), and the
//     group sums are added one after the other in group order.  One thread adding the 8160 slots of a screen-filling
//     splat held its block for 100+ us (c4 loop: reduce_grad 113 -> 670 us per frame once splats had grown).
#define GSR_REDUCE_SERIAL 64u

__global__ __launch_bounds__(256) void reduce_vis_kernel(const float* __restrict__ vis_partial,
                                                         const uint32_t* __restrict__ offsets,
                                                         const uint32_t* __restrict__ count,
                                                         const uint32_t* __restrict__ order, int64_t M,
                                                         float* __restrict__ vis, uint32_t limit,
                                                         const uint32_t* __restrict__ M_dev) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = k < M && !(M_dev && k >= (int64_t)*M_dev);
  const uint32_t b = live ? offsets[k] : 0u, n = live ? count[k] : 0u;
  const int lane = gsr_lane();
  float acc = 0.f;
  // limit = slots vis_partial holds: a speculative launch (buffers sized from a guess that turned out too small) must
  // not read past them; its output is discarded by the caller
  if (n <= GSR_REDUCE_SERIAL)
    for (uint32_t j = 0; j < n && b + j < limit; ++j) acc += vis_partial[b + j];
  uint64_t todo = __ballot(n > GSR_REDUCE_SERIAL);
  while (todo) {                                               // the wave's large ranks, one after the other
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)b, l), nl = (uint32_t)__builtin_amdgcn_readlane((int)n, l);
    float total = 0.f;                                         // (only lane 63's copy is meaningful)
    for (uint32_t g = 0; g < nl; g += 64u) {
      const uint32_t j = g + (uint32_t)lane;
      const float v = (j < nl && bl + j < limit) ? vis_partial[bl + j] : 0.f;
      total += gsr_wave_sum_to_lane63(v);
    }
    const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(total), 63));
    if (lane == l) acc = t;
  }
  if (live) vis[order ? order[k] : (uint32_t)k] = acc;
}

// One block = 256 consecutive depth ranks = one CONTIGUOUS range of instance slots.  The range is streamed through
// LDS in 256-slot chunks with fully coalesced 16-byte loads; each thread then adds the slots of its own rank from
// LDS in ascending id order (large ranks: by the whole wave afterwards, see above).
// The per-splat sums leave as ONE packed 64-byte row per splat, written whole by its thread (a full line: no partial
// write, no read-modify-write) at the splat's id -- the only crossing of the depth-order permutation on the backward
// side:   mx my mxx mxy | myy dop prune split | df0 df1 df2 visibility | 0 0 0 0   (m*: moments of G dL/dG about the
// mean, composite.hip K7; the sweep in splat order that reads the rows turns them into d(u, v, A, B, C)).
// The visibility column is the sum of the forward pass's per-pair partials in the same order reduce_vis_kernel uses
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
  const bool serial = n <= GSR_REDUCE_SERIAL;
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
    const uint32_t j0 = max(b, c), j1 = serial ? min(b + n, c + m) : 0u;
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
  // the wave's large ranks, one after the other, straight from global memory: lane i takes slot g + i of each group
  const int lane = gsr_lane();
  uint64_t todo = __ballot(!serial);
  while (todo) {
    const int l = __builtin_ctzll(todo);
    todo &= todo - 1ull;
    const uint32_t bl = (uint32_t)__builtin_amdgcn_readlane((int)b, l), nl = (uint32_t)__builtin_amdgcn_readlane((int)n, l);
    float t[12];                                               // (only lane 63's copies are meaningful)
#pragma unroll
    for (int i = 0; i < 12; ++i) t[i] = 0.f;
    for (uint32_t g = 0; g < nl; g += 64u) {
      const uint32_t j = g + (uint32_t)lane;
      const float pv = j < nl ? vis_partial[bl + j] : 0.f;
      float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0;
      if (pv > 0.f) {
        const float4* q = src + (size_t)3 * (bl + j);
        p0 = q[0]; p1 = q[1]; p2 = q[2];
      }
      t[0] += gsr_wave_sum_to_lane63(p0.x); t[1] += gsr_wave_sum_to_lane63(p0.y);
      t[2] += gsr_wave_sum_to_lane63(p0.z); t[3] += gsr_wave_sum_to_lane63(p0.w);
      t[4] += gsr_wave_sum_to_lane63(p1.x); t[5] += gsr_wave_sum_to_lane63(p1.y);
      t[6] += gsr_wave_sum_to_lane63(p1.z); t[7] += gsr_wave_sum_to_lane63(p1.w);
      t[8] += gsr_wave_sum_to_lane63(p2.x); t[9] += gsr_wave_sum_to_lane63(p2.y);
      t[10] += gsr_wave_sum_to_lane63(p2.z); t[11] += gsr_wave_sum_to_lane63(pv);
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) t[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(t[i]), 63));
    if (lane == l) {
      a0 = make_float4(t[0], t[1], t[2], t[3]);
      a1 = make_float4(t[4], t[5], t[6], t[7]);
      a2 = make_float4(t[8], t[9], t[10], 0.f);
      vsum = t[11];
    }
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
