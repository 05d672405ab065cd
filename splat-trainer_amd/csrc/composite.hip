// K6 alpha-composite forward and K7 alpha-composite backward (per-pixel reverse walk).
//
// Mapping: ONE wave64 owns one 16x16 tile; lane l owns 4 pixels, one in each 8x8 quadrant of the tile, at
//   (8*(p&1) + (l&7), 8*(p>>1) + (l>>3)), p = 0..3.  A small splat touches 1-2 quadrants, and a quadrant no lane
//   needs is skipped by a wave-uniform branch (s_cbranch_execz), so the per-pixel work follows the footprint.
//   * the tile's depth-sorted splat list is walked with WAVE-UNIFORM addresses, so the 48-byte record of the
//     current splat arrives through the scalar data cache into SGPRs (s_load_dwordx4/x8): no VGPRs, no LDS, no
//     barrier; the next record is prefetched while the current one is evaluated;
//   * every per-splat reduction over the tile's 256 pixels is paid once per 256 pixels, not once per 64: the backward
//     pass parks its 11 per-pair sums in the wave's LDS block and 44 reader lanes add them up (the VALU is the binding
//     unit, the LDS pipe is idle otherwise); the forward pass folds the visibility of four pairs with permlane swaps
//     + DPP row steps;
//   * no float atomics: each (tile, splat) pair owns one slot of a partial buffer, written after the fixed-order
//     reduction and summed per splat in id order afterwards -> bit-reproducible gradients and densification
//     heuristics;
//   * the forward pass records, per (tile, splat) pair, the visibility partial sum_px T*alpha; a pair whose
//     partial is zero touched no pixel, so the backward pass skips it without evaluating a single pixel.
// Heavy tiles (list segmentation): one wave walks one tile's list serially, so a tile with thousands of pairs
//   would bound both kernels from below (clustered scenes).  gsr_segment_plan cuts the list of every tile longer than
//   `heavy_min` pairs into segments of <= `seg_pairs` pairs; such a tile is composited by
//     A) one wave per segment: per-pixel product of (1 - alpha) over the segment (extra blocks of the K6 launch),
//     C) one wave per segment: the ordinary forward walk, started from T_in = product of the preceding segments,
//     D) one wave per tile: colours summed in segment order, final T / last / median combined, and the colour BEHIND
//        each segment left in place of its own colour,
//   and back-propagated by one wave per segment (extra blocks of the K7 launch) that starts its reverse walk from
//   the segment's own end state (T after the segment, colour behind it).  Everything stays fixed-order: no atomics.
// MFMA is deliberately not used: there is no dense contraction on this path.
#include "gsr_device.h"
#include "gsr_dpp_reduce.h"
#include "../../include/gsplat_hip.h"

namespace {

inline GsrRasterParams to_params(const GsrRasterParamsC* c) {
  GsrRasterParams rp;
  __builtin_memcpy(&rp, c, sizeof(rp));
  return rp;
}

typedef float v2f __attribute__((ext_vector_type(2)));
// four consecutive list words at an address that is only 4-byte aligned (a wave-uniform address: one s_load_dwordx4)
struct __attribute__((packed, aligned(4))) Rank4 { uint32_t x, y, z, w; };
#define GSR_V2(x) ((v2f){(x), (x)})

// The VALU is the binding unit of both kernels (a wave64 fp32 op holds its SIMD for 4 cycles; only the packed
// v_pk_{mul,add,fma}_f32 forms reach the 64 FLOP/clk/SIMD peak), so each lane evaluates its two pixels of one
// tile half (left/right 8x8 quadrant, same row => same dy) as ONE 2-wide packed computation, branch-free, with
// non-contributing pixels masked to alpha = 0.  A half no lane needs is skipped by a wave-uniform branch.
//
// eval_q2 / eval_qt / eval_G2 are shared by forward and backward and pin every rounding (explicit fma, contraction
// off), so both passes take bit-identical contribute / skip decisions.
// GSR_QT_FORM 1 forms q as d . t with t = conic d (the backward pass needs t anyway: two packed operations instead of
// six there).  Measured (same box, c2 / c3): K7 374 / 724 us against 380 / 735, but K6 228 / 715 against 203 / 721 -- the
// forward kernel goes from 79 to 83 SGPRs and loses more on the frame the headline is quoted on than the backward pass
// gains; capping its SGPRs at 80 leaves it at 212 / 743.  Default: the direct quadratic form in both.
#ifndef GSR_QT_FORM
#define GSR_QT_FORM 0
#endif
__device__ __forceinline__ v2f eval_q2(v2f dx, float dy, float A, float B, float C) {
#pragma clang fp contract(off)
#if GSR_QT_FORM
  const float bdy = B * dy, cdy = C * dy;
  const v2f tx = __builtin_elementwise_fma(dx, GSR_V2(A), GSR_V2(bdy));
  const v2f ty = __builtin_elementwise_fma(dx, GSR_V2(B), GSR_V2(cdy));
  return __builtin_elementwise_fma(GSR_V2(dy), ty, dx * tx);
#else
  // q = A dx^2 + 2B dx dy + C dy^2
  const float t = (B + B) * dy;
  const float u = (C * dy) * dy;
  return __builtin_elementwise_fma(dx * A, dx, __builtin_elementwise_fma(GSR_V2(t), dx, GSR_V2(u)));
#endif
}
// ... and t = conic d = (A dx + B dy, B dx + C dy) next to it (backward: gradient of the mean, split score)
__device__ __forceinline__ void eval_qt(v2f dx, float dy, float A, float B, float C, v2f& q, v2f& tx, v2f& ty) {
#pragma clang fp contract(off)
  const float bdy = B * dy, cdy = C * dy;
  tx = __builtin_elementwise_fma(dx, GSR_V2(A), GSR_V2(bdy));
  ty = __builtin_elementwise_fma(dx, GSR_V2(B), GSR_V2(cdy));
#if GSR_QT_FORM
  q = __builtin_elementwise_fma(GSR_V2(dy), ty, dx * tx);
#else
  q = eval_q2(dx, dy, A, B, C);
#endif
}
__device__ __forceinline__ v2f eval_alpha2(v2f q, float l2op) {     // opacity exp(-q/2) = 2^(q * -0.5*log2(e) + log2 opacity)
#pragma clang fp contract(off)
  const v2f e = __builtin_elementwise_fma(q, GSR_V2(-0.72134752044448170368f), GSR_V2(l2op));
  return (v2f){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
}
__device__ __forceinline__ v2f eval_G2(v2f q) {     // exp(-q/2) = 2^(q * -0.5*log2(e))
#pragma clang fp contract(off)
  const v2f e = q * -0.72134752044448170368f;
  return (v2f){__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)};
}

// min(a, cmax) for a >= 0 as one v_med3_f32 per pixel (fminf would add a canonicalising v_max per operand);
// shared by forward and backward so both take the same clamp decision.
__device__ __forceinline__ v2f clamp_alpha2(v2f a, float cmax) {
  return (v2f){__builtin_amdgcn_fmed3f(a.x, 0.f, cmax), __builtin_amdgcn_fmed3f(a.y, 0.f, cmax)};
}

struct Splat {            // one splat's packed row, wave-uniform (lives in SGPRs)
  float u, v, A, B, C, op, qlim, f0, f1, f2, depth, l2op;   // l2op = log2(opacity) (backward walk: alpha = 2^(q c + l2op))   // qlim = min(q_max, 2 ln(op / alpha_threshold)): q <= qlim <=> the pixel contributes
  uint32_t halves;        // bit h set: the splat's support reaches tile half h (from K4 emit)
};

// packed (splat id | half mask << 30) is wave-uniform: the three 16-byte loads from the splat's packed 64-byte row
// (geometry.hip: project_sh_fwd_kernel / pack_rows_kernel) become scalar-cache loads.
template <int C, bool DEPTH = false>
__device__ __forceinline__ Splat load_splat_packed(const float* __restrict__ rec, uint32_t packed) {
  const uint32_t k = packed & 0x3FFFFFFFu;
  const float4* r = reinterpret_cast<const float4*>(rec + (size_t)GSR_ROW_FLOATS * k);
  const float4 r0 = r[0], r1 = r[1];
  Splat s;
  s.u = r0.x; s.v = r0.y; s.A = r0.z; s.B = r0.w; s.C = r1.x; s.op = r1.y; s.qlim = r1.z; s.f0 = r1.w;
  s.f1 = 0.f; s.f2 = 0.f; s.depth = 0.f; s.l2op = 0.f;
  s.halves = packed >> 30;
  if (C > 1 || DEPTH) { const float4 r2 = r[2]; s.f1 = r2.x; s.f2 = r2.y; s.depth = r2.z; s.l2op = r2.w; }
  return s;
}

// The backward walk's row: with GSR_K7_XFLEX also the x-extent of the splat's pixel box (row word 12).
struct SplatB : Splat { uint32_t bx; };
template <int C>
__device__ __forceinline__ SplatB load_splat_bwd(const float* __restrict__ rec, uint32_t packed) {
  SplatB s;
  static_cast<Splat&>(s) = load_splat_packed<C, true>(rec, packed);
  s.bx = 0u;
#if GSR_K7_XFLEX
  s.bx = __float_as_uint(rec[(size_t)GSR_ROW_FLOATS * (packed & 0x3FFFFFFFu) + 12]);
#endif
  return s;
}

// i is wave-uniform: the index load and the three 16-byte record loads become scalar-cache loads.
template <int C, bool DEPTH = false>
__device__ __forceinline__ Splat load_splat(const float* __restrict__ rec, const uint32_t* __restrict__ sorted_rank,
                                            uint32_t i) {
  const uint32_t packed = (uint32_t)__builtin_amdgcn_readfirstlane((int)sorted_rank[i]);
  const uint32_t k = packed & 0x3FFFFFFFu;
  const float4* r = reinterpret_cast<const float4*>(rec + (size_t)GSR_ROW_FLOATS * k);
  const float4 r0 = r[0], r1 = r[1];
  Splat s;
  s.u = r0.x; s.v = r0.y; s.A = r0.z; s.B = r0.w; s.C = r1.x; s.op = r1.y; s.qlim = r1.z; s.f0 = r1.w;
  s.f1 = 0.f; s.f2 = 0.f; s.depth = 0.f; s.l2op = 0.f;
  s.halves = packed >> 30;
  if (C > 1 || DEPTH) { const float4 r2 = r[2]; s.f1 = r2.x; s.f2 = r2.y; s.depth = r2.z; s.l2op = r2.w; }
  return s;
}

// Device view of the segment tables (all NULL / 0 when the caller does not segment).
#define GSR_SEG_HEAVY 0x80000000u   // tile_seg count word: the tile's FORWARD pass is segmented too (passes A, C, D)

struct SegDev {
  const uint32_t* tile_seg;    // [num_tiles, 2]: first segment, number of segments (0: short tile) | GSR_SEG_HEAVY
  const uint32_t* seg_desc;    // [capacity, 4]: tile, list start, list end, index within the tile
  const uint32_t* seg_total;   // device words (GSR_SEG_TOTAL_WORDS): [0] segments of this frame, [1] of them in heavy tiles,
                               //   [8 + x] K6's queue head of XCD x, [16 + 32 x + c] tiles of XCD x in length class c
  const uint32_t* tile_order;  // [8, GSR_TILE_CLASSES, ceil(num_tiles / 8)] tiles per XCD band and length class, or NULL
  float* seg_P;                // [capacity, 256]    product of (1 - alpha) over the segment         (pass A)
  float4* seg_TC;              // [capacity, 256] (T, c0, c1, c2) per pixel slot: T after the segment (< 0: the pixel was
                               //   dead at its entry, pass C) and the colour composited up to the END of the segment
                               //   (heavy tiles: the segment's own colour until pass D turns it into that prefix)
  int* seg_last;               // [capacity, 256]
  float* seg_median;           // [capacity, 256] or NULL
};

// K6 work distribution.  With one wave per tile and every tile resident at once (1080p: 8160 tiles on 8192 wave slots)
// a SIMD's finishing time is the sum of whichever eight lists it was dealt, and the launch lasts as long as the unluckiest
// SIMD: measured 4.6 of 8 wave slots occupied on average, the VALUs 65 % busy -- and in image order a SIMD's eight tiles
// are neighbours, all long or all short.  So the plan kernel files every (not heavy) tile under an XCD band and one of 32
// length classes (eighths of the mean list length), and block b of the forward launch takes the (b / 8)-th tile of band b % 8 counted
// from the longest class down: the dispatcher deals consecutive blocks across the SIMDs, so every SIMD gets its share of
// each class.  K6 195 -> 183 us at 500k splats, 596 -> 528 us at 3M (same box).  Which slot composites a tile changes
// nothing the tile computes: results are bit-identical to the image-order launch.  (Measured and not kept: limiting the
// residency to 4-7 waves per SIMD with dummy LDS so that the dispatcher hands out the short tiles dynamically: 213-244 us;
// reversing every other round of 32-1024 blocks: no change; 4-8 persistent waves per SIMD pulling tiles from per-band
// queue heads with integer atomics, tile-only arguments re-read from the kernel-argument segment so that the walk keeps
// its registers: 190-228 us against 185, 525-590 against 526 at 3M.)
// Bands: runs of GSR_TILE_GROUP consecutive tiles (row-major: horizontal neighbours, which share splats and so an L2) are
// dealt to the eight XCDs in turn, so every XCD samples the whole image.  (With one contiguous eighth of the image per
// XCD -- the first form -- a scene that fills the middle of the frame loaded the middle XCDs with 1.5x the pairs of the
// mean: per-wave trace at 3M splats, tools/k6_trace.py.)
#define GSR_TILE_CLASSES 32
#define GSR_TILE_GROUP 16
#define GSR_SEG_CLASS_COUNT(band) (16u + GSR_TILE_CLASSES * (band))     // word of seg_total: band's first class count
__host__ __device__ inline int gsr_tile_band(int t, int n) { (void)n; return (t / GSR_TILE_GROUP) & 7; }
// capacity of one (band, class) list = the most tiles a band can hold
__host__ __device__ inline int gsr_tile_band_stride(int n) {
  return (n + 8 * GSR_TILE_GROUP - 1) / (8 * GSR_TILE_GROUP) * GSR_TILE_GROUP;
}

// The tile of block b in the ordered launch (see GSR_TILE_CLASSES), or -1 past the end of the band's list.
__device__ __forceinline__ int ordered_tile(const SegDev& seg, int num_tiles, uint32_t b, int lane) {
  const uint32_t band = b & 7u, t = b >> 3;
  // lane l < 32 gets the number of this band's tiles in classes 31 .. 31 - l
  uint32_t upto = lane < GSR_TILE_CLASSES ? seg.seg_total[GSR_SEG_CLASS_COUNT(band) + (GSR_TILE_CLASSES - 1 - lane)] : 0u;
#pragma unroll
  for (int o = 1; o < GSR_TILE_CLASSES; o <<= 1) {
    const uint32_t below = (uint32_t)__shfl_up((int)upto, o, 64);
    if (lane >= o) upto += below;
  }
  const uint64_t longer = __ballot(lane < GSR_TILE_CLASSES && t < upto);
  if (longer == 0ull) return -1;
  const int l = __builtin_ctzll(longer);
  const uint32_t before = l ? (uint32_t)__builtin_amdgcn_readlane((int)upto, l - 1) : 0u;
  return (int)seg.tile_order[((size_t)band * GSR_TILE_CLASSES + (size_t)(GSR_TILE_CLASSES - 1 - l)) *
                                 (size_t)gsr_tile_band_stride(num_tiles) + (t - before)];
}

// per-lane pixel state of the forward walk: pixel p = 2h + i, half h (rows py0 + 8h), side i (cols px0 + 8i)
template <int C>
struct FwdPix {
  v2f T2[2], col2[2][3], med2[2];
  int lastc[4];
  // row prefetch (GSR_K6_PREFETCH): packed splat ids of the pairs two blocks ahead / bits of a row one block ahead
  uint32_t pf_rank, pf_bits, pf_sink;
};

// The walk fetches each pair's row through the scalar cache one pair ahead; at millions of splats the row table is far
// larger than the L2s and that fetch comes from HBM (K6 on c3: 42 % of the wave cycles wait for data).  Every
// GSR_K6_PREFETCH pairs the lanes therefore touch, with ordinary vector loads whose values are only folded into a sink
// word one block later, the rows of the block after next -- by the time the scalar loads ask for them they sit in L2.
// PF is chosen by the caller per frame: it pays once the row table has outgrown the caches (3M splats: K6 722 -> 656 us)
// and costs a frame whose rows sit in L2 anyway (500k: 204 -> 226 us: two more SGPRs, one more loop branch).
#ifndef GSR_K6_PREFETCH
#define GSR_K6_PREFETCH 32
#endif
template <int C, bool PF>
__device__ __forceinline__ void fwd_prefetch_init(FwdPix<C>& px, const uint32_t* __restrict__ sorted_rank,
                                                  uint32_t tile_start, uint32_t begin, uint32_t pf_end, int lane) {
  px.pf_bits = 0u; px.pf_sink = 0u; px.pf_rank = 0u;
  if (!PF) return;
  // the walk steps the prefetch at list positions that are multiples of the block length from the tile's start
  const uint32_t first = tile_start + ((begin - tile_start + GSR_K6_PREFETCH - 1u) & ~(uint32_t)(GSR_K6_PREFETCH - 1));
  const uint32_t p = first + GSR_K6_PREFETCH + (uint32_t)lane;
  if (lane < GSR_K6_PREFETCH && p < pf_end) px.pf_rank = sorted_rank[p];
}
template <int C>
__device__ __forceinline__ void fwd_prefetch_step(FwdPix<C>& px, const float* __restrict__ rec,
                                                  const uint32_t* __restrict__ sorted_rank, uint32_t i, uint32_t pf_end,
                                                  int lane) {
  px.pf_sink |= px.pf_bits;                                   // issued a block ago: long arrived
  const uint32_t p1 = i + GSR_K6_PREFETCH + (uint32_t)lane, p2 = p1 + GSR_K6_PREFETCH;
  const bool mine = lane < GSR_K6_PREFETCH;
  px.pf_bits = (mine && p1 < pf_end)
                   ? __float_as_uint(rec[(size_t)GSR_ROW_FLOATS * (px.pf_rank & 0x3FFFFFFFu)]) : 0u;
  px.pf_rank = (mine && p2 < pf_end) ? sorted_rank[p2] : 0u;
}

// Front-to-back walk over list positions [begin, end) of one tile; `tile_start` makes the recorded last-contributor
// index tile-relative.  A pixel is live while T >= T_eps.
template <int C, bool VIS, bool MEDIAN, bool PF>
__device__ __forceinline__ void fwd_walk(FwdPix<C>& px, const float* __restrict__ rec,
                                         const uint32_t* __restrict__ sorted_rank,
                                         const uint32_t* __restrict__ sorted_inst, uint32_t tile_start, uint32_t begin,
                                         uint32_t end, float fx0, float fy0, const GsrRasterParams& rp, int lane,
                                         float* __restrict__ vis_partial, float* __restrict__ pair_vis,
                                         uint32_t pf_end) {
  // lane 16r+15 ends up with the visibility total of pair (i + {0,2,1,3}[r]) of each group of four
  const uint32_t vis_slot = (uint32_t)(((lane >> 4) & 1) * 2 + (lane >> 5));
  if (begin >= end) return;
  // PF instantiation (frames whose row table has outgrown the caches): the packed splat ids of the NEXT iteration's four
  // pairs arrive by one scalar load issued at the top of this iteration, so the row fetch of a pair no longer waits for
  // an index load issued just in front of it (one exposed scalar-load latency per pair; K6 653 -> 602 us at 3M splats;
  // at 500k it costs 8 SGPRs and 8 %: 191 -> 206 us).  Reads up to 3 words past `end` (values unused): the list buffer must
  // be readable that far -- gsplat_hip.h states it for gsr_composite_forward(prefetch_rows = 1), gsr_frame_plan adds the slack.
  uint32_t rk[4] = {0u, 0u, 0u, 0u}, rk_next[4] = {0u, 0u, 0u, 0u};
  Splat nxt;
  if constexpr (PF) {
    const Rank4 q4 = *reinterpret_cast<const Rank4*>(sorted_rank + begin);
    rk[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.x); rk[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.y);
    rk[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.z); rk[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.w);
    nxt = load_splat_packed<C, MEDIAN>(rec, rk[0]);
  } else {
    nxt = load_splat<C, MEDIAN>(rec, sorted_rank, begin);
  }
  for (uint32_t i = begin; i < end; i += 4) {
    if (PF && ((i - tile_start) & (GSR_K6_PREFETCH - 1)) == 0u) fwd_prefetch_step<C>(px, rec, sorted_rank, i, pf_end, lane);
    if constexpr (PF) {
      if (i + 4 < end) {
        const Rank4 q4 = *reinterpret_cast<const Rank4*>(sorted_rank + i + 4);
        rk_next[0] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.x); rk_next[1] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.y);
        rk_next[2] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.z); rk_next[3] = (uint32_t)__builtin_amdgcn_readfirstlane((int)q4.w);
      }
    }
    float wq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (i + m < end) {                                               // wave-uniform
        const Splat s = nxt;
        if constexpr (PF) {
          if (i + m + 1 < end) nxt = load_splat_packed<C, MEDIAN>(rec, m < 3 ? rk[m + 1] : rk_next[0]);
        } else {
          // prefetch of the next pair's row (scalar loads), UNCONDITIONAL -- the last pair of the list re-reads its own
          // row -- so that `nxt` is always a fresh value and the compiler alternates register sets instead of copying
          // the row from "next" to "current" once per pair (-16 scalar instructions per pair; K6 -2 % at 500k splats;
          // at 3M the re-reads cost more than the copies: +2 %, so the large-frame instantiation keeps the condition)
          nxt = load_splat<C, MEDIAN>(rec, sorted_rank, min(i + m + 1, end - 1u));
        }
        // offsets from the mean: ONE rounding per pixel (centre - mean, the centre exact), whichever half / side it is in
        const v2f dx2 = (v2f){fx0, fx0 + 8.f} - GSR_V2(s.u);
        const int idx = (int)(i - tile_start) + m + 1;
        v2f wsum2 = GSR_V2(0.f);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (!(s.halves & (1u << h))) continue;                        // scalar test: support misses this half
          const float dy = (h ? fy0 + 8.f : fy0) - s.v;
          const v2f q = eval_q2(dx2, dy, s.A, s.B, s.C);
          // q <= qlim = min(q_max, 2 ln(opacity / alpha_threshold)) says both "inside the support" and "alpha reaches
          // the threshold" (alpha = opacity exp(-q/2); the clamp at 0.99 lies above the threshold)
          const bool hit0 = px.T2[h].x >= rp.T_eps && q.x <= s.qlim;
          const bool hit1 = px.T2[h].y >= rp.T_eps && q.y <= s.qlim;
          if (__ballot(hit0 || hit1) != 0ull) {
            const v2f G = eval_G2(q);
            const v2f a_raw = G * s.op;
            v2f alpha = clamp_alpha2(a_raw, rp.clamp_max_alpha);
            alpha = (v2f){hit0 ? alpha.x : 0.f, hit1 ? alpha.y : 0.f};
            const v2f w = alpha * px.T2[h];
            px.col2[h][0] = __builtin_elementwise_fma(w, GSR_V2(s.f0), px.col2[h][0]);
            if (C > 1) px.col2[h][1] = __builtin_elementwise_fma(w, GSR_V2(s.f1), px.col2[h][1]);
            if (C > 2) px.col2[h][2] = __builtin_elementwise_fma(w, GSR_V2(s.f2), px.col2[h][2]);
            wsum2 += w;
            px.T2[h] = px.T2[h] - w;                           // T (1 - alpha), with w = alpha T already formed
            if (hit0) px.lastc[2 * h] = idx;
            if (hit1) px.lastc[2 * h + 1] = idx;
            if (MEDIAN) {
              if (hit0 && px.med2[h].x == 0.f && px.T2[h].x < 0.5f) px.med2[h].x = s.depth;
              if (hit1 && px.med2[h].y == 0.f && px.T2[h].y < 0.5f) px.med2[h].y = s.depth;
            }
          }
        }
        wq[m] = wsum2.x + wsum2.y;
      }
    }
    if (VIS) {
      // four pairs reduced at once, transposing: swap32 (4 -> 2 registers), swap16 (2 -> 1), then 4 row steps
      float r = gsr_swap16_add(gsr_swap32_add(wq[0], wq[1]), gsr_swap32_add(wq[2], wq[3]));
      r = gsr_row_sum_to_lane15(r);
      const uint32_t pos = i + vis_slot;
      if ((lane & 15) == 15 && pos < end) {
        // sorted-position copy (read back coalesced by the backward pass) + per-instance copy (per-splat sums)
        pair_vis[pos] = r;
        if (r > 0.f) vis_partial[sorted_inst[pos]] = r;
      }
    }
    const bool live = px.T2[0].x >= rp.T_eps || px.T2[0].y >= rp.T_eps || px.T2[1].x >= rp.T_eps ||
                      px.T2[1].y >= rp.T_eps;
    if (__ballot(live) == 0ull) break;
    if constexpr (PF) { rk[0] = rk_next[0]; rk[1] = rk_next[1]; rk[2] = rk_next[2]; rk[3] = rk_next[3]; }
  }
}

template <int C>
__device__ __forceinline__ void fwd_init(FwdPix<C>& px, int px0, int py0, int W, int H) {
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const bool in_y = (py0 + 8 * h) < H;            // pixels outside the image start at T = 0 and are never written
    px.T2[h] = (v2f){(in_y && px0 < W) ? 1.f : 0.f, (in_y && (px0 + 8) < W) ? 1.f : 0.f};
    px.med2[h] = GSR_V2(0.f);
    px.col2[h][0] = px.col2[h][1] = px.col2[h][2] = GSR_V2(0.f);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) px.lastc[p] = 0;
}

// Pass A of a heavy tile (extra blocks of the forward launch): per-pixel product of (1 - alpha) over one segment,
// formed with the same alpha and the same update (P -= alpha P) as the walk itself.
template <int C>
__device__ __forceinline__ void seg_alpha_pass(uint32_t sidx, const float* __restrict__ rec,
                                               const uint32_t* __restrict__ sorted_rank, int tiles_x,
                                               const GsrRasterParams& rp, const SegDev& seg) {
  const uint32_t* d = seg.seg_desc + 4 * (size_t)sidx;
  const int tile = (int)d[0];
  const uint32_t begin = d[1], end = d[2];
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const float fx0 = (float)(tx * 16 + (lane & 7)) + 0.5f, fy0 = (float)(ty * 16 + (lane >> 3)) + 0.5f;
  v2f P2[2] = {GSR_V2(1.f), GSR_V2(1.f)};
  Splat nxt = load_splat<1>(rec, sorted_rank, begin);
  for (uint32_t i = begin; i < end; ++i) {
    const Splat s = nxt;
    if (i + 1 < end) nxt = load_splat<1>(rec, sorted_rank, i + 1);
    const v2f dx2 = (v2f){fx0, fx0 + 8.f} - GSR_V2(s.u);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (!(s.halves & (1u << h))) continue;
      const float dy = (h ? fy0 + 8.f : fy0) - s.v;
      const v2f q = eval_q2(dx2, dy, s.A, s.B, s.C);
      const bool in0 = q.x <= s.qlim, in1 = q.y <= s.qlim;
      if (__ballot(in0 || in1) != 0ull) {
        v2f alpha = clamp_alpha2(eval_G2(q) * s.op, rp.clamp_max_alpha);
        alpha = (v2f){in0 ? alpha.x : 0.f, in1 ? alpha.y : 0.f};
        P2[h] = P2[h] - alpha * P2[h];
      }
    }
  }
  float* out = seg.seg_P + 256 * (size_t)sidx + lane;
  out[0] = P2[0].x; out[64] = P2[0].y; out[128] = P2[1].x; out[192] = P2[1].y;
}

#ifdef GSR_K6_TRACE
// Diagnostic build only (tools/k6_trace.py): every forward wave records when and where it ran.
__device__ uint64_t* g_k6_trace = nullptr;      // [num_tiles, 4]: start, end (s_memrealtime), hardware ids, list length | block
#endif
template <int C, bool VIS, bool MEDIAN, bool PF>
__global__ __launch_bounds__(64) void composite_fwd_kernel(const float* __restrict__ rec,
                                                           const uint32_t* __restrict__ sorted_rank,
                                                           const uint32_t* __restrict__ sorted_inst,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           float* __restrict__ image, float* __restrict__ final_T,
                                                           int* __restrict__ last, float* __restrict__ median,
                                                           float* __restrict__ vis_partial,
                                                           float* __restrict__ pair_vis, SegDev seg, int tile_blocks) {
  if ((int)blockIdx.x >= tile_blocks) {                              // extra blocks: pass A of the heavy tiles' segments
    const uint32_t h = blockIdx.x - (uint32_t)tile_blocks;           // index into the plan's compact list of them
    if (h < seg.seg_total[1]) {
      const uint32_t sidx = (seg.tile_seg + 2 * (size_t)num_tiles)[h];
      if (sidx != 0xFFFFFFFFu) {
        const uint32_t* d = seg.seg_desc + 4 * (size_t)sidx;
        if (d[1] < d[2]) seg_alpha_pass<C>(sidx, rec, sorted_rank, tiles_x, rp, seg);
      }
    }
    return;
  }
  const int lane = (int)threadIdx.x;
  int tile;
  if (seg.tile_order) {
    tile = ordered_tile(seg, num_tiles, blockIdx.x, lane);
    if (tile < 0) return;                                             // past the band's last tile
  } else {
    tile = gsr_xcd_remap((int)blockIdx.x, num_tiles);
  }
  const uint32_t tseg = seg.tile_seg ? seg.tile_seg[2 * tile + 1] : 0u;
  if (tseg & GSR_SEG_HEAVY) return;                                   // heavy tile: passes A, C, D composite it
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f;
  const uint32_t start = tile_range[2 * tile], end = tile_range[2 * tile + 1];
#ifdef GSR_K6_TRACE
  const uint64_t trace_t0 = __builtin_amdgcn_s_memrealtime();
#endif

  FwdPix<C> px;
  fwd_init<C>(px, px0, py0, W, H);
  fwd_prefetch_init<C, PF>(px, sorted_rank, start, start, end, lane);
  if (tseg == 0u) {
    fwd_walk<C, VIS, MEDIAN, PF>(px, rec, sorted_rank, sorted_inst, start, start, end, fx0, fy0, rp, lane, vis_partial,
                             pair_vis, end);
  } else {
    // A long (but not heavy) tile is still walked by this one wave, but the walk pauses at the segment ends and leaves
    // a checkpoint -- (T, colour composited so far) per pixel, one 16-byte store each -- so that the backward pass can
    // give every segment a wave of its own (finer work units: its tail and its latency-bound last waves shrink).
    // (Kept as a second copy of the walk: the one-segment form of this loop costs the short-tile path 10-20 %.)
    const uint32_t first = seg.tile_seg[2 * tile];
    for (uint32_t j = 0; j < tseg; ++j) {
      const uint32_t* d = seg.seg_desc + 4 * (size_t)(first + j);
      fwd_walk<C, VIS, MEDIAN, PF>(px, rec, sorted_rank, sorted_inst, start, d[1], d[2], fx0, fy0, rp, lane, vis_partial,
                               pair_vis, end);
      float4* out = seg.seg_TC + 256 * (size_t)(first + j) + lane;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int h = p >> 1;
        out[64 * p] = (p & 1) ? make_float4(px.T2[h].y, px.col2[h][0].y, px.col2[h][1].y, px.col2[h][2].y)
                              : make_float4(px.T2[h].x, px.col2[h][0].x, px.col2[h][1].x, px.col2[h][2].x);
      }
      const bool live = px.T2[0].x >= rp.T_eps || px.T2[0].y >= rp.T_eps || px.T2[1].x >= rp.T_eps ||
                        px.T2[1].y >= rp.T_eps;
      if (__ballot(live) == 0ull) break;            // every pixel saturated: later segments contribute to nothing
    }
  }

  if (PF && (px.pf_sink | px.pf_bits) == 0x7FC0FFEEu) final_T[0] = 0.f;   // never true (a quiet-NaN pattern no row holds): keeps the prefetch loads alive
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = px0 + 8 * (p & 1), y = py0 + 8 * (p >> 1);
    if (x < W && y < H) {
      const size_t pix = (size_t)y * W + x;
      const int h = p >> 1;
#pragma unroll
      for (int c = 0; c < C; ++c) image[pix * C + c] = (p & 1) ? px.col2[h][c].y : px.col2[h][c].x;
      final_T[pix] = (p & 1) ? px.T2[h].y : px.T2[h].x;
      last[pix] = px.lastc[p];
      if (MEDIAN) median[pix] = (p & 1) ? px.med2[h].y : px.med2[h].x;
    }
  }
#ifdef GSR_K6_TRACE
  if (g_k6_trace && lane == 0) {
    uint64_t* t = g_k6_trace + 4 * (size_t)tile;
    t[0] = trace_t0; t[1] = __builtin_amdgcn_s_memrealtime();
    t[2] = (uint64_t)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |            // HW_REG_HW_ID
           ((uint64_t)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);    // HW_REG_XCC_ID
    t[3] = (uint64_t)(end - start) | ((uint64_t)blockIdx.x << 32);
  }
#endif
}

// Pass C of a heavy tile: the forward walk over one segment, entered with T_in = product of the preceding segments'
// products (taken in segment order).  Outputs go to the segment's own 256-pixel slots (lane-major: slot p*64 + lane).
template <int C, bool VIS, bool MEDIAN, bool PF>
__global__ __launch_bounds__(64) void seg_composite_kernel(const float* __restrict__ rec,
                                                           const uint32_t* __restrict__ sorted_rank,
                                                           const uint32_t* __restrict__ sorted_inst,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           float* __restrict__ vis_partial,
                                                           float* __restrict__ pair_vis, SegDev seg) {
  if (blockIdx.x >= seg.seg_total[1]) return;                         // one block per segment of a HEAVY tile
  const uint32_t sidx = (seg.tile_seg + 2 * (size_t)num_tiles)[blockIdx.x];
  if (sidx == 0xFFFFFFFFu) return;
  const uint32_t* d = seg.seg_desc + 4 * (size_t)sidx;
  const int tile = (int)d[0];
  const uint32_t begin = d[1], end = d[2];
  const uint32_t first = seg.tile_seg[2 * tile];
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f;

  FwdPix<C> px;
  fwd_init<C>(px, px0, py0, W, H);
  fwd_prefetch_init<C, PF>(px, sorted_rank, tile_range[2 * tile], begin, end, lane);
  for (uint32_t s = first; s < sidx; ++s) {
    const float* P = seg.seg_P + 256 * (size_t)s + lane;
    px.T2[0] = px.T2[0] * (v2f){P[0], P[64]};
    px.T2[1] = px.T2[1] * (v2f){P[128], P[192]};
  }
  const bool alive[4] = {px.T2[0].x >= rp.T_eps, px.T2[0].y >= rp.T_eps, px.T2[1].x >= rp.T_eps, px.T2[1].y >= rp.T_eps};
  if (__ballot(alive[0] || alive[1] || alive[2] || alive[3]) != 0ull)
    fwd_walk<C, VIS, MEDIAN, PF>(px, rec, sorted_rank, sorted_inst, tile_range[2 * tile], begin, end, fx0, fy0, rp, lane,
                             vis_partial, pair_vis, end);
  const size_t o = 256 * (size_t)sidx + lane;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int h = p >> 1;
    const float t = (p & 1) ? px.T2[h].y : px.T2[h].x;
    seg.seg_TC[o + 64 * p] = (p & 1) ? make_float4(alive[p] ? t : -1.f, px.col2[h][0].y, px.col2[h][1].y, px.col2[h][2].y)
                                     : make_float4(alive[p] ? t : -1.f, px.col2[h][0].x, px.col2[h][1].x, px.col2[h][2].x);
    seg.seg_last[o + 64 * p] = px.lastc[p];
    if (MEDIAN) seg.seg_median[o + 64 * p] = (p & 1) ? px.med2[h].y : px.med2[h].x;
  }
}

// Pass D of a heavy tile: one wave per tile adds the segment colours in list order, takes T / last / median from the
// segments that saw the pixel alive, writes the image, and replaces every segment's colour by the colour BEHIND it
// (what its reverse walk starts from).
template <int C, bool MEDIAN>
__global__ __launch_bounds__(64) void seg_combine_kernel(int W, int H, int tiles_x, int num_tiles,
                                                         float* __restrict__ image, float* __restrict__ final_T,
                                                         int* __restrict__ last, float* __restrict__ median,
                                                         SegDev seg) {
  const int tile = (int)blockIdx.x;
  const uint32_t tseg = seg.tile_seg[2 * tile + 1];
  if (!(tseg & GSR_SEG_HEAVY)) return;
  const uint32_t n = tseg & ~GSR_SEG_HEAVY;
  const uint32_t first = seg.tile_seg[2 * tile];
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  float col[4][3], T[4], med[4];
  int lastc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    col[p][0] = col[p][1] = col[p][2] = 0.f;
    T[p] = 1.f; med[p] = 0.f; lastc[p] = 0;
  }
  for (uint32_t j = 0; j < n; ++j) {
    const size_t s = first + j, o = 256 * s + lane;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float4 tc = seg.seg_TC[o + 64 * p];
      if (tc.x >= 0.f) T[p] = tc.x;
      const int l = seg.seg_last[o + 64 * p];
      if (l != 0) lastc[p] = l;
      if (MEDIAN) {
        const float m = seg.seg_median[o + 64 * p];
        if (med[p] == 0.f && m != 0.f) med[p] = m;
      }
      col[p][0] += tc.y; col[p][1] += tc.z; col[p][2] += tc.w;      // the segment's own colour becomes the colour up to its end
      seg.seg_TC[o + 64 * p] = make_float4(tc.x, col[p][0], col[p][1], col[p][2]);
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int x = px0 + 8 * (p & 1), y = py0 + 8 * (p >> 1);
    if (x < W && y < H) {
      const size_t pix = (size_t)y * W + x;
#pragma unroll
      for (int c = 0; c < C; ++c) image[pix * C + c] = col[p][c];
      final_T[pix] = T[p];
      last[pix] = lastc[p];
      if (MEDIAN) median[pix] = med[p];
    }
  }
}

#ifndef GSR_K7_WAVES
#define GSR_K7_WAVES 6
#endif
#ifndef GSR_K7_SEGMAP             // segment block -> segment: 0 identity, 1 contiguous range per XCD, 2 grouped
#define GSR_K7_SEGMAP 2
#endif
#ifndef GSR_K7_SEG_GROUP_LOG2
#define GSR_K7_SEG_GROUP_LOG2 5
#endif
#ifndef GSR_K7_UNCOND_PREFETCH   // measured: K7 -1.5 % at 500k splats, +2.5 % at 3M (the chunk ends' re-reads): off
#define GSR_K7_UNCOND_PREFETCH 0
#endif
#ifndef GSR_K7_SEG_FIRST
#define GSR_K7_SEG_FIRST 1
#endif
#ifndef GSR_K7_XFLEX             // narrow footprints: one 8-pixel-wide window per tile half instead of the packed half (see the walk)
#define GSR_K7_XFLEX 0
#endif
#ifndef GSR_K7_ADDTID            // per-pair sums parked with ds_write_addtid_b32 (see the reduction)
#define GSR_K7_ADDTID 1
#endif
template <int C>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(GSR_K7_WAVES, GSR_K7_WAVES))) void composite_bwd_kernel(const float* __restrict__ rec,
                                                           const uint32_t* __restrict__ sorted_rank,
                                                           const uint32_t* __restrict__ sorted_inst,
                                                           const float* __restrict__ pair_vis,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           const float* __restrict__ final_T,
                                                           const int* __restrict__ last,
                                                           const float* __restrict__ dL_dimage,
                                                           const float* __restrict__ image,
                                                           float* __restrict__ partial, SegDev seg,
                                                           uint32_t seg_capacity, uint32_t seg_blocks) {
  // A block walks tile-relative list positions [lo, hi) in reverse: a short tile's whole list, or -- extra blocks of
  // the launch -- one segment of a longer tile, entered with that segment's own end state from the forward pass.
  int tile, lo = 0, seg_hi = 0x7fffffff;
  uint32_t sidx = 0u;
#if GSR_K7_SEG_FIRST
  // segments (the longest work units) first, then the unsegmented tiles longest class first: the launch has several
  // times more blocks than the chip has slots, the dispatcher hands them out in index order, so what starts last -- and
  // sets the length of the launch's tail -- should be the shortest units
  const bool is_seg = blockIdx.x < seg_blocks;
  if (is_seg) {
    sidx = blockIdx.x;
#else
  const bool is_seg = (int)blockIdx.x >= num_tiles;
  if (is_seg) {
    sidx = blockIdx.x - (uint32_t)num_tiles;
#endif
#if GSR_K7_SEGMAP == 2
    sidx = gsr_xcd_group_remap(sidx, GSR_K7_SEG_GROUP_LOG2);     // a tile's segments share an XCD (and its L2)
    if (sidx >= min(seg.seg_total[0], seg_capacity)) return;     // (the grid is rounded up past the tables' capacity)
#elif GSR_K7_SEGMAP == 1
    if (sidx >= seg.seg_total[0]) return;
    sidx = (uint32_t)gsr_xcd_remap((int)sidx, (int)seg.seg_total[0]);
#else
    if (sidx >= seg.seg_total[0]) return;
#endif
    const uint32_t* d = seg.seg_desc + 4 * (size_t)sidx;
    tile = (int)d[0];
    const uint32_t tstart = tile_range[2 * tile];
    lo = (int)(d[1] - tstart);
    seg_hi = (int)(d[2] - tstart);
  } else {
#if GSR_K7_SEG_FIRST
    const uint32_t b = blockIdx.x - seg_blocks;                       // (seg_blocks is a multiple of 8: same XCD)
    if (seg.tile_order) {
      tile = ordered_tile(seg, num_tiles, b, (int)threadIdx.x);
      if (tile < 0) return;
    } else {
      if ((int)b >= num_tiles) return;
      tile = gsr_xcd_remap((int)b, num_tiles);
    }
#else
    tile = gsr_xcd_remap((int)blockIdx.x, num_tiles);
#endif
    if (seg.tile_seg && seg.tile_seg[2 * tile + 1] != 0u) return;     // segmented tile: its segment blocks handle it
  }
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f;
  const uint32_t start = tile_range[2 * tile];
#if GSR_K7_XFLEX
  const int lx8 = lane & 7, tile_px0 = tx * 16;
#endif

  // per pixel (packed over the two sides of a half): T behind the current splat, g = dL/dC,
  // ga = g . (colour accumulated behind the current splat)
  v2f T2[2], g2[2][3], ga2[2];
  int lastc[4];
  int tile_last = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    T2[h] = GSR_V2(1.f); ga2[h] = GSR_V2(0.f);
    g2[h][0] = g2[h][1] = g2[h][2] = GSR_V2(0.f);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int px = px0 + 8 * (p & 1), py = py0 + 8 * (p >> 1);
    const int h = p >> 1;
    lastc[p] = 0;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
      float t = final_T[pix];
      float4 tc = make_float4(0.f, 0.f, 0.f, 0.f);
      if (is_seg) {                                   // T after this segment (dead-at-entry pixels never contribute)
        tc = seg.seg_TC[256 * (size_t)sidx + 64 * p + lane];
        t = tc.x < 0.f ? 1.f : tc.x;
      }
      const float upto[3] = {tc.y, tc.z, tc.w};
      if (p & 1) T2[h].y = t; else T2[h].x = t;
      lastc[p] = last[pix];
      float gb = 0.f;
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float gv = dL_dimage[pix * C + c];
        if (p & 1) g2[h][c].y = gv; else g2[h][c].x = gv;
        // colour behind the segment = final colour - colour composited up to the segment's end
        if (is_seg) gb = fmaf(gv, image[pix * C + c] - upto[c], gb);
      }
      if (p & 1) ga2[h].y = gb; else ga2[h].x = gb;   // g . (colour behind the segment); 0 for a whole tile
    }
    tile_last = max(tile_last, lastc[p]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tile_last = max(tile_last, __shfl_xor(tile_last, o, 64));
  tile_last = __builtin_amdgcn_readfirstlane(tile_last);
  const int hi = min(tile_last, seg_hi);
  if (hi <= lo) return;
  __shared__ float red[12 * 80];
#if GSR_K7_ADDTID
  const int rslot = lane < 44 ? (lane >> 2) * 68 + (lane & 3) * 16 : 0;   // reader 4k+p: quarter p of value k (68-word rows)
#else
  const int wslot = (lane >> 4) * 20 + (lane & 15);                  // my cell inside a value's 80-word row
  const int rslot = lane < 48 ? (lane >> 2) * 80 + (lane & 3) * 20 : 0;   // reader 4k+p: quarter p of value k
#endif

  for (int cbase = lo + (((hi - lo - 1) >> 6) << 6); cbase >= lo; cbase -= 64) {
    const int n = min(64, hi - cbase);
    // pairs that touched no pixel in the forward pass are skipped without evaluating anything
    // one coalesced vector load per 64 pairs for the skip flags, the packed ranks and the slot ids; a pair's
    // values are then broadcast with v_readlane (no dependent index load in front of the record fetch)
    const uint32_t li = start + (uint32_t)cbase + (uint32_t)lane;
    const float pv = (lane < n) ? pair_vis[li] : 0.f;
    const int my_rank = (lane < n) ? (int)sorted_rank[li] : 0;
    const int my_inst = (lane < n) ? (int)sorted_inst[li] : 0;
    uint64_t flags = __ballot(pv > 0.f);
    if (flags == 0ull) continue;
    int j = 63 - __builtin_clzll(flags);
    SplatB nxt = load_splat_bwd<C>(rec, (uint32_t)__builtin_amdgcn_readlane(my_rank, j));
    uint32_t inst_nxt = (uint32_t)__builtin_amdgcn_readlane(my_inst, j);
    while (true) {
      const SplatB s = nxt;
      const uint32_t inst_j = inst_nxt;
      const int pos = cbase + j;
      flags &= ~(1ull << j);
      const bool more = flags != 0ull;
#if GSR_K7_UNCOND_PREFETCH
      // prefetch the next contributing pair -- unconditionally (the chunk's last pair re-reads its own row), so that
      // `nxt` is always a fresh value: no copy of the row from "next" to "current" registers per pair
      j = more ? 63 - __builtin_clzll(flags) : j;
      nxt = load_splat_bwd<C>(rec, (uint32_t)__builtin_amdgcn_readlane(my_rank, j));
      inst_nxt = (uint32_t)__builtin_amdgcn_readlane(my_inst, j);
#else
      if (more) {                                             // prefetch the next contributing pair
        j = 63 - __builtin_clzll(flags);
        nxt = load_splat_bwd<C>(rec, (uint32_t)__builtin_amdgcn_readlane(my_rank, j));
        inst_nxt = (uint32_t)__builtin_amdgcn_readlane(my_inst, j);
      }
#endif
      const v2f dx2 = (v2f){fx0, fx0 + 8.f} - GSR_V2(s.u);      // (one rounding per pixel: fwd_walk's)
      // Per pair and pixel the geometry gradient enters through ONE scalar, GdG = G dL/dG; what is accumulated are its
      // moments about the splat's mean, sum GdG {dx, dy, dx^2, dx dy, dy^2} -- fewer packed operations per half than
      // accumulating du, dv, dA, dB, dC themselves -- and the per-splat sweep that consumes the reduced rows converts
      // them once per splat:  du = A mx + B my, dv = B mx + C my, dA = -mxx / 2, dB = -mxy, dC = -myy / 2,
      // dopacity = m0 / opacity (m0 = sum GdG, the zeroth moment).
      v2f mx2 = GSR_V2(0.f), my2 = mx2, mxx2 = mx2, mxy2 = mx2, myy2 = mx2, dop2 = mx2, prune2 = mx2, split2 = mx2;
      v2f df2[3] = {mx2, mx2, mx2};
#if GSR_K7_XFLEX
      // A footprint at most 8 pixels wide inside this tile (wave-uniform, from the row's pixel box) is evaluated on ONE
      // 8-wide window per tile half instead of the half's 16 columns: lane (lx, ly) takes the column of [a, a + 8)
      // congruent to lx -- its left-quadrant pixel where lx >= a, its right-quadrant one otherwise -- in plain fp32, the
      // pixel's state picked out of (and put back into) the halves of the packed register pairs by a per-lane select.
      const int bx0 = max((int)(int16_t)(s.bx & 0xFFFFu) - tile_px0, 0), bx1 = min(((int)s.bx >> 16) - tile_px0, 15);
      const bool narrow = bx1 - bx0 < 8;
      const int xa = min(bx0, 8);
      const bool sel = lx8 < xa;                                 // this lane's pixel of the window is its right-quadrant one
      const float dxs = (sel ? fx0 + 8.f : fx0) - s.u;
#endif
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (!(s.halves & (1u << h))) continue;                            // scalar test: support misses this half
        const float dy = (h ? fy0 + 8.f : fy0) - s.v;
#if GSR_K7_XFLEX
#include "composite_k7_xflex.inc"
#endif
        v2f q, tx_, ty_;
        eval_qt(dx2, dy, s.A, s.B, s.C, q, tx_, ty_);
        const bool hit0 = pos < lastc[2 * h] && q.x <= s.qlim;       // the forward walk's test (see fwd_walk)
        const bool hit1 = pos < lastc[2 * h + 1] && q.y <= s.qlim;
        if (__ballot(hit0 || hit1) != 0ull) {
          // alpha = opacity exp(-q/2) in one step, 2^(q c + log2 opacity): the backward walk never needs G = exp(-q/2)
          // by itself -- G dL/dG = alpha dL/dalpha where the clamp is inactive -- so one packed multiply per half goes
          const v2f a_raw = eval_alpha2(q, s.l2op);
          v2f alpha = clamp_alpha2(a_raw, rp.clamp_max_alpha);
          alpha = (v2f){hit0 ? alpha.x : 0.f, hit1 ? alpha.y : 0.f};
          const v2f om = GSR_V2(1.f) - alpha;
          const v2f inv = {__builtin_amdgcn_rcpf(om.x), __builtin_amdgcn_rcpf(om.y)};   // rcp(1) == 1 exactly
          const v2f Tb = T2[h] * inv;                        // transmittance in front of this splat
          T2[h] = Tb;
          const v2f w = alpha * Tb;
          v2f gc = g2[h][0] * s.f0;
          if (C > 1) gc = __builtin_elementwise_fma(g2[h][1], GSR_V2(s.f1), gc);
          if (C > 2) gc = __builtin_elementwise_fma(g2[h][2], GSR_V2(s.f2), gc);
          df2[0] = __builtin_elementwise_fma(w, g2[h][0], df2[0]);
          if (C > 1) df2[1] = __builtin_elementwise_fma(w, g2[h][1], df2[1]);
          if (C > 2) df2[2] = __builtin_elementwise_fma(w, g2[h][2], df2[2]);
          const v2f dLda = Tb * gc - ga2[h] * inv;           // dC/dalpha = T c - (colour behind)/(1-alpha)
          ga2[h] = __builtin_elementwise_fma(gc, w, ga2[h]);
          // |x| rides as a source modifier on the plain fma (the packed form would need a separate v_and per pixel)
          prune2 = (v2f){__builtin_fmaf(__builtin_fabsf(dLda.x), alpha.x, prune2.x),
                         __builtin_fmaf(__builtin_fabsf(dLda.y), alpha.y, prune2.y)};
          // gradient through G = exp(-q/2) only where the pixel contributed and the alpha clamp is inactive
          const bool m0 = hit0 && a_raw.x <= rp.clamp_max_alpha;
          const bool m1 = hit1 && a_raw.y <= rp.clamp_max_alpha;
          v2f GdG = a_raw * dLda;                            // G dL/dG = alpha dL/dalpha (clamp inactive)
          GdG = (v2f){m0 ? GdG.x : 0.f, m1 ? GdG.y : 0.f};
          dop2 += GdG;                                       // zeroth moment: dL/dopacity = (sum GdG) / opacity, formed per splat
          const v2f px_ = GdG * dx2, py_ = GdG * dy;
          mx2 += px_;
          my2 += py_;
          mxx2 = __builtin_elementwise_fma(px_, dx2, mxx2);
          mxy2 = __builtin_elementwise_fma(px_, GSR_V2(dy), mxy2);
          myy2 = __builtin_elementwise_fma(py_, GSR_V2(dy), myy2);
          // split score: sum_px |dL_px/d(u,v)| = |GdG| |conic d|
          const v2f nn = __builtin_elementwise_fma(ty_, ty_, tx_ * tx_);
          split2 = (v2f){__builtin_fmaf(__builtin_fabsf(GdG.x), __builtin_amdgcn_sqrtf(nn.x), split2.x),
                         __builtin_fmaf(__builtin_fabsf(GdG.y), __builtin_amdgcn_sqrtf(nn.y), split2.y)};
        }
      }
      float du = mx2.x + mx2.y, dv = my2.x + my2.y, dA = mxx2.x + mxx2.y, dB = mxy2.x + mxy2.y, dC = myy2.x + myy2.y;
      float dop = dop2.x + dop2.y, prune = prune2.x + prune2.y, split = split2.x + split2.y;
      float df[3] = {df2[0].x + df2[0].y, df2[1].x + df2[1].y, df2[2].x + df2[2].y};
      // Per-pair reduction of the 11 sums over the tile's 256 pixels THROUGH LDS (the wave owns the block's LDS, no
      // barrier): every lane parks its 11 values, value-major, then lane 4k+p adds quarter p (16 lanes) of value k with
      // packed adds and two quad DPP steps finish it -- 8 packed adds + 3 scalar ops on the VALU instead of 9 permlane
      // swaps + 9 adds + 12 DPP steps of the register-only transposing tree (the VALU is the binding unit here, the LDS
      // pipe is idle otherwise).  Rows are padded (16 -> 20 words per quarter, 80 per value) so that the 128-bit reads
      // of 8 consecutive lanes fall into 32 distinct banks.  Fixed association order => bit-reproducible.
      {
#if GSR_K7_ADDTID
        // value rows of 64 lanes (+4 words: conflict-free 16-byte reads), parked by lane id -- no address register, 16-bit
        // offsets (the two-address form above needs three extra address adds per pair), half the store-path cycles
        asm volatile("s_mov_b32 m0, %11\n"
                     "s_nop 0\n"                                  /* hazard: SALU write of M0 -> LDS add-TID needs a wait state */
                     "ds_write_addtid_b32 %0 offset:0\n"
                     "ds_write_addtid_b32 %1 offset:272\n"
                     "ds_write_addtid_b32 %2 offset:544\n"
                     "ds_write_addtid_b32 %3 offset:816\n"
                     "ds_write_addtid_b32 %4 offset:1088\n"
                     "ds_write_addtid_b32 %5 offset:1360\n"
                     "ds_write_addtid_b32 %6 offset:1632\n"
                     "ds_write_addtid_b32 %7 offset:1904\n"
                     "ds_write_addtid_b32 %8 offset:2176\n"
                     "ds_write_addtid_b32 %9 offset:2448\n"
                     "ds_write_addtid_b32 %10 offset:2720\n"
                     :: "v"(du), "v"(dv), "v"(dA), "v"(dB), "v"(dC), "v"(dop), "v"(prune), "v"(split), "v"(df[0]), "v"(df[1]),
                        "v"(df[2]), "s"((uint32_t)(uintptr_t)red) : "memory");
#else
        float* wr = red + wslot;
        wr[0 * 80] = du; wr[1 * 80] = dv; wr[2 * 80] = dA; wr[3 * 80] = dB; wr[4 * 80] = dC; wr[5 * 80] = dop;
        wr[6 * 80] = prune; wr[7 * 80] = split; wr[8 * 80] = df[0]; wr[9 * 80] = df[1]; wr[10 * 80] = df[2];
#endif
        gsr_wave_lds_fence();                                  // parks above are visible to the reader lanes below
        const float4* rd = reinterpret_cast<const float4*>(red + rslot);
        const float4 a = rd[0], b = rd[1], c = rd[2], d = rd[3];
        gsr_wave_lds_fence();                                  // reads done before the next pair parks into the same cells
        const v2f t = (((v2f){a.x, a.y} + (v2f){a.z, a.w}) + ((v2f){b.x, b.y} + (v2f){b.z, b.w})) +
                      (((v2f){c.x, c.y} + (v2f){c.z, c.w}) + ((v2f){d.x, d.y} + (v2f){d.z, d.w}));
        float tot = t.x + t.y;
        asm volatile("s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                     "s_nop 1\n"
                     "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
                     : "+v"(tot));
        if (lane < 44 && (lane & 3) == 0) partial[(size_t)GSR_PARTIAL_FLOATS * inst_j + (lane >> 2)] = tot;
      }
      if (!more) break;
    }
  }
}

// K7 with flexible 64-pixel windows and the tile's pixel state in LDS: measured and rejected, lives in
// composite_k7_windows.inc and is compiled only for the variant build that reproduces the numbers on file.
#ifndef GSR_K7_WINDOWS
#define GSR_K7_WINDOWS 0
#endif
#if GSR_K7_WINDOWS
#include "composite_k7_windows.inc"
#endif

// Plans the frame, one thread per tile.  A tile longer than `seg_pairs` is cut into segments (the last one shorter),
// numbered consecutively from a slot range the tile reserves with one integer atomic on the (zero-initialised) segment
// counter; a tile longer than `heavy_min` is flagged GSR_SEG_HEAVY: its forward pass is segmented too.  Which range a tile gets depends on arrival
// order; nothing else does: a tile's segments are contiguous and in list order, every result is a function of that alone.
// (segment length, heavy-tile threshold) of a frame with O (tile, splat) pairs; configuration values <= 0 mean automatic.
// One definition for the host (buffer sizing, gsr_segment_thresholds) and the plan kernel (which may be launched with O
// still on the device), so a frame is cut the same way whichever side evaluates it.
//
// Heavy threshold: cutting a tile's FORWARD walk only pays when that tile would otherwise outlast the rest of the launch
// -- a segmented forward pass costs an extra alpha-product pass, so a frame whose tiles are all equally long (3M splats
// at 1080p: ~800 pairs on EVERY tile) must not be cut there.  A lone wave walks ~6 pairs per microsecond while the
// balanced launch takes ~(40 + 0.11 O / 1000) us (measured K6 fit): a tile is heavy above about half of what one wave
// can walk in that time.
// Segment length: the BACKWARD pass is cheaper to split -- the one-wave forward walk just stores a 4 KB checkpoint per
// segment end -- and gains from finer work units (measured: K7 -5 % on c2 at 64-pair segments, -13..20 % on c3 at
// 256..128), so with gradients a tile is cut into about six segments (mean list length / 6 within [64, 256], multiple
// of 4); without gradients only heavy tiles are cut, into segments half a threshold long.
__host__ __device__ inline void segment_thresholds(int32_t seg_cfg, int32_t heavy_cfg, int64_t O, int32_t num_tiles,
                                                   int32_t needs_grad, uint32_t* seg_out, uint32_t* heavy_out) {
  if (O < 0) O = 0;
  int64_t heavy = heavy_cfg > 0 ? (int64_t)heavy_cfg : 120 + O / 2900;
  if (heavy_cfg <= 0 && heavy < 512) heavy = 512;
  int64_t seg;
  if (seg_cfg > 0) {
    seg = seg_cfg;
  } else {
    if (needs_grad) {
      const int64_t per = 6 * (int64_t)(num_tiles > 0 ? num_tiles : 1);
      seg = (O / per + 3) & ~(int64_t)3;
      seg = seg < 64 ? 64 : (seg > 256 ? 256 : seg);
    } else {
      seg = (heavy / 2) & ~(int64_t)3;
      if (seg < 256) seg = 256;
    }
    seg = (seg + 3) & ~(int64_t)3;
    if (seg < 4) seg = 4;
  }
  *seg_out = (uint32_t)seg;
  *heavy_out = (uint32_t)(heavy > seg ? heavy : seg);
}

__global__ __launch_bounds__(256) void segment_plan_kernel(const uint32_t* __restrict__ tile_range, int num_tiles,
                                                           int32_t seg_cfg, int32_t heavy_cfg, int32_t needs_grad,
                                                           int64_t O, const uint32_t* __restrict__ O_dev,
                                                           uint32_t capacity, uint32_t heavy_capacity,
                                                           uint32_t* __restrict__ tile_seg,
                                                           uint32_t* __restrict__ seg_desc,
                                                           uint32_t* __restrict__ seg_total,
                                                           uint32_t* __restrict__ tile_order) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  // K6's launch order (see GSR_TILE_CLASSES): the block counts its tiles per (band, class) in LDS first, so that the
  // global counters take one atomic per non-empty (band, class) of the block instead of one per tile
  __shared__ uint32_t s_count[8 * GSR_TILE_CLASSES], s_base[8 * GSR_TILE_CLASSES];
  static_assert(8 * GSR_TILE_CLASSES == 256, "one LDS counter per thread of the block");
  s_count[threadIdx.x] = 0u;
  __syncthreads();
  bool filed = false;
  uint32_t key = 0u, local = 0u;
  if (t < num_tiles) {
    uint32_t seg_pairs, heavy_min;
    const int64_t O_now = O_dev ? (int64_t)*O_dev : O;
    segment_thresholds(seg_cfg, heavy_cfg, O_now, num_tiles, needs_grad, &seg_pairs, &heavy_min);
    const uint32_t a = tile_range[2 * t], len = tile_range[2 * t + 1] - a;
    // a heavy tile's segments are also FORWARD work units (alpha-product pass + a prologue over the preceding segments):
    // about 256 pairs each and at most ~128 per tile; a long tile's segments are only checkpoints: seg_pairs each
    bool heavy = len > heavy_min;
    const uint32_t seg_heavy = max(seg_pairs, min(256u, (heavy_min / 2u) & ~3u));
    const uint32_t seg_t = heavy ? max(seg_heavy, (((len + 127u) / 128u) + 3u) & ~3u) : seg_pairs;
    uint32_t n = len > seg_pairs ? (len + seg_t - 1) / seg_t : 0u;
    uint32_t at = 0u;
    if (n) {
      at = atomicAdd(seg_total, n);
      if (at + n > capacity) {                       // cannot happen with the host's bound; stay in range regardless:
        for (uint32_t j = at; j < capacity; ++j) {   // the tile stays unsegmented and its reserved slots become empty segments
          uint32_t* d = seg_desc + 4 * (size_t)j;
          d[0] = (uint32_t)t; d[1] = a; d[2] = a; d[3] = 0u;
        }
        n = 0u;
      }
    }
    // the segments of heavy tiles are also listed compactly (behind the tile table): the forward passes that only
    // concern them launch one block per entry of that list instead of one per segment of the frame
    if (n && heavy) {
      const uint32_t hat = atomicAdd(seg_total + 1, n);
      if (hat + n <= heavy_capacity) {
        uint32_t* list = tile_seg + 2 * (size_t)num_tiles;
        for (uint32_t j = 0; j < n; ++j) list[hat + j] = at + j;
      } else {                  // cannot happen with the host's bound; stay in range regardless: the one-wave walk with
        heavy = false;          // checkpoints takes the tile and the list's tail holds "no segment"
        uint32_t* list = tile_seg + 2 * (size_t)num_tiles;
        for (uint32_t j = hat; j < heavy_capacity; ++j) list[j] = 0xFFFFFFFFu;
      }
    }
    tile_seg[2 * t] = at;
    tile_seg[2 * t + 1] = n | ((n && heavy) ? GSR_SEG_HEAVY : 0u);
    if (tile_order && !(n && heavy)) {               // (heavy tiles are composited by the segment passes)
      const uint32_t mean = (uint32_t)max((int64_t)1, O_now / num_tiles);
      const uint32_t cls = min((uint32_t)(GSR_TILE_CLASSES - 1), len * 8u / mean);
      key = (uint32_t)gsr_tile_band(t, num_tiles) * GSR_TILE_CLASSES + cls;
      local = atomicAdd(&s_count[key], 1u);
      filed = true;
    }
    for (uint32_t j = 0; j < n; ++j) {
      uint32_t* d = seg_desc + 4 * (size_t)(at + j);
      d[0] = (uint32_t)t; d[1] = a + j * seg_t; d[2] = min(a + (j + 1) * seg_t, a + len); d[3] = j;
    }
  }
  __syncthreads();
  if (s_count[threadIdx.x])                          // thread i owns (band, class) i: GSR_SEG_CLASS_COUNT(band) + class = 16 + i
    s_base[threadIdx.x] = atomicAdd(seg_total + GSR_SEG_CLASS_COUNT(0) + threadIdx.x, s_count[threadIdx.x]);
  __syncthreads();
  if (filed)
    tile_order[(size_t)key * (size_t)gsr_tile_band_stride(num_tiles) + s_base[key] + local] = (uint32_t)t;
}

inline SegDev to_segdev(const GsrSegmentsC* sg) {
  SegDev d;
  if (sg) {
    d.tile_seg = sg->tile_seg; d.seg_desc = sg->seg_desc; d.seg_total = sg->seg_total; d.seg_P = sg->seg_P;
    d.tile_order = sg->tile_order;
    d.seg_TC = reinterpret_cast<float4*>(sg->seg_TC); d.seg_last = sg->seg_last; d.seg_median = sg->seg_median;
  } else {
    d.tile_seg = nullptr; d.seg_desc = nullptr; d.seg_total = nullptr; d.seg_P = nullptr; d.seg_TC = nullptr;
    d.tile_order = nullptr;
    d.seg_last = nullptr; d.seg_median = nullptr;
  }
  return d;
}

inline bool seg_ok(const GsrSegmentsC* sg, bool median) {
  if (!sg) return true;
  if (sg->capacity <= 0 || sg->heavy_capacity < 0 || sg->heavy_capacity > sg->capacity) return false;
  return sg->tile_seg && sg->seg_desc && sg->seg_total && sg->seg_P && sg->seg_TC && sg->seg_last &&
         (!median || sg->seg_median);
}

}  // namespace

extern "C" {

#ifdef GSR_K6_TRACE
int gsr_debug_set_k6_trace(uint64_t* buffer) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_k6_trace), &buffer, sizeof(buffer)) == hipSuccess ? 0 : -1;
}
#endif

int gsr_segment_thresholds(int32_t seg_pairs_cfg, int32_t heavy_min_cfg, int64_t O, int32_t num_tiles, int32_t needs_grad,
                           int32_t* seg_pairs_out, int32_t* heavy_min_out) {
  if (!seg_pairs_out || !heavy_min_out || num_tiles <= 0) return GSR_ERR_INVALID_ARGUMENT;
  uint32_t seg, heavy;
  segment_thresholds(seg_pairs_cfg, heavy_min_cfg, O, num_tiles, needs_grad, &seg, &heavy);
  *seg_pairs_out = (int32_t)seg;
  *heavy_min_out = (int32_t)heavy;
  return GSR_OK;
}

int64_t gsr_segment_capacity(int64_t O, int32_t O_is_bound, int32_t seg_pairs_cfg, int32_t heavy_min_cfg, int32_t num_tiles,
                             int32_t needs_grad) {
  if (O <= 0 || num_tiles <= 0) return 0;
  // a tile of len pairs (len > seg) yields at most ceil(len / seg) <= len / seg + 1 segments, and at most
  // min(tiles, O / (seg + 1)) tiles are that long
  if (!O_is_bound || seg_pairs_cfg > 0) {
    uint32_t seg, heavy;
    segment_thresholds(seg_pairs_cfg, heavy_min_cfg, O, num_tiles, needs_grad, &seg, &heavy);
    return O / seg + O / ((int64_t)seg + 1) + 1;
  }
  // only a bound on the pair count is known and the segment length follows the true count: for any count <= O the
  // automatic length is >= 64; while it is below 256 it is >= floor(count / (6 tiles)) >= 64, i.e. at most
  // 6 (1 + 1/64) segments per tile on average, plus one remainder per tile; at 256 the first formula applies
  const int64_t by_min = O / 64 + O / 65 + 1;
  const int64_t T = num_tiles;
  const int64_t small = 7 * T + T / 8 + 1;
  const int64_t by_rule = (small > O / 256 + T ? small : O / 256 + T) + 1;
  return by_min < by_rule ? by_min : by_rule;
}

int64_t gsr_segment_heavy_capacity(int64_t O, int32_t O_is_bound, int32_t seg_pairs_cfg, int32_t heavy_min_cfg,
                                   int32_t num_tiles, int32_t needs_grad) {
  if (O <= 0 || num_tiles <= 0) return 0;
  // a heavy tile (len > heavy) is cut into pieces of at least seg_heavy = max(seg, min(256, heavy / 2)) pairs
  uint32_t seg, heavy;
  segment_thresholds(seg_pairs_cfg, heavy_min_cfg, O, num_tiles, needs_grad, &seg, &heavy);
  if (O_is_bound && heavy_min_cfg <= 0) heavy = 512;                 // the automatic threshold of any smaller count
  if (O_is_bound && seg_pairs_cfg <= 0) seg = needs_grad ? 64 : 256;
  uint32_t piece = (heavy / 2u) & ~3u;
  if (piece > 256u) piece = 256u;
  if (piece < seg) piece = seg;
  if (piece < 1u) piece = 1u;
  return O / piece + O / ((int64_t)heavy + 1) + 1;
}

int gsr_segment_plan(const uint32_t* tile_range, int32_t num_tiles, int32_t seg_pairs_cfg, int32_t heavy_min_cfg,
                     int32_t needs_grad, int64_t O, const uint32_t* O_dev, int64_t capacity, int64_t heavy_capacity,
                     uint32_t* tile_seg_out, uint32_t* seg_desc_out, uint32_t* seg_total_out, uint32_t* tile_order_out,
                     void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (num_tiles <= 0 || capacity <= 0 || capacity > 0x7fffffffll || (O < 0 && !O_dev)) return GSR_ERR_INVALID_ARGUMENT;
  if (heavy_capacity < 0 || heavy_capacity > capacity) return GSR_ERR_INVALID_ARGUMENT;
  if (seg_pairs_cfg > 0 && heavy_min_cfg > 0 && heavy_min_cfg < seg_pairs_cfg) return GSR_ERR_INVALID_ARGUMENT;
  if (!tile_range || !tile_seg_out || !seg_desc_out || !seg_total_out) return GSR_ERR_INVALID_ARGUMENT;
  segment_plan_kernel<<<(num_tiles + 255) / 256, 256, 0, stream>>>(tile_range, num_tiles, seg_pairs_cfg, heavy_min_cfg,
                                                                  needs_grad, O, O_dev, (uint32_t)capacity,
                                                                  (uint32_t)heavy_capacity, tile_seg_out, seg_desc_out,
                                                                  seg_total_out, tile_order_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_composite_forward(const float* rec, const uint32_t* sorted_rank, const uint32_t* sorted_inst,
                          const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                          const GsrRasterParamsC* params_host, float* image_out, float* final_T_out, int32_t* last_out,
                          float* median_depth_out, float* vis_partial_out, float* pair_vis_out,
                          const GsrSegmentsC* segments_host, int32_t prefetch_rows, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16 || C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (!tile_range || !image_out || !final_T_out || !last_out) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16, nt = tx * ty;
  const GsrRasterParams rp = to_params(params_host);
  const bool vis = vis_partial_out != nullptr, med = median_depth_out != nullptr;
  if (vis && !pair_vis_out) return GSR_ERR_INVALID_ARGUMENT;
  if (!seg_ok(segments_host, med)) return GSR_ERR_INVALID_ARGUMENT;
  const SegDev seg = to_segdev(segments_host);
  const int cap = segments_host ? (int)segments_host->heavy_capacity : 0;   // blocks of the heavy-tile passes
  const int tb = seg.tile_order ? 8 * gsr_tile_band_stride(nt) : nt;   // tile blocks (ordered form: a band's capacity per XCD)
#define GSR_LAUNCH_TILES(CC, VV, MM, PP)                                                                               \
  composite_fwd_kernel<CC, VV, MM, PP><<<tb + cap, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, tile_range, W, H,   \
                                                                    tx, nt, rp, image_out, final_T_out, last_out,      \
                                                                    median_depth_out, vis_partial_out, pair_vis_out,   \
                                                                    seg, tb)
#define GSR_LAUNCH_FWD2(CC, VV, MM, PP)                                                                                \
  do {                                                                                                                 \
    GSR_LAUNCH_TILES(CC, VV, MM, PP);                                                                                  \
    if (cap) {                                                                                                         \
      seg_composite_kernel<CC, VV, MM, PP><<<cap, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, tile_range, W, H,    \
                                                                   tx, nt, rp, vis_partial_out, pair_vis_out, seg);    \
      seg_combine_kernel<CC, MM><<<nt, 64, 0, stream>>>(W, H, tx, nt, image_out, final_T_out, last_out,                \
                                                        median_depth_out, seg);                                        \
    }                                                                                                                  \
  } while (0)
#define GSR_LAUNCH_FWD(CC, VV, MM)                 \
  do {                                             \
    if (prefetch_rows) GSR_LAUNCH_FWD2(CC, VV, MM, true); \
    else GSR_LAUNCH_FWD2(CC, VV, MM, false);       \
  } while (0)
#define GSR_DISPATCH_FWD(CC)                                       \
  do {                                                             \
    if (vis && med) GSR_LAUNCH_FWD(CC, true, true);                \
    else if (vis) GSR_LAUNCH_FWD(CC, true, false);                 \
    else if (med) GSR_LAUNCH_FWD(CC, false, true);                 \
    else GSR_LAUNCH_FWD(CC, false, false);                         \
  } while (0)
  if (C == 1) GSR_DISPATCH_FWD(1);
  else if (C == 2) GSR_DISPATCH_FWD(2);
  else GSR_DISPATCH_FWD(3);
#undef GSR_DISPATCH_FWD
#undef GSR_LAUNCH_FWD
#undef GSR_LAUNCH_FWD2
#undef GSR_LAUNCH_TILES
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_composite_backward(const float* rec, const uint32_t* sorted_rank, const uint32_t* sorted_inst,
                           const float* pair_vis, const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                           const GsrRasterParamsC* params_host, const float* final_T, const int32_t* last,
                           const float* dL_dimage, const float* image, float* partial_out,
                           const GsrSegmentsC* segments_host, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16 || C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (!tile_range || !final_T || !last || !dL_dimage) return GSR_ERR_INVALID_ARGUMENT;
  if (!seg_ok(segments_host, false) || (segments_host && !image)) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16, nt = tx * ty;
  const GsrRasterParams rp = to_params(params_host);
  const SegDev seg = to_segdev(segments_host);
  // segment blocks: rounded up to the XCD grouping of gsr_xcd_group_remap (blocks past seg_total return)
  const int seg_round = 8 << GSR_K7_SEG_GROUP_LOG2;
  const uint32_t seg_blocks = segments_host ? (uint32_t)((segments_host->capacity + seg_round - 1) / seg_round * seg_round) : 0u;
  const uint32_t seg_cap = (uint32_t)(segments_host ? segments_host->capacity : 0);
  const int grid = 8 * gsr_tile_band_stride(nt) + (int)seg_blocks;
#if GSR_K7_WINDOWS
  if (C == 1) composite_bwd_win_kernel<1><<<grid, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, image, partial_out, seg, seg_cap, seg_blocks);
  else if (C == 2) composite_bwd_win_kernel<2><<<grid, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, image, partial_out, seg, seg_cap, seg_blocks);
  else composite_bwd_win_kernel<3><<<grid, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, image, partial_out, seg, seg_cap, seg_blocks);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
#endif
  if (C == 1) composite_bwd_kernel<1><<<grid, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, image, partial_out, seg, seg_cap, seg_blocks);
  else if (C == 2) composite_bwd_kernel<2><<<grid, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, image, partial_out, seg, seg_cap, seg_blocks);
  else composite_bwd_kernel<3><<<grid, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, image, partial_out, seg, seg_cap, seg_blocks);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
