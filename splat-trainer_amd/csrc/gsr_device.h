// Device-side helpers for gfx950 (wave64).  No portability layer: DPP row shifts / row broadcasts are the
// CDNA cross-lane path, ballots are 64-bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_math.h"

#define GSR_WAVE 64

#define GSR_OK 0
#define GSR_ERR_INVALID_ARGUMENT -1
#define GSR_ERR_WORKSPACE_TOO_SMALL -2
#define GSR_ERR_LAUNCH_FAILED -3
#define GSR_ERR_UNSUPPORTED -4

#define GSR_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return GSR_ERR_LAUNCH_FAILED;     \
  } while (0)

// Orders this wave's LDS accesses across its lanes (one wave per workgroup exchanges data through LDS without
// s_barrier): a wavefront-scope fence + wave barrier.  Emits no instruction -- LDS operations of one wave execute in
// order -- but keeps the compiler from moving the reads above the writes (or the next writes above the reads).
__device__ __forceinline__ void gsr_wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int gsr_lane() { return (int)(threadIdx.x & 63); }

// v + (DPP-moved v); lanes without a source read 0 (bound_ctrl).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float gsr_dpp_add(float v) {
  int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, true);
  return v + __int_as_float(moved);
}

// Sum over the 64 lanes in a FIXED association order (bit-reproducible); the total lands in lane 63.
__device__ __forceinline__ float gsr_wave_sum_to_lane63(float v) {
  v = gsr_dpp_add<0x111, 0xf, 0xf>(v);   // row_shr:1
  v = gsr_dpp_add<0x112, 0xf, 0xf>(v);   // row_shr:2
  v = gsr_dpp_add<0x114, 0xf, 0xf>(v);   // row_shr:4
  v = gsr_dpp_add<0x118, 0xf, 0xf>(v);   // row_shr:8   -> lane 15 of each row holds its row sum
  v = gsr_dpp_add<0x142, 0xa, 0xf>(v);   // row_bcast:15 into rows 1,3
  v = gsr_dpp_add<0x143, 0xc, 0xf>(v);   // row_bcast:31 into rows 2,3 -> lane 63 holds the total
  return v;
}

__device__ __forceinline__ float gsr_wave_sum(float v) {   // total broadcast to every lane (via SGPR)
  v = gsr_wave_sum_to_lane63(v);
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ uint32_t gsr_wave_sum_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// inclusive prefix sum across the wave
__device__ __forceinline__ uint32_t gsr_wave_scan_incl_u32(uint32_t v) {
  const int lane = gsr_lane();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t n = __shfl_up(v, o, 64);
    if (lane >= o) v += n;
  }
  return v;
}

__device__ __forceinline__ int gsr_mbcnt(uint64_t mask) {  // number of set bits of mask below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// Depth -> sortable u32 key.  The IEEE bit pattern of a positive float is monotone (sign bit set keeps negatives, which
// the cull never lets through, below); `bias` = key of the near plane makes the keys of a frame start at 0, so that
// they span only bits(far) - bits(near) (27 bits for 0.1 .. 100) and the radix sort needs fewer passes.  Depths outside
// [near, far] -- only possible for a caller-made depth tensor -- clamp to the ends (order among them: by index).
__device__ __forceinline__ uint32_t gsr_depth_key(float depth, uint32_t bias, uint32_t max_key) {
  const uint32_t b = __float_as_uint(depth);
  const uint32_t k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  const uint32_t rel = k > bias ? k - bias : 0u;
  return rel < max_key ? rel : max_key;
}

// Bijective remap of a linear block id so that the blocks dealt to one XCD (ids congruent mod 8 under the
// observed round-robin placement; speed only, never correctness) cover a contiguous range of work items.
__device__ __forceinline__ int gsr_xcd_remap(int bid, int n) {
  int q = n >> 3, r = n & 7;
  int xcd = bid & 7, within = bid >> 3;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + within;
}

// Grouped variant for work lists whose length is only known on the device: runs of 2^log2_group consecutive items
// stay on one XCD (they share per-tile data in that XCD's L2) while the runs themselves are dealt round-robin, so every
// XCD still sees a uniform sample of the image.  Bijective on [0, 8 * 2^log2_group * k); the launch rounds its grid up
// to that multiple and blocks whose item falls past the end of the list return.
__device__ __forceinline__ uint32_t gsr_xcd_group_remap(uint32_t bid, uint32_t log2_group) {
  const uint32_t xcd = bid & 7u, within = bid >> 3;
  return ((((within >> log2_group) << 3) | xcd) << log2_group) | (within & ((1u << log2_group) - 1u));
}
