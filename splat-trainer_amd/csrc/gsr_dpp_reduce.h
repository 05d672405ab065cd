// Fused in-place wave64 cross-lane sums used by the composite forward pass (visibility of four pairs at once):
//   gsr_swap32_add / gsr_swap16_add   v_permlane{32,16}_swap + add: two registers fold into one, each half (quarter) of
//                                     the lanes ends up holding the partial sum of one of the two values
//   gsr_row_sum_to_lane15             v_add_f32_dpp vN, vN, vN row_shr:1/2/4/8 -- one VALU op per step (the
//                                     __builtin_amdgcn_update_dpp form compiles to v_mov_b32_dpp + v_add_f32, two ops);
//                                     a lane whose DPP source is out of range is not written and keeps its own partial
//                                     sum, which is the "+ 0" the tree needs; s_nop covers the DPP read-after-VALU-write
//                                     hazard.  Totals land in lane 15 of every row.
// Fixed association order => bit-reproducible.  (The backward pass reduces through LDS instead, see composite.hip; its
// former register-only 12-value transposing tree was removed with it.)
#pragma once

__device__ __forceinline__ float gsr_swap32_add(float a, float b) {
  typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
  v2u_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ float gsr_swap16_add(float a, float b) {
  typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
  v2u_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  return __uint_as_float(r.x) + __uint_as_float(r.y);
}

__device__ __forceinline__ float gsr_row_sum_to_lane15(float a) {
  asm volatile(
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n"
      "s_nop 1\n"
      "v_add_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n"
      : "+v"(a));
  return a;
}
