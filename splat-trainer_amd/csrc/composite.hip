// K6 alpha-composite forward and K7 alpha-composite backward (per-pixel reverse walk).
//
// Mapping: ONE wave64 owns one 16x16 tile; lane l owns 4 pixels, one in each 8x8 quadrant of the tile, at
//   (8*(p&1) + (l&7), 8*(p>>1) + (l>>3)), p = 0..3.  A small splat touches 1-2 quadrants, and a quadrant no lane
//   needs is skipped by a wave-uniform branch (s_cbranch_execz), so the per-pixel work follows the footprint.
//   * the tile's depth-sorted splat list is walked with WAVE-UNIFORM addresses, so the 48-byte record of the
//     current splat arrives through the scalar data cache into SGPRs (s_load_dwordx4/x8): no VGPRs, no LDS, no
//     barrier; the next record is prefetched while the current one is evaluated;
//   * every per-splat reduction over the tile's 256 pixels is 4 in-register adds + one DPP wave reduction
//     (row_shr / row_bcast) -- i.e. the cross-lane cost is paid once per 256 pixels, not once per 64;
//   * no float atomics: each (tile, splat) pair owns one slot of a partial buffer, written by lane 63 after
//     the fixed-order wave reduction and summed per splat in id order afterwards -> bit-reproducible
//     gradients and densification heuristics;
//   * the forward pass records, per (tile, splat) pair, the visibility partial sum_px T*alpha; a pair whose
//     partial is zero touched no pixel, so the backward pass skips it without evaluating a single pixel.
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

// Shared by forward and backward so both take bit-identical contribute / skip decisions: every operation is
// an explicit round-to-nearest intrinsic, so the compiler cannot contract the two kernels differently.
__device__ __forceinline__ float eval_q(float dx, float dy, float A, float B2, float C) {
  // q = A dx^2 + 2B dx dy + C dy^2,   B2 = 2B
  return __fmaf_rn(__fmul_rn(A, dx), dx, __fmaf_rn(__fmul_rn(B2, dy), dx, __fmul_rn(__fmul_rn(C, dy), dy)));
}
__device__ __forceinline__ float eval_G(float q) { return __expf(__fmul_rn(-0.5f, q)); }

struct Splat {            // one depth-ordered record, wave-uniform (lives in SGPRs)
  float u, v, A, B, C, op, depth, f0, f1, f2;
};

// i is wave-uniform: the index load and the three 16-byte record loads become scalar-cache loads.
template <int C>
__device__ __forceinline__ Splat load_splat(const float* __restrict__ rec, const uint32_t* __restrict__ sorted_rank,
                                            uint32_t i) {
  const uint32_t k = (uint32_t)__builtin_amdgcn_readfirstlane((int)sorted_rank[i]);
  const float4* r = reinterpret_cast<const float4*>(rec + (size_t)GSR_REC_FLOATS * k);
  const float4 r0 = r[0], r1 = r[1];
  Splat s;
  s.u = r0.x; s.v = r0.y; s.A = r0.z; s.B = r0.w; s.C = r1.x; s.op = r1.y; s.depth = r1.z; s.f0 = r1.w;
  s.f1 = 0.f; s.f2 = 0.f;
  if (C > 1) { const float4 r2 = r[2]; s.f1 = r2.x; s.f2 = r2.y; }
  return s;
}

template <int C, bool VIS, bool MEDIAN>
__global__ __launch_bounds__(64) void composite_fwd_kernel(const float* __restrict__ rec,
                                                           const uint32_t* __restrict__ sorted_rank,
                                                           const uint32_t* __restrict__ sorted_inst,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           float* __restrict__ image, float* __restrict__ final_T,
                                                           int* __restrict__ last, float* __restrict__ median,
                                                           float* __restrict__ vis_partial,
                                                           float* __restrict__ pair_vis) {
  const int tile = gsr_xcd_remap((int)blockIdx.x, num_tiles);
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f;
  const uint32_t start = tile_range[2 * tile], end = tile_range[2 * tile + 1];

  float T[4], col[4][3], med[4];
  int lastc[4];
  bool done[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    T[p] = 1.f; lastc[p] = 0; med[p] = 0.f;
    col[p][0] = col[p][1] = col[p][2] = 0.f;
    done[p] = !((px0 + 8 * (p & 1)) < W && (py0 + 8 * (p >> 1)) < H);
  }

  bool all_done = (start >= end);
  for (uint32_t base = start; base < end && !all_done; base += 64) {
    const uint32_t n = min(64u, end - base);
    float my_vis = 0.f;
    Splat nxt = load_splat<C>(rec, sorted_rank, base);
    for (uint32_t j = 0; j < n; ++j) {
      const Splat s = nxt;
      if (j + 1 < n) nxt = load_splat<C>(rec, sorted_rank, base + j + 1);     // prefetch (scalar loads)
      const float dxa = fx0 - s.u, dya = fy0 - s.v;
      const float B2 = s.B + s.B;
      float wsum = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float dx = (p & 1) ? dxa + 8.f : dxa;
        const float dy = (p >> 1) ? dya + 8.f : dya;
        const float q = eval_q(dx, dy, s.A, B2, s.C);
        if (!done[p] && q <= rp.q_max) {
          const float a_raw = __fmul_rn(s.op, eval_G(q));
          const float alpha = fminf(rp.clamp_max_alpha, a_raw);
          if (alpha >= rp.alpha_threshold) {
            const float w = __fmul_rn(alpha, T[p]);
            col[p][0] = __fmaf_rn(w, s.f0, col[p][0]);
            if (C > 1) col[p][1] = __fmaf_rn(w, s.f1, col[p][1]);
            if (C > 2) col[p][2] = __fmaf_rn(w, s.f2, col[p][2]);
            wsum += w;
            T[p] = __fmul_rn(T[p], 1.f - alpha);
            lastc[p] = (int)(base - start + j) + 1;
            if (MEDIAN && med[p] == 0.f && T[p] < 0.5f) med[p] = s.depth;
            if (T[p] < rp.T_eps) done[p] = true;
          }
        }
      }
      if (VIS) {
        if (__ballot(wsum > 0.f) != 0ull) {
          const float tot = gsr_wave_sum(wsum);
          if ((uint32_t)lane == j) my_vis = tot;
        }
      }
      if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) { all_done = true; break; }
    }
    if (VIS && (uint32_t)lane < n) {
      // sorted-position copy (read back coalesced by the backward pass) + per-instance copy (per-splat reduction)
      pair_vis[base + lane] = my_vis;
      if (my_vis > 0.f) vis_partial[sorted_inst[base + lane]] = my_vis;
    }
  }

#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int px = px0 + 8 * (p & 1), py = py0 + 8 * (p >> 1);
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
#pragma unroll
      for (int c = 0; c < C; ++c) image[pix * C + c] = col[p][c];
      final_T[pix] = T[p];
      last[pix] = lastc[p];
      if (MEDIAN) median[pix] = med[p];
    }
  }
}

template <int C>
__global__ __launch_bounds__(64) void composite_bwd_kernel(const float* __restrict__ rec,
                                                           const uint32_t* __restrict__ sorted_rank,
                                                           const uint32_t* __restrict__ sorted_inst,
                                                           const float* __restrict__ pair_vis,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           const float* __restrict__ final_T,
                                                           const int* __restrict__ last,
                                                           const float* __restrict__ dL_dimage,
                                                           float* __restrict__ partial) {
  const int tile = gsr_xcd_remap((int)blockIdx.x, num_tiles);
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f;
  const uint32_t start = tile_range[2 * tile];

  // per pixel: T behind the current splat, g = dL/dC, ga = g . (colour accumulated behind the current splat)
  float T[4], g[4][3], ga[4];
  int lastc[4];
  int tile_last = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int px = px0 + 8 * (p & 1), py = py0 + 8 * (p >> 1);
    T[p] = 1.f; lastc[p] = 0; ga[p] = 0.f;
    g[p][0] = g[p][1] = g[p][2] = 0.f;
    if (px < W && py < H) {
      const size_t pix = (size_t)py * W + px;
      T[p] = final_T[pix];
      lastc[p] = last[pix];
#pragma unroll
      for (int c = 0; c < C; ++c) g[p][c] = dL_dimage[pix * C + c];
    }
    tile_last = max(tile_last, lastc[p]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) tile_last = max(tile_last, __shfl_xor(tile_last, o, 64));
  tile_last = __builtin_amdgcn_readfirstlane(tile_last);
  if (tile_last == 0) return;

  for (int cbase = ((tile_last - 1) >> 6) << 6; cbase >= 0; cbase -= 64) {
    const int n = min(64, tile_last - cbase);
    // pairs that touched no pixel in the forward pass are skipped without evaluating anything
    const float pv = (lane < n) ? pair_vis[start + (uint32_t)cbase + (uint32_t)lane] : 0.f;
    uint64_t flags = __ballot(pv > 0.f);
    if (flags == 0ull) continue;
    int j = 63 - __builtin_clzll(flags);
    Splat nxt = load_splat<C>(rec, sorted_rank, start + (uint32_t)(cbase + j));
    uint32_t inst_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)sorted_inst[start + (uint32_t)(cbase + j)]);
    while (true) {
      const Splat s = nxt;
      const uint32_t inst_j = inst_nxt;
      const int pos = cbase + j;
      flags &= ~(1ull << j);
      const bool more = flags != 0ull;
      if (more) {                                             // prefetch the next contributing pair
        j = 63 - __builtin_clzll(flags);
        nxt = load_splat<C>(rec, sorted_rank, start + (uint32_t)(cbase + j));
        inst_nxt = (uint32_t)__builtin_amdgcn_readfirstlane((int)sorted_inst[start + (uint32_t)(cbase + j)]);
      }
      const float dxa = fx0 - s.u, dya = fy0 - s.v;
      const float B2 = s.B + s.B;
      float du = 0.f, dv = 0.f, dA = 0.f, dB = 0.f, dC = 0.f, dop = 0.f, prune = 0.f, split = 0.f;
      float df[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float dx = (p & 1) ? dxa + 8.f : dxa;
        const float dy = (p >> 1) ? dya + 8.f : dya;
        const float q = eval_q(dx, dy, s.A, B2, s.C);
        if (pos < lastc[p] && q <= rp.q_max) {
          const float G = eval_G(q);
          const float a_raw = __fmul_rn(s.op, G);
          const float alpha = fminf(rp.clamp_max_alpha, a_raw);
          if (alpha >= rp.alpha_threshold) {
            const float inv = __builtin_amdgcn_rcpf(1.f - alpha);
            const float Tb = T[p] * inv;                    // transmittance in front of this splat
            T[p] = Tb;
            const float w = alpha * Tb;
            float gc = g[p][0] * s.f0;
            if (C > 1) gc += g[p][1] * s.f1;
            if (C > 2) gc += g[p][2] * s.f2;
            df[0] += w * g[p][0];
            if (C > 1) df[1] += w * g[p][1];
            if (C > 2) df[2] += w * g[p][2];
            const float dLda = Tb * gc - ga[p] * inv;       // dC/dalpha = T c - (colour behind)/(1-alpha)
            ga[p] += gc * w;
            prune += fabsf(dLda) * alpha;
            if (a_raw <= rp.clamp_max_alpha) {
              dop += dLda * G;
              const float GdG = G * dLda * s.op;            // G * dL/dG
              const float dq = -0.5f * GdG;
              dA += dq * dx * dx;
              dB += dq * 2.f * dx * dy;
              dC += dq * dy * dy;
              const float gmx = GdG * (s.A * dx + s.B * dy);
              const float gmy = GdG * (s.B * dx + s.C * dy);
              du += gmx;
              dv += gmy;
              split += __builtin_amdgcn_sqrtf(gmx * gmx + gmy * gmy);
            }
          }
        }
      }
      // fixed-order fused-DPP wave reductions; totals land in lane 63, which owns the store
      if (C == 1) gsr_wave_sum9_to_lane63(du, dv, dA, dB, dC, dop, prune, split, df[0]);
      else if (C == 2) gsr_wave_sum10_to_lane63(du, dv, dA, dB, dC, dop, prune, split, df[0], df[1]);
      else gsr_wave_sum11_to_lane63(du, dv, dA, dB, dC, dop, prune, split, df[0], df[1], df[2]);
      if (lane == 63) {
        float4* out = reinterpret_cast<float4*>(partial + (size_t)GSR_PARTIAL_FLOATS * inst_j);
        out[0] = make_float4(du, dv, dA, dB);
        out[1] = make_float4(dC, dop, prune, split);
        out[2] = make_float4(df[0], df[1], df[2], 0.f);
      }
      if (!more) break;
    }
  }
}

}  // namespace

extern "C" {

int gsr_composite_forward(const float* rec, const uint32_t* sorted_rank, const uint32_t* sorted_inst,
                          const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                          const GsrRasterParamsC* params_host, float* image_out, float* final_T_out, int32_t* last_out,
                          float* median_depth_out, float* vis_partial_out, float* pair_vis_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16 || C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (!tile_range || !image_out || !final_T_out || !last_out) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16, nt = tx * ty;
  const GsrRasterParams rp = to_params(params_host);
  const bool vis = vis_partial_out != nullptr, med = median_depth_out != nullptr;
  if (vis && !pair_vis_out) return GSR_ERR_INVALID_ARGUMENT;
#define GSR_LAUNCH_FWD(CC, VV, MM)                                                                                  \
  composite_fwd_kernel<CC, VV, MM><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, tile_range, W, H, tx, nt, \
                                                          rp, image_out, final_T_out, last_out, median_depth_out,  \
                                                          vis_partial_out, pair_vis_out)
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
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_composite_backward(const float* rec, const uint32_t* sorted_rank, const uint32_t* sorted_inst,
                           const float* pair_vis, const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                           const GsrRasterParamsC* params_host, const float* final_T, const int32_t* last,
                           const float* dL_dimage, float* partial_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16 || C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (!tile_range || !final_T || !last || !dL_dimage) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16, nt = tx * ty;
  const GsrRasterParams rp = to_params(params_host);
  if (C == 1) composite_bwd_kernel<1><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, partial_out);
  else if (C == 2) composite_bwd_kernel<2><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, partial_out);
  else composite_bwd_kernel<3><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, pair_vis, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, partial_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
