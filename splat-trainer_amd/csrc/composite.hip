// K6 alpha-composite forward and K7 alpha-composite backward (per-pixel reverse walk).
//
// Mapping: ONE wave64 owns one 16x16 tile; lane l owns 4 pixels, one in each 8x8 quadrant of the tile, at
//   (8*(p&1) + (l&7), 8*(p>>1) + (l>>3)), p = 0..3.  A small splat touches 1-2 quadrants, and a quadrant no lane
//   needs is skipped by a wave-uniform branch (s_cbranch_execz), so the per-pixel work follows the footprint.
//   * the tile's depth-sorted splat list is staged through LDS 64 records at a time (one coalesced index
//     load + one 48-byte record gather per lane), then read back with wave-uniform (broadcast) ds_reads;
//   * a single-wave workgroup needs no cross-wave synchronisation, and every per-splat reduction over the
//     tile's 256 pixels is 4 in-register adds + one DPP wave reduction (row_shr / row_bcast) -- i.e. the
//     cross-lane cost is paid once per 256 pixels, not once per 64;
//   * no float atomics: each (tile, splat) pair owns one slot of a partial buffer, written by lane 63 after
//     the fixed-order wave reduction and summed per splat in id order afterwards -> bit-reproducible
//     gradients and densification heuristics;
//   * the forward pass records, per (tile, splat) pair, the visibility partial sum_px T*alpha; a pair whose
//     partial is zero touched no pixel, so the backward pass skips it without evaluating a single pixel.
// MFMA is deliberately not used: there is no dense contraction on this path.
#include "gsr_device.h"
#include "../../include/gsplat_hip.h"

namespace {

inline GsrRasterParams to_params(const GsrRasterParamsC* c) {
  GsrRasterParams rp;
  __builtin_memcpy(&rp, c, sizeof(rp));
  return rp;
}

// Shared by forward and backward so both take bit-identical contribute / skip decisions: every operation is
// an explicit round-to-nearest intrinsic, so the compiler cannot contract the two kernels differently.
struct PixelEval {
  float q, G, a_raw, alpha;
  bool hit;
};

__device__ __forceinline__ PixelEval eval_pixel(float dx, float dy, float A, float B2, float C, float op,
                                                float qmax, float cmax, float thr) {
  PixelEval e;
  // q = A dx^2 + 2B dx dy + C dy^2,   B2 = 2B
  float t0 = __fmul_rn(A, dx);
  float t1 = __fmul_rn(B2, dy);
  float t2 = __fmul_rn(C, dy);
  e.q = __fmaf_rn(t0, dx, __fmaf_rn(t1, dx, __fmul_rn(t2, dy)));
  e.hit = false;
  e.G = 0.f; e.a_raw = 0.f; e.alpha = 0.f;
  if (e.q <= qmax) {
    e.G = __expf(__fmul_rn(-0.5f, e.q));
    e.a_raw = __fmul_rn(op, e.G);
    e.alpha = fminf(cmax, e.a_raw);
    e.hit = e.alpha >= thr;
  }
  return e;
}

template <int C, bool VIS, bool MEDIAN>
__global__ __launch_bounds__(64) void composite_fwd_kernel(const float* __restrict__ rec,
                                                           const uint32_t* __restrict__ sorted_rank,
                                                           const uint32_t* __restrict__ sorted_inst,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           float* __restrict__ image, float* __restrict__ final_T,
                                                           int* __restrict__ last, float* __restrict__ median,
                                                           float* __restrict__ vis_partial) {
  __shared__ float4 s_rec[64][3];
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

  for (uint32_t base = start; base < end; base += 64) {
    if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
    const uint32_t n = min(64u, end - base);
    uint32_t my_inst = 0;
    __syncthreads();
    if ((uint32_t)lane < n) {
      const uint32_t k = sorted_rank[base + lane];
      const float4* r = reinterpret_cast<const float4*>(rec + (size_t)GSR_REC_FLOATS * k);
      s_rec[lane][0] = r[0];
      s_rec[lane][1] = r[1];
      if (C > 1) s_rec[lane][2] = r[2];
      if (VIS) my_inst = sorted_inst[base + lane];
    }
    __syncthreads();
    float my_vis = 0.f;
    for (uint32_t j = 0; j < n; ++j) {
      if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
      const float4 r0 = s_rec[j][0];
      const float4 r1 = s_rec[j][1];
      float f[3] = {r1.w, 0.f, 0.f};
      if (C > 1) { const float4 r2 = s_rec[j][2]; f[1] = r2.x; f[2] = r2.y; }
      const float dxa = fx0 - r0.x, dya = fy0 - r0.y;
      const float B2 = r0.w + r0.w;
      float wsum = 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (!done[p]) {
          const float dx = (p & 1) ? dxa + 8.f : dxa;
          const float dy = (p >> 1) ? dya + 8.f : dya;
          PixelEval e = eval_pixel(dx, dy, r0.z, B2, r1.x, r1.y, rp.q_max, rp.clamp_max_alpha, rp.alpha_threshold);
          if (e.hit) {
            const float w = __fmul_rn(e.alpha, T[p]);
#pragma unroll
            for (int c = 0; c < C; ++c) col[p][c] = __fmaf_rn(w, f[c], col[p][c]);
            wsum += w;
            T[p] = __fmul_rn(T[p], 1.f - e.alpha);
            lastc[p] = (int)(base - start + j) + 1;
            if (MEDIAN && med[p] == 0.f && T[p] < 0.5f) med[p] = r1.z;
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
    }
    if (VIS && (uint32_t)lane < n && my_vis > 0.f) vis_partial[my_inst] = my_vis;
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
                                                           const float* __restrict__ vis_partial,
                                                           const uint32_t* __restrict__ tile_range, int W, int H,
                                                           int tiles_x, int num_tiles, GsrRasterParams rp,
                                                           const float* __restrict__ final_T,
                                                           const int* __restrict__ last,
                                                           const float* __restrict__ dL_dimage,
                                                           float* __restrict__ partial) {
  __shared__ float4 s_rec[64][3];
  const int tile = gsr_xcd_remap((int)blockIdx.x, num_tiles);
  const int lane = (int)threadIdx.x;
  const int tx = tile % tiles_x, ty = tile / tiles_x;
  const int px0 = tx * 16 + (lane & 7), py0 = ty * 16 + (lane >> 3);
  const float fx0 = (float)px0 + 0.5f, fy0 = (float)py0 + 0.5f;
  const uint32_t start = tile_range[2 * tile];

  float T[4], g[4][3], acc[4][3];
  int lastc[4];
  int tile_last = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int px = px0 + 8 * (p & 1), py = py0 + 8 * (p >> 1);
    T[p] = 1.f; lastc[p] = 0;
    g[p][0] = g[p][1] = g[p][2] = 0.f;
    acc[p][0] = acc[p][1] = acc[p][2] = 0.f;
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
  if (tile_last == 0) return;

  for (int cbase = ((tile_last - 1) >> 6) << 6; cbase >= 0; cbase -= 64) {
    const int n = min(64, tile_last - cbase);
    uint32_t my_inst = 0;
    bool my_flag = false;
    __syncthreads();
    if (lane < n) {
      const uint32_t i = start + (uint32_t)cbase + (uint32_t)lane;
      my_inst = sorted_inst[i];
      my_flag = vis_partial[my_inst] > 0.f;
      if (my_flag) {
        const uint32_t k = sorted_rank[i];
        const float4* r = reinterpret_cast<const float4*>(rec + (size_t)GSR_REC_FLOATS * k);
        s_rec[lane][0] = r[0];
        s_rec[lane][1] = r[1];
        if (C > 1) s_rec[lane][2] = r[2];
      }
    }
    __syncthreads();
    const uint64_t flags = __ballot(my_flag);
    for (int j = n - 1; j >= 0; --j) {
      if (!((flags >> j) & 1ull)) continue;                 // this (tile, splat) pair touched no pixel
      const int pos = cbase + j;
      const float4 r0 = s_rec[j][0];
      const float4 r1 = s_rec[j][1];
      float f[3] = {r1.w, 0.f, 0.f};
      if (C > 1) { const float4 r2 = s_rec[j][2]; f[1] = r2.x; f[2] = r2.y; }
      const float dxa = fx0 - r0.x, dya = fy0 - r0.y;
      const float A = r0.z, B = r0.w, Cc = r1.x, op = r1.y;
      const float B2 = B + B;
      float du = 0.f, dv = 0.f, dA = 0.f, dB = 0.f, dC = 0.f, dop = 0.f, prune = 0.f, split = 0.f;
      float df[3] = {0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (pos < lastc[p]) {
          const float dx = (p & 1) ? dxa + 8.f : dxa;
          const float dy = (p >> 1) ? dya + 8.f : dya;
          PixelEval e = eval_pixel(dx, dy, A, B2, Cc, op, rp.q_max, rp.clamp_max_alpha, rp.alpha_threshold);
          if (e.hit) {
            const float inv = __builtin_amdgcn_rcpf(1.f - e.alpha);
            const float Tb = T[p] * inv;                    // transmittance in front of this splat
            T[p] = Tb;
            const float w = e.alpha * Tb;
            float gc = 0.f, gs = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
              gc += g[p][c] * f[c];
              gs += g[p][c] * acc[p][c];
              df[c] += w * g[p][c];
              acc[p][c] += f[c] * w;
            }
            const float dLda = Tb * gc - gs * inv;
            prune += fabsf(dLda) * e.alpha;
            if (e.a_raw <= rp.clamp_max_alpha) {
              dop += dLda * e.G;
              const float GdG = e.G * dLda * op;            // G * dL/dG
              const float dq = -0.5f * GdG;
              dA += dq * dx * dx;
              dB += dq * 2.f * dx * dy;
              dC += dq * dy * dy;
              const float gmx = GdG * (A * dx + B * dy);
              const float gmy = GdG * (B * dx + Cc * dy);
              du += gmx;
              dv += gmy;
              split += __builtin_amdgcn_sqrtf(gmx * gmx + gmy * gmy);
            }
          }
        }
      }
      // fixed-order wave reductions; totals land in lane 63, which owns the store
      du = gsr_wave_sum_to_lane63(du);
      dv = gsr_wave_sum_to_lane63(dv);
      dA = gsr_wave_sum_to_lane63(dA);
      dB = gsr_wave_sum_to_lane63(dB);
      dC = gsr_wave_sum_to_lane63(dC);
      dop = gsr_wave_sum_to_lane63(dop);
      prune = gsr_wave_sum_to_lane63(prune);
      split = gsr_wave_sum_to_lane63(split);
#pragma unroll
      for (int c = 0; c < C; ++c) df[c] = gsr_wave_sum_to_lane63(df[c]);
      const uint32_t inst_j = (uint32_t)__builtin_amdgcn_readlane((int)my_inst, j);
      if (lane == 63) {
        float4* out = reinterpret_cast<float4*>(partial + (size_t)GSR_PARTIAL_FLOATS * inst_j);
        out[0] = make_float4(du, dv, dA, dB);
        out[1] = make_float4(dC, dop, prune, split);
        out[2] = make_float4(df[0], df[1], df[2], 0.f);
      }
    }
  }
}

}  // namespace

extern "C" {

int gsr_composite_forward(const float* rec, const uint32_t* sorted_rank, const uint32_t* sorted_inst,
                          const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                          const GsrRasterParamsC* params_host, float* image_out, float* final_T_out, int32_t* last_out,
                          float* median_depth_out, float* vis_partial_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16 || C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (!tile_range || !image_out || !final_T_out || !last_out) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16, nt = tx * ty;
  const GsrRasterParams rp = to_params(params_host);
  const bool vis = vis_partial_out != nullptr, med = median_depth_out != nullptr;
#define GSR_LAUNCH_FWD(CC, VV, MM)                                                                                  \
  composite_fwd_kernel<CC, VV, MM><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, tile_range, W, H, tx, nt, \
                                                          rp, image_out, final_T_out, last_out, median_depth_out,  \
                                                          vis_partial_out)
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
                           const float* vis_partial, const uint32_t* tile_range, int32_t W, int32_t H, int32_t C,
                           const GsrRasterParamsC* params_host, const float* final_T, const int32_t* last,
                           const float* dL_dimage, float* partial_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!params_host || W <= 0 || H <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (params_host->tile_size != 16 || C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (!tile_range || !final_T || !last || !dL_dimage) return GSR_ERR_INVALID_ARGUMENT;
  const int tx = (W + 15) / 16, ty = (H + 15) / 16, nt = tx * ty;
  const GsrRasterParams rp = to_params(params_host);
  if (C == 1) composite_bwd_kernel<1><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, vis_partial, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, partial_out);
  else if (C == 2) composite_bwd_kernel<2><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, vis_partial, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, partial_out);
  else composite_bwd_kernel<3><<<nt, 64, 0, stream>>>(rec, sorted_rank, sorted_inst, vis_partial, tile_range, W, H, tx, nt, rp, final_T, last, dL_dimage, partial_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
