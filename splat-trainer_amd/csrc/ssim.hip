// Fused SSIM forward / backward (the loss stage that produces dL/dimage for the composite backward).
// Replaces the CUDA-only ``fused_ssim`` package the reference imports (splat_trainer/trainer/trainer.py:17,112,450-462).
//
// One 32x32 output tile per 256-thread block and (batch, channel) plane.  The 42x42 input tile (5-pixel halo, zero
// outside the image) is staged in LDS once; the separable 11-tap Gaussian is applied as a horizontal pass into LDS
// (5 moment images x 42 rows x 32 columns) and a vertical pass in registers, 4 outputs per work item, so every input
// pixel is read from HBM once per tile and all 5 (forward) / 3 (backward) convolutions share the staging.
// Forward reads 2 and writes 3 planes, backward reads 5 and writes 1.
// The mean is reduced without atomics: per-block partial sums in a fixed order, then one block adds them in order.
#include "gsr_device.h"
#include "../../include/gsplat_hip.h"

namespace {

constexpr int TS = 32;              // output tile side (256 threads: 8 column groups / row groups of 4)
constexpr int HALO = 5;
constexpr int IN = TS + 2 * HALO;   // 42

struct Gauss11 {
  float w[11];
};

// img element (b, c, y, x) at b*sB + c*sC + y*sH + x*sW (so NCHW, channels_last and (H,W,C) views all work)
struct Strides {
  int64_t sB, sC, sH, sW;
};

__device__ __forceinline__ float block_sum_256(float v, float* s_red) {
  v = gsr_wave_sum(v);                                   // fixed-order DPP tree
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) s_red[wave] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

// Register blocking: a work item produces 4 adjacent outputs from 14 staged inputs, so an 11-tap pass costs 3.5 LDS
// reads per output and image instead of 11 (the first version of this kernel was LDS-read bound at ~90 reads/pixel).
template <bool TRAIN>
__global__ __launch_bounds__(256) void ssim_fwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                       Strides st1, Strides st2, int C, int H, int W, int crop,
                                                       float C1, float C2, float inv_count, Gauss11 g,
                                                       float* __restrict__ block_sums, float* __restrict__ dm_dmu1,
                                                       float* __restrict__ dm_dm11, float* __restrict__ dm_dm12) {
  __shared__ float s_x[IN][IN + 1], s_y[IN][IN + 1];
  __shared__ float s_h[5][IN][TS + 1];
  __shared__ float s_red[4];
  const int plane = blockIdx.z, b = plane / C, c = plane % C;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const float* p1 = img1 + b * st1.sB + c * st1.sC;
  const float* p2 = img2 + b * st2.sB + c * st2.sC;
  for (int i = threadIdx.x; i < IN * IN; i += 256) {
    const int ly = i / IN, lx = i % IN;
    const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    s_x[ly][lx] = in ? p1[gy * st1.sH + gx * st1.sW] : 0.f;
    s_y[ly][lx] = in ? p2[gy * st2.sH + gx * st2.sW] : 0.f;
  }
  __syncthreads();
  // horizontal pass: 42 rows x 8 groups of 4 columns
  for (int i = threadIdx.x; i < IN * (TS / 4); i += 256) {
    const int ly = i / (TS / 4), lx = (i % (TS / 4)) * 4;
    float x[14], y[14], xx[14], yy[14], xy[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) {
      x[k] = s_x[ly][lx + k]; y[k] = s_y[ly][lx + k];
      xx[k] = x[k] * x[k]; yy[k] = y[k] * y[k]; xy[k] = x[k] * y[k];
    }
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float a = 0.f, bb = 0.f, aa = 0.f, bb2 = 0.f, ab = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float w = g.w[k];
        a += w * x[o + k]; bb += w * y[o + k]; aa += w * xx[o + k]; bb2 += w * yy[o + k]; ab += w * xy[o + k];
      }
      s_h[0][ly][lx + o] = a; s_h[1][ly][lx + o] = bb; s_h[2][ly][lx + o] = aa; s_h[3][ly][lx + o] = bb2;
      s_h[4][ly][lx + o] = ab;
    }
  }
  __syncthreads();
  // vertical pass: thread = (column, group of 4 rows)
  const int tx = threadIdx.x & 31, tg = threadIdx.x >> 5;
  const int gx = x0 + tx;
  float acc[4][5];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[o][q] = 0.f;
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    float col[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) col[k] = s_h[q][tg * 4 + k][tx];
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int k = 0; k < 11; ++k) acc[o][q] += g.w[k] * col[o + k];
  }
  float local = 0.f;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int gy = y0 + tg * 4 + o;
    const float mu1 = acc[o][0], mu2 = acc[o][1], m11 = acc[o][2], m22 = acc[o][3], m12 = acc[o][4];
    const bool inside = gx < W && gy < H;
    const bool counted = inside && gx >= crop && gx < W - crop && gy >= crop && gy < H - crop;
    const float s1 = m11 - mu1 * mu1, s2 = m22 - mu2 * mu2, s12 = m12 - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2;
    const float B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s1 + s2 + C2;
    const float iB = 1.f / (B1 * B2);
    if (counted) local += A1 * A2 * iB;
    if (TRAIN && inside) {
      // derivatives of the map wrt mu1 and the raw moments m11 = G*x^2, m12 = G*xy; pre-scaled by 1/count and zero
      // outside the averaged region, so the backward pass only has to convolve them
      const float sc = counted ? inv_count : 0.f;
      const float d_mu1 = 2.f * mu2 * (A2 - A1) * iB - 2.f * mu1 * A1 * A2 * (B2 - B1) * iB * iB;
      const float d_m11 = -A1 * A2 * iB / B2;
      const float d_m12 = 2.f * A1 * iB;
      const int64_t oidx = ((int64_t)plane * H + gy) * W + gx;
      dm_dmu1[oidx] = sc * d_mu1; dm_dm11[oidx] = sc * d_m11; dm_dm12[oidx] = sc * d_m12;
    }
  }
  const float total = block_sum_256(local, s_red);
  if (threadIdx.x == 0)
    block_sums[(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = total;
}

// one block adds the per-block partials in index order (fixed association) and scales by 1/count
__global__ __launch_bounds__(256) void ssim_finish_kernel(const float* __restrict__ block_sums, int n, float inv_count,
                                                          float* __restrict__ out) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) acc += block_sums[i];
  const float total = block_sum_256(acc, s_red);
  if (threadIdx.x == 0) out[0] = total * inv_count;
}

// dL/dimg1[p] = gscale * ( (G * d_mu1)[p] + 2 x[p] (G * d_m11)[p] + y[p] (G * d_m12)[p] )
__global__ __launch_bounds__(256) void ssim_bwd_kernel(const float* __restrict__ img1, const float* __restrict__ img2,
                                                       Strides st1, Strides st2, Strides sto, int C, int H, int W,
                                                       Gauss11 g, const float* __restrict__ dm_dmu1,
                                                       const float* __restrict__ dm_dm11,
                                                       const float* __restrict__ dm_dm12,
                                                       const float* __restrict__ gscale_dev, float* __restrict__ dimg1) {
  __shared__ float s_in[3][IN][IN + 1];
  __shared__ float s_h[3][IN][TS + 1];
  const int plane = blockIdx.z, b = plane / C, c = plane % C;
  const int x0 = blockIdx.x * TS, y0 = blockIdx.y * TS;
  const int64_t pbase = (int64_t)plane * H * W;
  for (int i = threadIdx.x; i < IN * IN; i += 256) {
    const int ly = i / IN, lx = i % IN;
    const int gy = y0 + ly - HALO, gx = x0 + lx - HALO;
    const bool in = gy >= 0 && gy < H && gx >= 0 && gx < W;
    const int64_t o = pbase + (int64_t)gy * W + gx;
    s_in[0][ly][lx] = in ? dm_dmu1[o] : 0.f;
    s_in[1][ly][lx] = in ? dm_dm11[o] : 0.f;
    s_in[2][ly][lx] = in ? dm_dm12[o] : 0.f;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < IN * (TS / 4); i += 256) {
    const int ly = i / (TS / 4), lx = (i % (TS / 4)) * 4;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      float v[14];
#pragma unroll
      for (int k = 0; k < 14; ++k) v[k] = s_in[q][ly][lx + k];
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float a = 0.f;
#pragma unroll
        for (int k = 0; k < 11; ++k) a += g.w[k] * v[o + k];
        s_h[q][ly][lx + o] = a;
      }
    }
  }
  __syncthreads();
  const int tx = threadIdx.x & 31, tg = threadIdx.x >> 5;
  const int gx = x0 + tx;
  float acc[4][3];
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    float col[14];
#pragma unroll
    for (int k = 0; k < 14; ++k) col[k] = s_h[q][tg * 4 + k][tx];
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) a += g.w[k] * col[o + k];
      acc[o][q] = a;
    }
  }
  if (gx >= W) return;
  const float gs = gscale_dev[0];
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int gy = y0 + tg * 4 + o;
    if (gy >= H) continue;
    const float x = img1[b * st1.sB + c * st1.sC + gy * st1.sH + gx * st1.sW];
    const float y = img2[b * st2.sB + c * st2.sC + gy * st2.sH + gx * st2.sW];
    dimg1[b * sto.sB + c * sto.sC + gy * sto.sH + gx * sto.sW] = gs * (acc[o][0] + 2.f * x * acc[o][1] + y * acc[o][2]);
  }
}

Gauss11 make_gauss() {
  Gauss11 g;
  double sum = 0.0, w[11];
  for (int i = 0; i < 11; ++i) { const double d = i - 5; w[i] = exp(-(d * d) / (2.0 * 1.5 * 1.5)); sum += w[i]; }
  for (int i = 0; i < 11; ++i) g.w[i] = (float)(w[i] / sum);
  return g;
}

// ---- pixel losses of the same stage (splat_trainer/trainer/trainer.py:465-488: l1 / mse of the rendered image against
// the target, the image clamped to [0, 1] by the scene's post-activation, scene/color_model.py:154-160).  As torch ops
// that is clamp + sub + square + mean forward and four more elementwise passes backward over a 25 MB image; here one
// pass computes mean(f(clamp(x, lo, hi) - t)), f = square or abs, with a fixed-order two-level sum, and one pass writes
// the gradient (zero where the clamp is active, as torch.clamp's backward does).
constexpr int PL_THREADS = 256;
constexpr int PL_ITEMS = 8;      // floats per thread and round: two float4 loads

template <int KIND>   // 0: squared error, 1: absolute error
__global__ __launch_bounds__(PL_THREADS) void pixel_loss_fwd_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ t, int64_t n, float lo,
                                                                    float hi, float* __restrict__ block_sums) {
  __shared__ float s_w[PL_THREADS / 64];
  float acc = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * PL_THREADS + threadIdx.x) * 4; i < n; i += (int64_t)gridDim.x * PL_THREADS * 4) {
    float xv[4], tv[4];
    if (i + 3 < n) {
      const float4 a = *reinterpret_cast<const float4*>(x + i), b = *reinterpret_cast<const float4*>(t + i);
      xv[0] = a.x; xv[1] = a.y; xv[2] = a.z; xv[3] = a.w; tv[0] = b.x; tv[1] = b.y; tv[2] = b.z; tv[3] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { xv[k] = i + k < n ? x[i + k] : 0.f; tv[k] = i + k < n ? t[i + k] : fminf(fmaxf(0.f, lo), hi); }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float d = fminf(fmaxf(xv[k], lo), hi) - tv[k];
      acc += KIND == 0 ? d * d : fabsf(d);
    }
  }
  acc = gsr_wave_sum(acc);
  if (gsr_lane() == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(256) void pixel_loss_finish_kernel(const float* __restrict__ block_sums, int blocks,
                                                                float inv_n, float* __restrict__ out) {
  __shared__ float s_w[4];
  float acc = 0.f;
  for (int b = threadIdx.x; b < blocks; b += 256) acc += block_sums[b];
  acc = gsr_wave_sum(acc);
  if (gsr_lane() == 0) s_w[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = ((s_w[0] + s_w[1]) + (s_w[2] + s_w[3])) * inv_n;
}

template <int KIND>
__global__ __launch_bounds__(PL_THREADS) void pixel_loss_bwd_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ t, int64_t n, float lo,
                                                                    float hi, float inv_n,
                                                                    const float* __restrict__ grad_scale,
                                                                    float* __restrict__ dx) {
  const float gs = grad_scale[0] * inv_n;
  for (int64_t i = (int64_t)blockIdx.x * PL_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * PL_THREADS) {
    const float xv = x[i];
    const float d = fminf(fmaxf(xv, lo), hi) - t[i];
    const bool pass = xv >= lo && xv <= hi;                       // torch.clamp backward: inclusive on both ends
    const float g = KIND == 0 ? 2.f * d : (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f));
    dx[i] = pass ? gs * g : 0.f;
  }
}

}  // namespace

extern "C" {

size_t gsr_ssim_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 256;
  const size_t blocks = (size_t)((W + TS - 1) / TS) * ((H + TS - 1) / TS) * B * C;
  return blocks * sizeof(float) + 256;
}

// strides1/strides2/strides_out: 4 x int64 element strides (batch, channel, row, column).
// crop = 0 ("same": mean over the whole map) or 5 ("valid").  mean_out: device float.  The three derivative maps
// ([B,C,H,W] contiguous, may be NULL) are what gsr_ssim_backward consumes.
int gsr_ssim_forward(const float* img1, const float* img2, const int64_t* strides1_host, const int64_t* strides2_host,
                     int32_t B, int32_t C, int32_t H, int32_t W, int32_t crop, float* mean_out, float* dm_dmu1,
                     float* dm_dm11, float* dm_dm12, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || crop < 0 || !img1 || !img2 || !mean_out || !strides1_host || !strides2_host)
    return GSR_ERR_INVALID_ARGUMENT;
  if (W - 2 * crop <= 0 || H - 2 * crop <= 0) return GSR_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < gsr_ssim_workspace_bytes(B, C, H, W)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  const bool train = dm_dmu1 && dm_dm11 && dm_dm12;
  const Strides s1 = {strides1_host[0], strides1_host[1], strides1_host[2], strides1_host[3]};
  const Strides s2 = {strides2_host[0], strides2_host[1], strides2_host[2], strides2_host[3]};
  const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, B * C);
  const int nblocks = grid.x * grid.y * grid.z;
  const float inv_count = 1.f / ((float)B * C * (H - 2 * crop) * (float)(W - 2 * crop));
  float* block_sums = reinterpret_cast<float*>(workspace);
  const Gauss11 g = make_gauss();
  if (train)
    ssim_fwd_kernel<true><<<grid, 256, 0, stream>>>(img1, img2, s1, s2, C, H, W, crop, 0.01f * 0.01f, 0.03f * 0.03f,
                                                   inv_count, g, block_sums, dm_dmu1, dm_dm11, dm_dm12);
  else
    ssim_fwd_kernel<false><<<grid, 256, 0, stream>>>(img1, img2, s1, s2, C, H, W, crop, 0.01f * 0.01f, 0.03f * 0.03f,
                                                    inv_count, g, block_sums, nullptr, nullptr, nullptr);
  GSR_CHECK_LAUNCH();
  ssim_finish_kernel<<<1, 256, 0, stream>>>(block_sums, nblocks, inv_count, mean_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

// d_img1 = grad_scale * d mean_ssim / d img1;  grad_scale_dev: device float (the upstream gradient of the mean).
int gsr_ssim_backward(const float* img1, const float* img2, const int64_t* strides1_host, const int64_t* strides2_host,
                      const int64_t* strides_out_host, int32_t B, int32_t C, int32_t H, int32_t W, const float* dm_dmu1,
                      const float* dm_dm11, const float* dm_dm12, const float* grad_scale_dev, float* d_img1,
                      void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || !img1 || !img2 || !dm_dmu1 || !dm_dm11 || !dm_dm12 || !grad_scale_dev ||
      !d_img1 || !strides1_host || !strides2_host || !strides_out_host)
    return GSR_ERR_INVALID_ARGUMENT;
  const Strides s1 = {strides1_host[0], strides1_host[1], strides1_host[2], strides1_host[3]};
  const Strides s2 = {strides2_host[0], strides2_host[1], strides2_host[2], strides2_host[3]};
  const Strides so = {strides_out_host[0], strides_out_host[1], strides_out_host[2], strides_out_host[3]};
  const dim3 grid((W + TS - 1) / TS, (H + TS - 1) / TS, B * C);
  ssim_bwd_kernel<<<grid, 256, 0, stream>>>(img1, img2, s1, s2, so, C, H, W, make_gauss(), dm_dmu1, dm_dm11, dm_dm12,
                                           grad_scale_dev, d_img1);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

size_t gsr_pixel_loss_workspace_bytes(int64_t n) {
  const int64_t blocks = (n + (int64_t)PL_THREADS * 4 - 1) / ((int64_t)PL_THREADS * 4);
  return (size_t)(blocks < 1024 ? (blocks < 1 ? 1 : blocks) : 1024) * sizeof(float) + 256;
}

// loss_out[0] = mean(f(clamp(image, lo, hi) - target)), f = square (kind 0) or abs (kind 1); image / target contiguous.
int gsr_pixel_loss_forward(const float* image, const float* target, int64_t n, int32_t kind, float lo, float hi,
                           float* loss_out, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n <= 0 || !image || !target || !loss_out || (kind != 0 && kind != 1) || !(lo <= hi)) return GSR_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < gsr_pixel_loss_workspace_bytes(n)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  if ((reinterpret_cast<uintptr_t>(image) | reinterpret_cast<uintptr_t>(target)) & 15) return GSR_ERR_INVALID_ARGUMENT;
  const int64_t want = (n + (int64_t)PL_THREADS * 4 - 1) / ((int64_t)PL_THREADS * 4);
  const int blocks = (int)(want < 1024 ? want : 1024);
  float* sums = reinterpret_cast<float*>(workspace);
  if (kind == 0) pixel_loss_fwd_kernel<0><<<blocks, PL_THREADS, 0, stream>>>(image, target, n, lo, hi, sums);
  else pixel_loss_fwd_kernel<1><<<blocks, PL_THREADS, 0, stream>>>(image, target, n, lo, hi, sums);
  pixel_loss_finish_kernel<<<1, 256, 0, stream>>>(sums, blocks, 1.f / (float)n, loss_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

// d_image = grad_scale_dev[0] * d loss / d image (zero where the clamp is active).
int gsr_pixel_loss_backward(const float* image, const float* target, int64_t n, int32_t kind, float lo, float hi,
                            const float* grad_scale_dev, float* d_image, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (n <= 0 || !image || !target || !grad_scale_dev || !d_image || (kind != 0 && kind != 1)) return GSR_ERR_INVALID_ARGUMENT;
  const int64_t want = (n + PL_THREADS - 1) / PL_THREADS;
  const int blocks = (int)(want < 8192 ? want : 8192);
  if (kind == 0) pixel_loss_bwd_kernel<0><<<blocks, PL_THREADS, 0, stream>>>(image, target, n, lo, hi, 1.f / (float)n, grad_scale_dev, d_image);
  else pixel_loss_bwd_kernel<1><<<blocks, PL_THREADS, 0, stream>>>(image, target, n, lo, hi, 1.f / (float)n, grad_scale_dev, d_image);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
