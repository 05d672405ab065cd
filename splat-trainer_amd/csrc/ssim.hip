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
                                                       float* __restrict__ dm_dm11, float* __restrict__ dm_dm12,
                                                       float lo = -3.0e38f, float hi = 3.0e38f) {
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
    s_x[ly][lx] = in ? fminf(fmaxf(p1[gy * st1.sH + gx * st1.sW], lo), hi) : 0.f;     // (img1 clamped at load: fused loss)
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
                                                       const float* __restrict__ gscale_dev, float* __restrict__ dimg1,
                                                       float lo = -3.0e38f, float hi = 3.0e38f) {
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
  const float gs = gscale_dev ? gscale_dev[0] : 1.f;
#pragma unroll
  for (int o = 0; o < 4; ++o) {
    const int gy = y0 + tg * 4 + o;
    if (gy >= H) continue;
    const float x = fminf(fmaxf(img1[b * st1.sB + c * st1.sC + gy * st1.sH + gx * st1.sW], lo), hi);
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

// ---- the reference's loss mix in one native call per direction (splat_trainer/trainer/trainer.py:448-488, without
// reg_loss):   loss = w_l1 mean|x - t| + w_mse mean (x - t)^2 + w_ssim / L * sum_l (1 - ssim(pool^l x, pool^l t)),
// x = clamp(image, lo, hi) (the scene's post-activation, mlp_scene.py:421-423), pool = F.avg_pool2d(kernel 2, stride 2).
// As torch ops around fused_ssim that is ~60 launches per camera (clamp, 6 poolings and their 3 backward passes, two
// pixel losses, a dozen scalar combinations and four strided copies): +0.55 ms on c2's 0.91 ms step.  Here:
//   forward   msloss_pyramid (one pass over image and target: the pooled levels of both + the L1 / MSE partial sums)
//             -> ssim_fwd per level (level 0 clamps at load) -> msloss_finish (all means, the loss, the logged metrics)
//   backward  ssim_bwd per level (level 0 straight into d_image) -> msloss_combine (one pass: pixel terms, the levels'
//             gradients through the poolings, the clamp's mask, the upstream scale).
constexpr int ML_MAX_LEVELS = 4;

struct MsLevels {
  int n;                               // number of levels (1 .. 4)
  int H[ML_MAX_LEVELS], W[ML_MAX_LEVELS];
  float* pred[ML_MAX_LEVELS];          // [H_l, W_l, C] pooled clamped prediction (level 0: NULL, read from the image)
  float* targ[ML_MAX_LEVELS];
  float* dlev[ML_MAX_LEVELS];          // [H_l, W_l, C] d mean_ssim_l / d pred_l (level 0: the output image itself)
};

// thread = (8 x 8 block of level-0 pixels, channel): 2^(levels-1) <= 8.  Levels beyond `n` are not written.
__global__ __launch_bounds__(256) void msloss_pyramid_kernel(const float* __restrict__ image,
                                                             const float* __restrict__ target, int C, MsLevels lv,
                                                             float lo, float hi, float* __restrict__ block_l1,
                                                             float* __restrict__ block_mse) {
  __shared__ float s_a[4], s_b[4];
  const int H = lv.H[0], W = lv.W[0];
  const int bw = (W + 7) / 8, bh = (H + 7) / 8;
  const int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float l1 = 0.f, mse = 0.f;
  if (id < (int64_t)bw * bh * C) {
    const int c = (int)(id % C);
    const int bx = (int)((id / C) % bw), by = (int)(id / ((int64_t)C * bw));
    float x1[4][4], t1[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sx = 0.f, st = 0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int y = by * 8 + 2 * j + dy, x = bx * 8 + 2 * i + dx;
            if (y < H && x < W) {
              const int64_t o = ((int64_t)y * W + x) * C + c;
              const float xv = fminf(fmaxf(image[o], lo), hi), tv = target[o];
              const float d = xv - tv;
              l1 += fabsf(d);
              mse += d * d;
              sx += xv; st += tv;
            }
          }
        x1[j][i] = 0.25f * sx; t1[j][i] = 0.25f * st;
        const int y1 = by * 4 + j, xq = bx * 4 + i;
        if (lv.n > 1 && y1 < lv.H[1] && xq < lv.W[1]) {
          const int64_t o = ((int64_t)y1 * lv.W[1] + xq) * C + c;
          lv.pred[1][o] = x1[j][i]; lv.targ[1][o] = t1[j][i];
        }
      }
    if (lv.n > 2) {
      float x2[2][2], t2[2][2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          x2[j][i] = 0.25f * ((x1[2 * j][2 * i] + x1[2 * j][2 * i + 1]) + (x1[2 * j + 1][2 * i] + x1[2 * j + 1][2 * i + 1]));
          t2[j][i] = 0.25f * ((t1[2 * j][2 * i] + t1[2 * j][2 * i + 1]) + (t1[2 * j + 1][2 * i] + t1[2 * j + 1][2 * i + 1]));
          const int y2 = by * 2 + j, xq = bx * 2 + i;
          if (y2 < lv.H[2] && xq < lv.W[2]) {
            const int64_t o = ((int64_t)y2 * lv.W[2] + xq) * C + c;
            lv.pred[2][o] = x2[j][i]; lv.targ[2][o] = t2[j][i];
          }
        }
      if (lv.n > 3 && by < lv.H[3] && bx < lv.W[3]) {
        const int64_t o = ((int64_t)by * lv.W[3] + bx) * C + c;
        lv.pred[3][o] = 0.25f * ((x2[0][0] + x2[0][1]) + (x2[1][0] + x2[1][1]));
        lv.targ[3][o] = 0.25f * ((t2[0][0] + t2[0][1]) + (t2[1][0] + t2[1][1]));
      }
    }
  }
  l1 = gsr_wave_sum(l1);
  mse = gsr_wave_sum(mse);
  if (gsr_lane() == 0) { s_a[threadIdx.x >> 6] = l1; s_b[threadIdx.x >> 6] = mse; }
  __syncthreads();
  if (threadIdx.x == 0) {
    block_l1[blockIdx.x] = (s_a[0] + s_a[1]) + (s_a[2] + s_a[3]);
    block_mse[blockIdx.x] = (s_b[0] + s_b[1]) + (s_b[2] + s_b[3]);
  }
}

struct MsSums {
  const float* ssim[ML_MAX_LEVELS];   // per-level block partials of the SSIM map
  int ssim_n[ML_MAX_LEVELS];
  float ssim_inv[ML_MAX_LEVELS];
  const float* l1;
  const float* mse;
  int pix_n;
  float pix_inv;
};

// metrics_out: [loss, l1, mse, ssim_0 .. ssim_{L-1}]  (what the reference logs with .item(), trainer.py:466-481)
__global__ __launch_bounds__(256) void msloss_finish_kernel(MsSums sm, int levels, float w_l1, float w_mse, float w_ssim,
                                                            float* __restrict__ metrics_out) {
  __shared__ float s_red[4];
  float vals[2 + ML_MAX_LEVELS];
  float a = 0.f;
  for (int i = threadIdx.x; i < sm.pix_n; i += 256) a += sm.l1[i];
  vals[0] = block_sum_256(a, s_red) * sm.pix_inv;
  __syncthreads();
  a = 0.f;
  for (int i = threadIdx.x; i < sm.pix_n; i += 256) a += sm.mse[i];
  vals[1] = block_sum_256(a, s_red) * sm.pix_inv;
  __syncthreads();
  float ssim_loss = 0.f;
  for (int l = 0; l < levels; ++l) {
    a = 0.f;
    for (int i = threadIdx.x; i < sm.ssim_n[l]; i += 256) a += sm.ssim[l][i];
    vals[2 + l] = block_sum_256(a, s_red) * sm.ssim_inv[l];
    __syncthreads();
    ssim_loss += 1.f - vals[2 + l];
  }
  if (threadIdx.x == 0) {
    metrics_out[0] = vals[0] * w_l1 + vals[1] * w_mse + (ssim_loss / (float)levels) * w_ssim;
    for (int k = 0; k < 2 + levels; ++k) metrics_out[1 + k] = vals[k];
  }
}

// d_image holds d mean_ssim_0 / d x on entry (ssim_bwd of level 0 wrote it) and the loss gradient on exit.
__global__ __launch_bounds__(256) void msloss_combine_kernel(const float* __restrict__ image,
                                                             const float* __restrict__ target, int C, MsLevels lv,
                                                             float lo, float hi, float w_l1, float w_mse, float w_ssim,
                                                             const float* __restrict__ up, float* __restrict__ d_image) {
  const int H = lv.H[0], W = lv.W[0];
  const int64_t n = (int64_t)H * W * C;
  const float inv_n = 1.f / (float)n, gs = up[0], cs = -w_ssim / (float)lv.n;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float xv = image[i];
    if (!(xv >= lo && xv <= hi)) { d_image[i] = 0.f; continue; }     // torch.clamp backward: inclusive on both ends
    const int c = (int)(i % C);
    const int64_t p = i / C;
    const int x = (int)(p % W), y = (int)(p / W);
    const float d = xv - target[i];
    float g = w_l1 * inv_n * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) + w_mse * inv_n * 2.f * d;
    float s = d_image[i], sc = 0.25f;
#pragma unroll
    for (int l = 1; l < ML_MAX_LEVELS; ++l) {
      if (l < lv.n) {
        const int yl = y >> l, xl = x >> l;
        if (yl < lv.H[l] && xl < lv.W[l]) s += sc * lv.dlev[l][((int64_t)yl * lv.W[l] + xl) * C + c];
        sc *= 0.25f;
      }
    }
    d_image[i] = gs * (g + cs * s);
  }
}

struct MsPlan {                 // byte offsets into the workspace
  MsLevels lv;
  size_t maps[ML_MAX_LEVELS];   // three [C, H_l, W_l] derivative maps per level, back to back
  size_t sums[ML_MAX_LEVELS];   // SSIM block partials per level
  int sums_n[ML_MAX_LEVELS];
  size_t pix_l1, pix_mse;
  int pix_blocks;
  size_t total;
};

MsPlan ms_plan(int H, int W, int C, int levels, uint8_t* base) {
  MsPlan p;
  size_t at = 0;
  auto take = [&](size_t bytes) { const size_t o = at; at = (at + bytes + 255) / 256 * 256; return o; };
  p.lv.n = levels;
  int h = H, w = W;
  for (int l = 0; l < ML_MAX_LEVELS; ++l) {
    p.lv.H[l] = p.lv.W[l] = 0;
    p.lv.pred[l] = p.lv.targ[l] = p.lv.dlev[l] = nullptr;
    p.maps[l] = p.sums[l] = 0; p.sums_n[l] = 0;
    if (l >= levels) continue;
    if (l > 0) { h /= 2; w /= 2; }
    p.lv.H[l] = h; p.lv.W[l] = w;
    const size_t plane = (size_t)h * w * C * sizeof(float);
    if (l > 0) {
      p.lv.pred[l] = reinterpret_cast<float*>(base + take(plane));
      p.lv.targ[l] = reinterpret_cast<float*>(base + take(plane));
      p.lv.dlev[l] = reinterpret_cast<float*>(base + take(plane));
    }
    p.maps[l] = take(3 * plane);
    p.sums_n[l] = ((w + TS - 1) / TS) * ((h + TS - 1) / TS) * C;
    p.sums[l] = take((size_t)p.sums_n[l] * sizeof(float));
  }
  const int64_t work = (int64_t)((W + 7) / 8) * ((H + 7) / 8) * C;
  p.pix_blocks = (int)((work + 255) / 256);
  p.pix_l1 = take((size_t)p.pix_blocks * sizeof(float));
  p.pix_mse = take((size_t)p.pix_blocks * sizeof(float));
  p.total = at;
  return p;
}

inline bool ms_args_ok(int H, int W, int C, int levels) {
  if (H <= 0 || W <= 0 || C < 1 || C > 4 || levels < 1 || levels > ML_MAX_LEVELS) return false;
  return (H >> (levels - 1)) > 10 && (W >> (levels - 1)) > 10;      // padding = "valid" needs more than 10 pixels per side
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

size_t gsr_msloss_workspace_bytes(int32_t H, int32_t W, int32_t C, int32_t levels) {
  if (!ms_args_ok(H, W, C, levels)) return 256;
  return ms_plan(H, W, C, levels, nullptr).total + 256;
}

// metrics_out: device floats [loss, l1, mse, ssim_0 .. ssim_{levels-1}].  The workspace keeps what the backward pass reads.
int gsr_msloss_forward(const float* image, const float* target, int32_t H, int32_t W, int32_t C, int32_t levels,
                       float w_l1, float w_mse, float w_ssim, float lo, float hi, float* metrics_out, void* workspace,
                       size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!ms_args_ok(H, W, C, levels) || !image || !target || !metrics_out || !(lo <= hi)) return GSR_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < gsr_msloss_workspace_bytes(H, W, C, levels)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  uint8_t* base = reinterpret_cast<uint8_t*>(workspace);
  const MsPlan p = ms_plan(H, W, C, levels, base);
  msloss_pyramid_kernel<<<p.pix_blocks, 256, 0, stream>>>(image, target, C, p.lv, lo, hi,
                                                         reinterpret_cast<float*>(base + p.pix_l1),
                                                         reinterpret_cast<float*>(base + p.pix_mse));
  GSR_CHECK_LAUNCH();
  const Gauss11 g = make_gauss();
  MsSums sm;
  for (int l = 0; l < ML_MAX_LEVELS; ++l) { sm.ssim[l] = nullptr; sm.ssim_n[l] = 0; sm.ssim_inv[l] = 0.f; }
  for (int l = 0; l < levels; ++l) {
    const int h = p.lv.H[l], w = p.lv.W[l];
    const Strides st = {0, 1, (int64_t)w * C, C};                 // (H, W, C) image seen as one batch of C planes
    const float* x = l ? p.lv.pred[l] : image;
    const float* t = l ? p.lv.targ[l] : target;
    const dim3 grid((w + TS - 1) / TS, (h + TS - 1) / TS, C);
    const float inv_count = 1.f / ((float)C * (h - 10) * (float)(w - 10));
    float* maps = reinterpret_cast<float*>(base + p.maps[l]);
    const size_t plane = (size_t)h * w * C;
    ssim_fwd_kernel<true><<<grid, 256, 0, stream>>>(x, t, st, st, C, h, w, 5, 0.01f * 0.01f, 0.03f * 0.03f, inv_count, g,
                                                   reinterpret_cast<float*>(base + p.sums[l]), maps, maps + plane,
                                                   maps + 2 * plane, l ? -3.0e38f : lo, l ? 3.0e38f : hi);
    GSR_CHECK_LAUNCH();
    sm.ssim[l] = reinterpret_cast<const float*>(base + p.sums[l]);
    sm.ssim_n[l] = p.sums_n[l];
    sm.ssim_inv[l] = inv_count;
  }
  sm.l1 = reinterpret_cast<const float*>(base + p.pix_l1);
  sm.mse = reinterpret_cast<const float*>(base + p.pix_mse);
  sm.pix_n = p.pix_blocks;
  sm.pix_inv = 1.f / ((float)H * (float)W * (float)C);
  msloss_finish_kernel<<<1, 256, 0, stream>>>(sm, levels, w_l1, w_mse, w_ssim, metrics_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

// d_image [H, W, C] = grad_scale_dev[0] * d loss / d image, from the workspace gsr_msloss_forward left.
int gsr_msloss_backward(const float* image, const float* target, int32_t H, int32_t W, int32_t C, int32_t levels,
                        float w_l1, float w_mse, float w_ssim, float lo, float hi, const float* grad_scale_dev,
                        void* workspace, size_t workspace_bytes, float* d_image, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (!ms_args_ok(H, W, C, levels) || !image || !target || !grad_scale_dev || !d_image) return GSR_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < gsr_msloss_workspace_bytes(H, W, C, levels)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  uint8_t* base = reinterpret_cast<uint8_t*>(workspace);
  const MsPlan p = ms_plan(H, W, C, levels, base);
  const Gauss11 g = make_gauss();
  for (int l = 0; l < levels; ++l) {
    const int h = p.lv.H[l], w = p.lv.W[l];
    const Strides st = {0, 1, (int64_t)w * C, C};
    const float* x = l ? p.lv.pred[l] : image;
    const float* t = l ? p.lv.targ[l] : target;
    const float* maps = reinterpret_cast<const float*>(base + p.maps[l]);
    const size_t plane = (size_t)h * w * C;
    const dim3 grid((w + TS - 1) / TS, (h + TS - 1) / TS, C);
    ssim_bwd_kernel<<<grid, 256, 0, stream>>>(x, t, st, st, st, C, h, w, g, maps, maps + plane, maps + 2 * plane, nullptr,
                                             l ? p.lv.dlev[l] : d_image, l ? -3.0e38f : lo, l ? 3.0e38f : hi);
    GSR_CHECK_LAUNCH();
  }
  const int64_t n = (int64_t)H * W * C;
  const int64_t want = (n + 255) / 256;
  msloss_combine_kernel<<<(int)(want < 16384 ? want : 16384), 256, 0, stream>>>(image, target, C, p.lv, lo, hi, w_l1, w_mse,
                                                                              w_ssim, grad_scale_dev, d_image);
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
