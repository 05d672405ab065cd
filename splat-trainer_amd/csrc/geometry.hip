// K1 frustum cull + wave-ballot compaction, K2 3D->2D projection fwd/bwd, K3 SH colour fwd/bwd.
// All HBM-streaming kernels: one thread per point, coalesced row reads, no LDS.
#include "gsr_device.h"
#include "../../include/gsplat_hip.h"

static_assert(sizeof(GsrRasterParams) == sizeof(GsrRasterParamsC), "raster params layout");

namespace {

constexpr int CULL_ITEMS = 16;                     // points per lane
constexpr int CULL_WAVE_SPAN = 64 * CULL_ITEMS;    // 1024 consecutive points per wave
constexpr int64_t CULL_SELF_PREFIX_MAX_WAVES = 4096;    // up to 4.2 M points the write pass sums the wave counts itself

// pass 1: each wave counts the in-view points of its 1024-point span (ballot + popcount)
__global__ __launch_bounds__(256) void cull_count_kernel(const float* __restrict__ pos, int64_t N,
                                                         const float* __restrict__ Tcw, const float* __restrict__ proj,
                                                         int W, int H, float near_p, float far_p, float margin,
                                                         uint32_t* __restrict__ wave_counts, int64_t num_waves) {
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= num_waves) return;
  const int lane = gsr_lane();
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  const int64_t base = wave * CULL_WAVE_SPAN;
  uint32_t cnt = 0;
#pragma unroll 4
  for (int it = 0; it < CULL_ITEMS; ++it) {
    int64_t i = base + it * 64 + lane;
    bool in = false;
    if (i < N) in = gsr_in_view(cam, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], W, H, near_p, far_p, margin);
    cnt += (uint32_t)__popcll(__ballot(in));
  }
  if (lane == 0) wave_counts[wave] = cnt;
}

// pass 2: recompute the flags, rank by mbcnt below the lane, write indices in ascending point order
__global__ __launch_bounds__(256) void cull_write_kernel(const float* __restrict__ pos, int64_t N,
                                                         const float* __restrict__ Tcw, const float* __restrict__ proj,
                                                         int W, int H, float near_p, float far_p, float margin,
                                                         const uint32_t* __restrict__ wave_offsets, int64_t num_waves,
                                                         int64_t* __restrict__ indexes,
                                                         uint32_t* __restrict__ count_dev) {
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (wave >= num_waves) return;
  const int lane = gsr_lane();
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  const int64_t base = wave * CULL_WAVE_SPAN;
  uint32_t run;
  if (count_dev) {
    // wave_offsets holds the waves' RAW counts: every wave adds up its predecessors' itself (a few thousand words at
    // millions of points -- cheaper than a scan launch in between) and the last one leaves the total
    uint32_t before = 0u;
    for (int64_t j = lane; j < wave; j += 64) before += wave_offsets[j];
    run = gsr_wave_sum_u32(before);
    if (wave == num_waves - 1 && lane == 0) *count_dev = run + wave_offsets[wave];
  } else {
    run = wave_offsets[wave];
  }
#pragma unroll 4
  for (int it = 0; it < CULL_ITEMS; ++it) {
    int64_t i = base + it * 64 + lane;
    bool in = false;
    if (i < N) in = gsr_in_view(cam, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], W, H, near_p, far_p, margin);
    uint64_t m = __ballot(in);
    if (in) indexes[run + (uint32_t)gsr_mbcnt(m)] = i;
    run += (uint32_t)__popcll(m);
  }
}

__global__ __launch_bounds__(256) void project_fwd_kernel(const float* __restrict__ pos, const float* __restrict__ ls,
                                                          const float* __restrict__ rot, const float* __restrict__ logit,
                                                          const int64_t* __restrict__ idx, int64_t M,
                                                          const float* __restrict__ Tcw, const float* __restrict__ proj,
                                                          GsrRasterParams rp, float* __restrict__ g2d,
                                                          float* __restrict__ depth,
                                                          const uint32_t* __restrict__ count_dev,
                                                          uint32_t* __restrict__ depth_keys, uint32_t key_bias,
                                                          uint32_t key_max) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // M is an upper bound when count_dev is given: the true count is still on the device (K1 just produced it)
  if (m >= M || (count_dev != nullptr && m >= (int64_t)*count_dev)) return;
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  const int64_t i = idx[m];
  float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  float s[3] = {ls[3 * i], ls[3 * i + 1], ls[3 * i + 2]};
  const float4 qv = *reinterpret_cast<const float4*>(rot + 4 * i);
  float q[4] = {qv.x, qv.y, qv.z, qv.w};
  GsrProjected o = gsr_project_one(cam, rp, p, s, q, logit[i]);
  float* g = g2d + 6 * m;
  *reinterpret_cast<float2*>(g) = make_float2(o.u, o.v);
  *reinterpret_cast<float2*>(g + 2) = make_float2(o.A, o.B);
  *reinterpret_cast<float2*>(g + 4) = make_float2(o.C, o.opacity);
  depth[m] = o.depth;
  // the depth sort's keys (binning.hip: depth_keys_kernel) while the value is at hand
  if (depth_keys) depth_keys[m] = gsr_depth_key(o.depth, key_bias, key_max);
}

// The depth sort's keys straight from the positions (12 bytes per visible splat): lets the sort start right behind the cull,
// on a second stream, while the fused K2 + K3 kernel streams the coefficient rows on the first.  Same depth bits as the
// projection writes into the rows (gsr_to_camera: formed in double, rounded once).
__global__ __launch_bounds__(256) void depth_keys_pos_kernel(const float* __restrict__ pos, const int64_t* __restrict__ idx,
                                                             int64_t M, const uint32_t* __restrict__ count_dev,
                                                             const float* __restrict__ Tcw, const float* __restrict__ proj,
                                                             uint32_t* __restrict__ keys, uint32_t key_bias,
                                                             uint32_t key_max) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M || (count_dev != nullptr && m >= (int64_t)*count_dev)) return;
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  const int64_t i = idx[m];
  float x, y, z;
  gsr_to_camera(cam, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], x, y, z);
  keys[m] = gsr_depth_key(z, key_bias, key_max);
}

template <bool ACC>
__global__ __launch_bounds__(256) void project_bwd_kernel(const float* __restrict__ pos, const float* __restrict__ ls,
                                                          const float* __restrict__ rot, const float* __restrict__ logit,
                                                          const int64_t* __restrict__ idx, int64_t M,
                                                          const float* __restrict__ Tcw, const float* __restrict__ proj,
                                                          GsrRasterParams rp, const float* __restrict__ dg2d,
                                                          const float* __restrict__ ddepth, float* __restrict__ dpos,
                                                          float* __restrict__ dls, float* __restrict__ drot,
                                                          float* __restrict__ dlogit) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  const int64_t i = idx[m];
  float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  float s[3] = {ls[3 * i], ls[3 * i + 1], ls[3 * i + 2]};
  const float4 qv = *reinterpret_cast<const float4*>(rot + 4 * i);
  float q[4] = {qv.x, qv.y, qv.z, qv.w};
  float g[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) g[k] = dg2d[6 * m + k];
  const float gd = ddepth ? ddepth[m] : 0.f;
  GsrProjectGrad o = gsr_project_one_bwd(cam, rp, p, s, q, logit[i], g, gd);
  // rows of ``idx`` are unique, so the accumulate form (+=) is a race-free read-modify-write
  float4 r = make_float4(o.dq[0], o.dq[1], o.dq[2], o.dq[3]);
  if (ACC) {
    const float4 old = *reinterpret_cast<const float4*>(drot + 4 * i);
    r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dpos[3 * i + k] = ACC ? dpos[3 * i + k] + o.dp[k] : o.dp[k];
    dls[3 * i + k] = ACC ? dls[3 * i + k] + o.dls[k] : o.dls[k];
  }
  *reinterpret_cast<float4*>(drot + 4 * i) = r;
  dlogit[i] = ACC ? dlogit[i] + o.dlogit : o.dlogit;
}

// Colour of one splat from its coefficient row (and, JAC, d colour / d position through the view direction d = v/|v|).
// Shared by the stand-alone K3 kernel and the fused K2+K3 kernel: explicit fma chains wherever the colour is formed, so
// every instantiation (JAC or not, fused or not) rounds identically.
// Core: the 3K coefficients are in registers already (from global memory or from an LDS-staged copy).  ONE sweep over k
// feeds the three colours and (JAC) their derivatives wrt the view direction: dY_k is formed where it is consumed, so
// only the coefficients and the K basis values are live across the sweep.
template <int K, bool JAC>
__device__ __forceinline__ void gsr_sh_colour_w(const float (&w)[3][K], float dx, float dy, float dz, float col[3],
                                                float J[9]) {
  const float inv = 1.f / sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
  const float x = dx * inv, y = dy * inv, z = dz * inv;
  float Y[K];
  gsr_sh_basis<K>(x, y, z, Y);
  float acc[3] = {0.5f, 0.5f, 0.5f};
  float g[3][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
#pragma unroll
  for (int k = 0; k < K; ++k) {
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) acc[ch] = fmaf(w[ch][k], Y[k], acc[ch]);   // per channel: the same fma chain over k
    if (JAC && k > 0) {
      float ax, ay, az;
      gsr_sh_basis_grad_at<K>(k, x, y, z, ax, ay, az);
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        g[ch][0] += w[ch][k] * ax; g[ch][1] += w[ch][k] * ay; g[ch][2] += w[ch][k] * az;
      }
    }
  }
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    col[ch] = acc[ch];
    if (JAC) {
      // d colour_ch / d position; saved (36 B per splat) so that the backward pass does not have to stream the
      // 12K-byte coefficient row again
      const float dot = g[ch][0] * x + g[ch][1] * y + g[ch][2] * z;
      J[3 * ch] = (g[ch][0] - x * dot) * inv;
      J[3 * ch + 1] = (g[ch][1] - y * dot) * inv;
      J[3 * ch + 2] = (g[ch][2] - z * dot) * inv;
    }
  }
}

template <int K, bool JAC>
__device__ __forceinline__ void gsr_sh_colour(const float* __restrict__ row, float dx, float dy, float dz, float col[3],
                                              float J[9]) {
  float w[3][K];                     // all 3K coefficients first (3K/4 16-byte loads in flight)
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    if (K % 4 == 0) {
#pragma unroll
      for (int k = 0; k < K; k += 4) {
        const float4 v = *reinterpret_cast<const float4*>(row + ch * K + k);
        w[ch][k] = v.x; w[ch][k + 1] = v.y; w[ch][k + 2] = v.z; w[ch][k + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) w[ch][k] = row[ch * K + k];
    }
  }
  gsr_sh_colour_w<K, JAC>(w, dx, dy, dz, col, J);
}

template <int K, bool JAC>
__global__ __launch_bounds__(256) void sh_fwd_kernel(const float* __restrict__ sh, const float* __restrict__ pos,
                                                     const int64_t* __restrict__ idx, int64_t M,
                                                     const float* __restrict__ cam_pos, float* __restrict__ out,
                                                     float* __restrict__ jac, const uint32_t* __restrict__ count_dev) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  // M is an upper bound when count_dev is given (the visible count is still on the device, as in project_fwd_kernel)
  if (m >= M || (count_dev != nullptr && m >= (int64_t)*count_dev)) return;
  const int64_t i = idx[m];
  const float dx = pos[3 * i] - cam_pos[0], dy = pos[3 * i + 1] - cam_pos[1], dz = pos[3 * i + 2] - cam_pos[2];
  float col[3], J[9];
  gsr_sh_colour<K, JAC>(sh + (int64_t)3 * K * i, dx, dy, dz, col, J);
  out[3 * m] = col[0]; out[3 * m + 1] = col[1]; out[3 * m + 2] = col[2];
  if (JAC) {
#pragma unroll
    for (int e = 0; e < 9; ++e) jac[9 * m + e] = J[e];
  }
}

// q <= qlim  <=>  the pixel is inside the support (q <= q_max) AND alpha = opacity exp(-q/2) reaches the threshold:
// the composite kernels test ONE per-splat number per pixel instead of q and alpha separately.  One definition for every
// kernel that writes rows (same bits for the same opacity).  opacity below the threshold: negative (no pixel passes).
__device__ __forceinline__ float gsr_qlim(float opacity, const GsrRasterParams& rp) {
  return fminf(rp.q_max, 2.f * logf(opacity / rp.alpha_threshold));
}

// Pixel bounding box of a splat's support, for the rows' words 12 / 13 (x0 | x1 << 16, y0 | y1 << 16; signed 16-bit pixel
// coordinates, inclusive; x0 > x1: empty): every pixel whose centre passes the composite kernels' test q <= qlim lies
// inside -- the half-widths are gsr_splat_extent's (the same inflated support K4 bins with).  Read by the windowed
// backward walk (composite.hip: composite_bwd_win_kernel), which evaluates 64-pixel windows laid over box-and-tile.
__device__ __forceinline__ float2 gsr_pixel_bbox(float u, float v, float A, float B, float C, float opacity,
                                                 const GsrRasterParams& rp) {
  int x0 = 1, x1 = 0, y0 = 1, y1 = 0;
  const float det = A * C - B * B;
  if (opacity >= rp.alpha_threshold && det > 0.f) {
    const float qop = 2.f * logf(opacity / rp.alpha_threshold) * 1.0001f + 1e-4f;
    const float qmax = fminf(rp.q_max * 1.00001f, qop);
    const float hx = sqrtf(qmax * C / det) + 1e-3f, hy = sqrtf(qmax * A / det) + 1e-3f;
    const float lo = -32768.f, hi = 32767.f;
    x0 = (int)fminf(fmaxf(ceilf(u - hx - 0.5f), lo), hi);
    x1 = (int)fminf(fmaxf(floorf(u + hx - 0.5f), lo), hi);
    y0 = (int)fminf(fmaxf(ceilf(v - hy - 0.5f), lo), hi);
    y1 = (int)fminf(fmaxf(floorf(v + hy - 0.5f), lo), hi);
  }
  return make_float2(__uint_as_float(((uint32_t)x0 & 0xFFFFu) | ((uint32_t)x1 << 16)),
                     __uint_as_float(((uint32_t)y0 & 0xFFFFu) | ((uint32_t)y1 << 16)));
}

// sigma = sqrt(eig(cov)) of a projected splat from its conic: cov = conic^-1 = [C -B; -B A] / det(conic).
// One definition (rounding pinned) for every kernel that reports points.screen_scale.
__device__ __forceinline__ float2 gsr_screen_scale(float A, float B, float C) {
#pragma clang fp contract(off)
  const float idet = 1.f / (A * C - B * B);
  const float mid = 0.5f * (A + C) * idet;
  const float rad = sqrtf(fmaxf(mid * mid - idet, 0.f));
  return make_float2(sqrtf(mid + rad), sqrtf(fmaxf(mid - rad, 0.f)));
}

// K2 + K3 fused: ONE 64-byte row per visible splat, written whole (four 16-byte stores by one thread = one full
// line), in splat order:   u v A B | C opacity qlim f0 | f1 f2 depth log2(opacity) | 0 0 0 0.
// Everything downstream of the projection -- K4's gather through the depth order, the scalar record loads of K6 / K7 --
// reads this row and nothing else, so the depth-order permutation costs one line per splat instead of one per source
// array.  Also written: screen_scale (M,2), the depth sort's keys, and (JAC) d colour / d position for the backward pass.
// The coefficient rows (12K bytes each, 192 B at degree 3 -- 5/6 of what this kernel reads) are staged through LDS: the
// block's 256 rows are fetched with fully coalesced 16-byte loads (consecutive lanes walk along a row, so every wave
// instruction covers whole 64-byte lines instead of 64 different ones) and each thread then takes its own row out of
// LDS.  Odd row pitch (3K or 3K + 1 words): conflict-free 4-byte LDS accesses, and at degree 3 three blocks (48 waves'
// worth of rows would not fit with a 16-byte aligned pitch) share a CU's 160 KB.
#ifndef GSR_PSF_STAGE_OUT
#define GSR_PSF_STAGE_OUT 1
#endif
template <int K>
struct ShStage {
  static constexpr int ROW = 3 * K;
  static constexpr bool VEC = (ROW % 4) == 0;
  static constexpr int PITCH = ROW | 1;
};

template <int K, bool JAC>
__global__ __launch_bounds__(256) void project_sh_fwd_kernel(
    const float* __restrict__ pos, const float* __restrict__ ls, const float* __restrict__ rot,
    const float* __restrict__ logit, const float* __restrict__ sh, const int64_t* __restrict__ idx, int64_t M,
    const float* __restrict__ Tcw, const float* __restrict__ proj, const float* __restrict__ cam_pos,
    GsrRasterParams rp, float* __restrict__ rows, float* __restrict__ sscale, float* __restrict__ jac,
    const uint32_t* __restrict__ count_dev, uint32_t* __restrict__ depth_keys, uint32_t key_bias, uint32_t key_max) {
  constexpr int ROW = ShStage<K>::ROW, PITCH = ShStage<K>::PITCH;
  constexpr int OUT_FLOATS = GSR_PSF_STAGE_OUT ? 256 * (GSR_ROW_FLOATS + 1 + 9) : 0;     // rows + Jacobians on the way out
  constexpr int STAGE_FLOATS = 256 * PITCH > OUT_FLOATS ? 256 * PITCH : OUT_FLOATS;
  __shared__ float s_rows[STAGE_FLOATS];
  __shared__ int32_t s_idx[256];
  // M is an upper bound when count_dev is given: the true count is still on the device (K1 just produced it)
  const int64_t count = count_dev ? min(M, (int64_t)*count_dev) : M;
  const int64_t m0 = (int64_t)blockIdx.x * 256;
  if (m0 >= count) return;                                   // block-uniform
  const int tid = (int)threadIdx.x;
  const int64_t m = m0 + tid;
  const bool valid = m < count;
  const int64_t i = valid ? idx[m] : -1;
  s_idx[tid] = (int32_t)i;                                    // N <= 2^31 - 1 (gsr_frustum_cull)
  __syncthreads();
  if (ShStage<K>::VEC) {
    constexpr int PARTS = ROW / 4;                             // 16-byte pieces per row
#pragma unroll
    for (int it = 0; it < PARTS; ++it) {
      const int e = it * 256 + tid;
      const int row = e / PARTS, part = e - row * PARTS;
      const int64_t src = s_idx[row];
      if (src >= 0) {
        const float4 v = *reinterpret_cast<const float4*>(sh + src * ROW + 4 * part);
        float* dst = s_rows + row * PITCH + 4 * part;
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      }
    }
  } else {
#pragma unroll
    for (int it = 0; it < ROW; ++it) {
      const int e = it * 256 + tid;
      const int row = e / ROW, col = e - row * ROW;
      const int64_t src = s_idx[row];
      if (src >= 0) s_rows[row * PITCH + col] = sh[src * ROW + col];
    }
  }
  __syncthreads();
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  const int64_t ii = valid ? i : idx[m0];                     // threads past the count redo the block's first row (unused)
  float p[3] = {pos[3 * ii], pos[3 * ii + 1], pos[3 * ii + 2]};
  float s[3] = {ls[3 * ii], ls[3 * ii + 1], ls[3 * ii + 2]};
  const float4 qv = *reinterpret_cast<const float4*>(rot + 4 * ii);
  float q[4] = {qv.x, qv.y, qv.z, qv.w};
  const GsrProjected o = gsr_project_one(cam, rp, p, s, q, logit[ii]);
  float w[3][K];
  const float* mine = s_rows + tid * PITCH;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) w[ch][k] = mine[ch * K + k];
  float col[3], J[9];
  gsr_sh_colour_w<K, JAC>(w, p[0] - cam_pos[0], p[1] - cam_pos[1], p[2] - cam_pos[2], col, J);
#if GSR_PSF_STAGE_OUT
  // The rows and the Jacobians leave through LDS as well: the block's 256 x 64 B (and 256 x 36 B) are contiguous in
  // memory, so the staged copy goes out with fully coalesced 16-byte stores (whole lines per wave instruction) instead
  // of four (nine) stores per thread at a 64-byte (36-byte) stride.
  __syncthreads();                                            // every thread has taken its coefficients out of s_rows
  constexpr int OP = GSR_ROW_FLOATS + 1;                      // odd pitch: conflict-free
  float* s_out = s_rows;
  float* s_jac = s_rows + 256 * OP;
  if (valid) {
    float* d = s_out + tid * OP;
    d[0] = o.u; d[1] = o.v; d[2] = o.A; d[3] = o.B; d[4] = o.C; d[5] = o.opacity; d[6] = gsr_qlim(o.opacity, rp);
    const float2 bb = gsr_pixel_bbox(o.u, o.v, o.A, o.B, o.C, o.opacity, rp);
    d[7] = col[0]; d[8] = col[1]; d[9] = col[2]; d[10] = o.depth; d[11] = log2f(o.opacity); d[12] = bb.x; d[13] = bb.y;
    d[14] = 0.f; d[15] = 0.f;
    if (JAC) {
#pragma unroll
      for (int e = 0; e < 9; ++e) s_jac[tid * 9 + e] = J[e];
    }
    *reinterpret_cast<float2*>(sscale + 2 * m) = gsr_screen_scale(o.A, o.B, o.C);
    if (depth_keys) depth_keys[m] = gsr_depth_key(o.depth, key_bias, key_max);
  }
  __syncthreads();
  const int nrows = (int)min((int64_t)256, count - m0);
  float4* out4 = reinterpret_cast<float4*>(rows + (int64_t)GSR_ROW_FLOATS * m0);
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int e = it * 256 + tid, row = e >> 2, part = e & 3;
    if (row < nrows) {
      const float* d = s_out + row * OP + 4 * part;
      out4[e] = make_float4(d[0], d[1], d[2], d[3]);
    }
  }
  if (JAC) {
    float* jout = jac + 9 * m0;
    for (int e = tid; e < nrows * 9; e += 256) jout[e] = s_jac[e];
  }
#else
  if (!valid) return;
  float4* r = reinterpret_cast<float4*>(rows + (int64_t)GSR_ROW_FLOATS * m);
  r[0] = make_float4(o.u, o.v, o.A, o.B);
  r[1] = make_float4(o.C, o.opacity, gsr_qlim(o.opacity, rp), col[0]);
  r[2] = make_float4(col[1], col[2], o.depth, log2f(o.opacity));
  const float2 bb = gsr_pixel_bbox(o.u, o.v, o.A, o.B, o.C, o.opacity, rp);
  r[3] = make_float4(bb.x, bb.y, 0.f, 0.f);
  *reinterpret_cast<float2*>(sscale + 2 * m) = gsr_screen_scale(o.A, o.B, o.C);
  if (depth_keys) depth_keys[m] = gsr_depth_key(o.depth, key_bias, key_max);
  if (JAC) {
#pragma unroll
    for (int e = 0; e < 9; ++e) jac[9 * m + e] = J[e];
  }
#endif
}

// The (M,6) + (M,) + (M,C) tensors of the three-call form packed into the same rows (render_projected takes them from the
// caller: the reference's colour MLP sits between the projection and the rasterizer, mlp_scene.py:415-419).
template <int C>
__global__ __launch_bounds__(256) void pack_rows_kernel(const float* __restrict__ g2d, const float* __restrict__ depth,
                                                        const float* __restrict__ feat, int64_t M, GsrRasterParams rp,
                                                        float* __restrict__ rows, float* __restrict__ sscale) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const float* g = g2d + 6 * m;
  const float2 uv = *reinterpret_cast<const float2*>(g);
  const float2 ab = *reinterpret_cast<const float2*>(g + 2);
  const float2 co = *reinterpret_cast<const float2*>(g + 4);
  const float f0 = feat[C * m], f1 = C > 1 ? feat[C * m + 1] : 0.f, f2 = C > 2 ? feat[C * m + 2] : 0.f;
  float4* r = reinterpret_cast<float4*>(rows + (int64_t)GSR_ROW_FLOATS * m);
  r[0] = make_float4(uv.x, uv.y, ab.x, ab.y);
  r[1] = make_float4(co.x, co.y, gsr_qlim(co.y, rp), f0);
  r[2] = make_float4(f1, f2, depth[m], log2f(co.y));
  const float2 bb = gsr_pixel_bbox(uv.x, uv.y, ab.x, ab.y, co.x, co.y, rp);
  r[3] = make_float4(bb.x, bb.y, 0.f, 0.f);
  *reinterpret_cast<float2*>(sscale + 2 * m) = gsr_screen_scale(ab.x, ab.y, co.x);
}

// Backward of the fused kernel's geometry half, fed by the packed per-splat gradient rows the reduction leaves
// (binning.hip: reduce_grad_kernel):   du dv dA dB | dC dop prune split | df0 df1 df2 visibility | - - - -.
// One sequential sweep in splat order: K2 backward into the N-sized gradient tensors (rows ``idx``; "+=" when ACC), the
// position term of the colour gradient through the saved Jacobian, and the row's scalar columns copied out to the
// contiguous per-point outputs (prune_cost, split_score, visibility) and the colour gradient (M,3) the SH coefficient
// backward consumes.  ``dg2d_extra`` / ``ddepth``: gradients that reached gaussians2d / depth from outside the
// rasterizer (a regularizer on points.opacity / points.depths, mlp_scene.py:268-288); may be NULL.
// MODE 0: rows ``idx`` of the gradient tensors are written (the others untouched); 1: added to ("+=");
// 2: EVERY scene row is written -- zeros where the camera saw nothing -- so the caller needs neither a zero-filled buffer
// nor a read-modify-write (thread per scene row, ``inv`` = row -> visible rank or -1, NULL when every row is visible).
template <int MODE>
__global__ __launch_bounds__(256) void project_bwd_rows_kernel(
    const float* __restrict__ pos, const float* __restrict__ ls, const float* __restrict__ rot,
    const float* __restrict__ logit, const int64_t* __restrict__ idx, int64_t M, const int32_t* __restrict__ inv,
    int64_t N, const float* __restrict__ Tcw, const float* __restrict__ proj, GsrRasterParams rp,
    const float* __restrict__ rows, const float* __restrict__ grows, const float* __restrict__ dg2d_extra,
    const float* __restrict__ ddepth,
    const float* __restrict__ jac, float* __restrict__ dpos, float* __restrict__ dls, float* __restrict__ drot,
    float* __restrict__ dlogit, float* __restrict__ dcol_out, float* __restrict__ prune_out,
    float* __restrict__ split_out, float* __restrict__ vis_out) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t m, i;
  if (MODE == 2) {
    if (t >= N) return;
    i = t;
    m = inv ? (int64_t)inv[t] : t;
    if (m < 0) {                                              // not seen by this camera: zero gradient
      dpos[3 * i] = 0.f; dpos[3 * i + 1] = 0.f; dpos[3 * i + 2] = 0.f;
      dls[3 * i] = 0.f; dls[3 * i + 1] = 0.f; dls[3 * i + 2] = 0.f;
      *reinterpret_cast<float4*>(drot + 4 * i) = make_float4(0.f, 0.f, 0.f, 0.f);
      dlogit[i] = 0.f;
      return;
    }
  } else {
    if (t >= M) return;
    m = t;
    i = dpos ? idx[m] : 0;
  }
  const float4* gr = reinterpret_cast<const float4*>(grows + (int64_t)GSR_ROW_FLOATS * m);
  const float4 g0 = gr[0], g1 = gr[1], g2 = gr[2];
  if (prune_out) prune_out[m] = g1.z;
  if (split_out) split_out[m] = g1.w;
  if (vis_out) vis_out[m] = g2.w;
  if (dcol_out) { dcol_out[3 * m] = g2.x; dcol_out[3 * m + 1] = g2.y; dcol_out[3 * m + 2] = g2.z; }
  if (!dpos) return;
  const GsrCam cam = gsr_load_cam(Tcw, proj);
  float p[3] = {pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]};
  float s[3] = {ls[3 * i], ls[3 * i + 1], ls[3 * i + 2]};
  const float4 qv = *reinterpret_cast<const float4*>(rot + 4 * i);
  float q[4] = {qv.x, qv.y, qv.z, qv.w};
  // the rows carry the MOMENTS of G dL/dG about the mean (composite.hip, K7): mx my mxx mxy | myy dop ...; with the
  // conic the composite kernels used they become d(u, v, A, B, C).  The conic is RECOMPUTED from the parameters this
  // thread holds anyway -- the forward projection's rounding is pinned, so these are the forward row's bits -- instead
  // of fetched from the row table (a 64-byte line per splat for 12 bytes: +35 us at 3M splats, measured).
  const GsrProjected fo = gsr_project_one(cam, rp, p, s, q, logit[i]);
  const float cA = fo.A, cB = fo.B, cC = fo.C;
  float g[6] = {cA * g0.x + cB * g0.y, cB * g0.x + cC * g0.y, -0.5f * g0.z, -g0.w, -0.5f * g1.x,
                fo.opacity > 0.f ? g1.y / fo.opacity : 0.f};
  if (dg2d_extra) {
#pragma unroll
    for (int k = 0; k < 6; ++k) g[k] += dg2d_extra[6 * m + k];
  }
  const float gd = ddepth ? ddepth[m] : 0.f;
  GsrProjectGrad o = gsr_project_one_bwd(cam, rp, p, s, q, logit[i], g, gd);
  if (jac) {
    const float g3[3] = {g2.x, g2.y, g2.z};
    float pj[3];
    gsr_jac_apply(g3, jac + 9 * m, pj);
    o.dp[0] += pj[0]; o.dp[1] += pj[1]; o.dp[2] += pj[2];
  }
  constexpr bool ACC = MODE == 1;
  float4 r = make_float4(o.dq[0], o.dq[1], o.dq[2], o.dq[3]);
  if (ACC) {
    const float4 old = *reinterpret_cast<const float4*>(drot + 4 * i);
    r.x += old.x; r.y += old.y; r.z += old.z; r.w += old.w;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dpos[3 * i + k] = ACC ? dpos[3 * i + k] + o.dp[k] : o.dp[k];
    dls[3 * i + k] = ACC ? dls[3 * i + k] + o.dls[k] : o.dls[k];
  }
  *reinterpret_cast<float4*>(drot + 4 * i) = r;
  dlogit[i] = ACC ? dlogit[i] + o.dlogit : o.dlogit;
}

template <int K, bool ACC>
__global__ __launch_bounds__(256) void sh_bwd_kernel(const float* __restrict__ dcol, const float* __restrict__ sh,
                                                     const float* __restrict__ pos, const int64_t* __restrict__ idx,
                                                     int64_t M, const float* __restrict__ cam_pos,
                                                     float* __restrict__ dsh, float* __restrict__ dpos,
                                                     const float* __restrict__ jac) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int64_t i = idx[m];
  float vx = pos[3 * i] - cam_pos[0], vy = pos[3 * i + 1] - cam_pos[1], vz = pos[3 * i + 2] - cam_pos[2];
  float inv = 1.f / sqrtf(vx * vx + vy * vy + vz * vz);
  const float x = vx * inv, y = vy * inv, z = vz * inv;
  float Y[K];
  gsr_sh_basis<K>(x, y, z, Y);
  float* row = dsh + (int64_t)3 * K * i;
  const float g3[3] = {dcol[3 * m], dcol[3 * m + 1], dcol[3 * m + 2]};
#pragma unroll
  for (int ch = 0; ch < 3; ++ch) {
    const float g = g3[ch];
    if (K % 4 == 0) {
#pragma unroll
      for (int k = 0; k < K; k += 4) {
        float4 v = make_float4(g * Y[k], g * Y[k + 1], g * Y[k + 2], g * Y[k + 3]);
        if (ACC) {
          const float4 old = *reinterpret_cast<const float4*>(row + ch * K + k);
          v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
        }
        *reinterpret_cast<float4*>(row + ch * K + k) = v;
      }
    } else {
#pragma unroll
      for (int k = 0; k < K; ++k) row[ch * K + k] = ACC ? row[ch * K + k] + g * Y[k] : g * Y[k];
    }
  }
  if (dpos != nullptr && jac != nullptr) {
    float pj[3];
    gsr_jac_apply(g3, jac + 9 * m, pj);
    dpos[3 * i] = ACC ? dpos[3 * i] + pj[0] : pj[0];
    dpos[3 * i + 1] = ACC ? dpos[3 * i + 1] + pj[1] : pj[1];
    dpos[3 * i + 2] = ACC ? dpos[3 * i + 2] + pj[2] : pj[2];
  } else if (dpos != nullptr && K > 1) {
    // colour depends on the point through the view direction d = v/|v|:  dL/dp = (I - d d^T)/|v| * dL/dd
    float dYx[K], dYy[K], dYz[K];
    gsr_sh_basis_grad<K>(x, y, z, dYx, dYy, dYz);
    const float* c = sh + (int64_t)3 * K * i;
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
      for (int k = 1; k < K; ++k) {
        const float w = g3[ch] * c[ch * K + k];
        gx += w * dYx[k]; gy += w * dYy[k]; gz += w * dYz[k];
      }
    }
    const float dot = gx * x + gy * y + gz * z;
    const float px_ = (gx - x * dot) * inv, py_ = (gy - y * dot) * inv, pz_ = (gz - z * dot) * inv;
    dpos[3 * i] = ACC ? dpos[3 * i] + px_ : px_;
    dpos[3 * i + 1] = ACC ? dpos[3 * i + 1] + py_ : py_;
    dpos[3 * i + 2] = ACC ? dpos[3 * i + 2] + pz_ : pz_;
  }
}

// Row -> visible-rank map of an ascending index list: inv[idx[m]] = m and -1 for the rows in the gaps.  Thread m also
// fills the gap in front of its row (the last thread the tail), so no pre-fill pass is needed.
__global__ __launch_bounds__(256) void inverse_map_kernel(const int64_t* __restrict__ idx, int64_t M, int64_t N,
                                                          int32_t* __restrict__ inv) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m > M) return;
  const int64_t lo = m > 0 ? idx[m - 1] + 1 : 0;
  const int64_t hi = m < M ? idx[m] : N;
  for (int64_t r = lo; r < hi; ++r) inv[r] = -1;
  if (m < M) inv[hi] = (int32_t)m;
}

// SH backward that OVERWRITES the whole (N,3,K) gradient: rows a camera did not see get zeros, so the caller needs
// neither a zero-fill of the buffer nor a read-modify-write of the visible rows (one 4*3K-byte store per row instead of
// fill + load + store).  A block handles 256 consecutive rows: each thread forms its row in LDS (row pitch 3K+1 words:
// conflict-free), then the block streams the 256 rows out with fully coalesced stores.  The position gradient is
// accumulated ("+=") as in sh_bwd_kernel<K, true>.
template <int K>
__global__ __launch_bounds__(256) void sh_bwd_dense_kernel(const float* __restrict__ dcol, const float* __restrict__ sh,
                                                           const float* __restrict__ pos,
                                                           const int32_t* __restrict__ inv, int64_t N,
                                                           const float* __restrict__ cam_pos,
                                                           float* __restrict__ dsh, float* __restrict__ dpos,
                                                           const float* __restrict__ jac) {
  constexpr int ROW = 3 * K, PITCH = ROW + 1;
  __shared__ float s_rows[256 * PITCH];
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int64_t i = row0 + threadIdx.x;
  float* mine = s_rows + threadIdx.x * PITCH;
  const int64_t m = (i < N) ? (inv ? (int64_t)inv[i] : i) : -1;
  if (m >= 0) {
    const float vx = pos[3 * i] - cam_pos[0], vy = pos[3 * i + 1] - cam_pos[1], vz = pos[3 * i + 2] - cam_pos[2];
    const float rinv = 1.f / sqrtf(vx * vx + vy * vy + vz * vz);
    const float x = vx * rinv, y = vy * rinv, z = vz * rinv;
    float Y[K];
    gsr_sh_basis<K>(x, y, z, Y);
    const float g3[3] = {dcol[3 * m], dcol[3 * m + 1], dcol[3 * m + 2]};
#pragma unroll
    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
      for (int k = 0; k < K; ++k) mine[ch * K + k] = g3[ch] * Y[k];
    if (dpos != nullptr && jac != nullptr) {
      float pj[3];
      gsr_jac_apply(g3, jac + 9 * m, pj);
      dpos[3 * i] += pj[0];
      dpos[3 * i + 1] += pj[1];
      dpos[3 * i + 2] += pj[2];
    } else if (dpos != nullptr && K > 1) {
      float dYx[K], dYy[K], dYz[K];
      gsr_sh_basis_grad<K>(x, y, z, dYx, dYy, dYz);
      const float* c = sh + (int64_t)ROW * i;
      float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
#pragma unroll
        for (int k = 1; k < K; ++k) {
          const float w = g3[ch] * c[ch * K + k];
          gx += w * dYx[k]; gy += w * dYy[k]; gz += w * dYz[k];
        }
      }
      const float dot = gx * x + gy * y + gz * z;
      dpos[3 * i] += (gx - x * dot) * rinv;
      dpos[3 * i + 1] += (gy - y * dot) * rinv;
      dpos[3 * i + 2] += (gz - z * dot) * rinv;
    }
  } else {
#pragma unroll
    for (int e = 0; e < ROW; ++e) mine[e] = 0.f;
  }
  __syncthreads();
  const int64_t rows_here = (N - row0) < 256 ? (N - row0) : 256;
  const int total = (int)rows_here * ROW;
  float* out = dsh + row0 * ROW;
  for (int e = threadIdx.x; e < total; e += 256) out[e] = s_rows[(e / ROW) * PITCH + (e % ROW)];
}

// Multi-camera SH backward for the data-parallel path: instead of all-reducing the (N,3,K) coefficient gradient (81 %
// of the gradient bytes at K = 16), ranks all-gather the (cameras,N,3) colour gradients -- 16x smaller -- and every
// rank rebuilds  d_sh[i] = sum_c g_c[i] (x) Y(dir_c(i))  locally, cameras in index order (deterministic and identical
// on every rank, unlike the association order of a ring all-reduce).  Positions and coefficients are replicated, so
// the basis and the view-direction Jacobian are recomputed from them.  Rows with an all-zero colour gradient (splat
// not seen by that camera) are skipped.
template <int K>
__global__ __launch_bounds__(256) void sh_bwd_multi_kernel(const float* __restrict__ G, int64_t g_stride,
                                                           const float* __restrict__ cams, int64_t cam_stride,
                                                           int ncam, const float* __restrict__ sh,
                                                           const float* __restrict__ pos, int64_t N,
                                                           float* __restrict__ dsh, float* __restrict__ dpos,
                                                           int accumulate) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
  const float* coef = sh + (int64_t)3 * K * i;
  float acc[3][K];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[ch][k] = 0.f;
  float gpx = 0.f, gpy = 0.f, gpz = 0.f;
  bool any = false;
  for (int c = 0; c < ncam; ++c) {
    const float* g = G + (int64_t)c * g_stride + i * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
    any = true;
    const float* cp = cams + (int64_t)c * cam_stride;
    const float vx = px - cp[0], vy = py - cp[1], vz = pz - cp[2];
    const float inv = 1.f / sqrtf(vx * vx + vy * vy + vz * vz);
    const float x = vx * inv, y = vy * inv, z = vz * inv;
    float Y[K];
    gsr_sh_basis<K>(x, y, z, Y);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      acc[0][k] += g0 * Y[k]; acc[1][k] += g1 * Y[k]; acc[2][k] += g2 * Y[k];
    }
    if (dpos != nullptr && K > 1) {
      float dYx[K], dYy[K], dYz[K];
      gsr_sh_basis_grad<K>(x, y, z, dYx, dYy, dYz);
      float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
      for (int k = 1; k < K; ++k) {
        const float w = g0 * coef[k] + g1 * coef[K + k] + g2 * coef[2 * K + k];
        gx += w * dYx[k]; gy += w * dYy[k]; gz += w * dYz[k];
      }
      const float dot = gx * x + gy * y + gz * z;
      gpx += (gx - x * dot) * inv; gpy += (gy - y * dot) * inv; gpz += (gz - z * dot) * inv;
    }
  }
  if (!any && accumulate) return;
  float* row = dsh + (int64_t)3 * K * i;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) row[ch * K + k] = accumulate ? row[ch * K + k] + acc[ch][k] : acc[ch][k];
  if (any && dpos != nullptr && K > 1) {
    dpos[3 * i] += gpx; dpos[3 * i + 1] += gpy; dpos[3 * i + 2] += gpz;
  }
}

// The same rebuild in its streaming form (overwrite, no position term -- every rank has added its own cameras' position
// term before the all-reduce): a block forms 256 consecutive rows in LDS (row pitch 3K+1 words: conflict-free) and
// streams them out with fully coalesced stores, as sh_bwd_dense_kernel does; nothing but the colour-gradient blocks and
// the positions is read.  3 M points, 8 cameras: 902 us per-thread rows -> see DESIGN.md section 6.
template <int K>
__global__ __launch_bounds__(256) void sh_bwd_multi_stream_kernel(const float* __restrict__ G, int64_t g_stride,
                                                                  const float* __restrict__ cams, int64_t cam_stride,
                                                                  int ncam, const float* __restrict__ pos, int64_t N,
                                                                  float* __restrict__ dsh) {
  constexpr int ROW = 3 * K, PITCH = ROW + 1;
  __shared__ float s_rows[256 * PITCH];
  const int64_t row0 = (int64_t)blockIdx.x * 256;
  const int64_t i = row0 + threadIdx.x;
  float acc[3][K];
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) acc[ch][k] = 0.f;
  if (i < N) {
    const float px = pos[3 * i], py = pos[3 * i + 1], pz = pos[3 * i + 2];
    for (int c = 0; c < ncam; ++c) {                               // cameras in index order: the same sums on every rank
      const float* g = G + (int64_t)c * g_stride + i * 3;
      const float g0 = g[0], g1 = g[1], g2 = g[2];
      if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
      const float* cp = cams + (int64_t)c * cam_stride;
      const float vx = px - cp[0], vy = py - cp[1], vz = pz - cp[2];
      const float inv = 1.f / sqrtf(vx * vx + vy * vy + vz * vz);
      float Y[K];
      gsr_sh_basis<K>(vx * inv, vy * inv, vz * inv, Y);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        acc[0][k] += g0 * Y[k]; acc[1][k] += g1 * Y[k]; acc[2][k] += g2 * Y[k];
      }
    }
  }
  float* mine = s_rows + threadIdx.x * PITCH;
#pragma unroll
  for (int ch = 0; ch < 3; ++ch)
#pragma unroll
    for (int k = 0; k < K; ++k) mine[ch * K + k] = acc[ch][k];
  __syncthreads();
  const int64_t rows_here = (N - row0) < 256 ? (N - row0) : 256;
  const int total = (int)rows_here * ROW;
  float* out = dsh + row0 * ROW;
  for (int e = threadIdx.x; e < total; e += 256) out[e] = s_rows[(e / ROW) * PITCH + (e % ROW)];
}

inline GsrRasterParams to_params(const GsrRasterParamsC* c) {
  GsrRasterParams rp;
  __builtin_memcpy(&rp, c, sizeof(rp));
  return rp;
}

inline unsigned grid_for(int64_t n, int block) { return (unsigned)((n + block - 1) / block); }

}  // namespace

extern "C" {

int gsr_abi_version(void) { return 30; }

const char* gsr_error_string(int code) {
  switch (code) {
    case GSR_OK: return "ok";
    case GSR_ERR_INVALID_ARGUMENT: return "invalid argument";
    case GSR_ERR_WORKSPACE_TOO_SMALL: return "workspace too small";
    case GSR_ERR_LAUNCH_FAILED: return "kernel launch failed";
    case GSR_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

size_t gsr_cull_workspace_bytes(int64_t N) {
  if (N < 0) N = 0;
  int64_t nw = (N + CULL_WAVE_SPAN - 1) / CULL_WAVE_SPAN;
  size_t counts = (((size_t)nw * sizeof(uint32_t) + 255) / 256) * 256;
  return counts + gsr_scan_workspace_bytes(nw) + 256;
}

int gsr_frustum_cull(const float* position, int64_t N, const float* T_camera_world, const float* projection, int32_t W,
                     int32_t H, float near_plane, float far_plane, float margin_px, int64_t* indexes_out,
                     uint32_t* count_dev, void* workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || N > 0x7FFFFFFFll || !count_dev || !T_camera_world || !projection) return GSR_ERR_INVALID_ARGUMENT;
  if (N == 0) { hipMemsetAsync(count_dev, 0, sizeof(uint32_t), stream); return GSR_OK; }
  if (!position || !indexes_out) return GSR_ERR_INVALID_ARGUMENT;
  if (!workspace || workspace_bytes < gsr_cull_workspace_bytes(N)) return GSR_ERR_WORKSPACE_TOO_SMALL;
  const int64_t nw = (N + CULL_WAVE_SPAN - 1) / CULL_WAVE_SPAN;
  uint32_t* wave_counts = reinterpret_cast<uint32_t*>(workspace);
  size_t counts_bytes = (((size_t)nw * sizeof(uint32_t) + 255) / 256) * 256;
  uint8_t* scan_ws = reinterpret_cast<uint8_t*>(workspace) + counts_bytes;
  const unsigned blocks = grid_for(nw, 4);
  cull_count_kernel<<<blocks, 256, 0, stream>>>(position, N, T_camera_world, projection, W, H, near_plane, far_plane,
                                               margin_px, wave_counts, nw);
  GSR_CHECK_LAUNCH();
  const bool self_prefix = nw <= CULL_SELF_PREFIX_MAX_WAVES;
  if (!self_prefix) {
    int rc = gsr_exclusive_scan_u32(wave_counts, wave_counts, nw, count_dev, scan_ws, workspace_bytes - counts_bytes, stream_);
    if (rc != GSR_OK) return rc;
  }
  cull_write_kernel<<<blocks, 256, 0, stream>>>(position, N, T_camera_world, projection, W, H, near_plane, far_plane,
                                               margin_px, wave_counts, nw, indexes_out, self_prefix ? count_dev : nullptr);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_project_forward(const float* position, const float* log_scaling, const float* rotation_xyzw,
                        const float* alpha_logit, const int64_t* indexes, int64_t M, const float* T_camera_world,
                        const float* projection, const GsrRasterParamsC* params_host, float* gaussians2d_out,
                        float* depth_out, const uint32_t* count_dev, uint32_t* depth_keys_out, uint32_t depth_key_bias,
                        uint32_t depth_key_max, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || !params_host) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!position || !log_scaling || !rotation_xyzw || !alpha_logit || !indexes || !T_camera_world || !projection ||
      !gaussians2d_out || !depth_out)
    return GSR_ERR_INVALID_ARGUMENT;
  project_fwd_kernel<<<grid_for(M, 256), 256, 0, stream>>>(position, log_scaling, rotation_xyzw, alpha_logit, indexes, M,
                                                          T_camera_world, projection, to_params(params_host),
                                                          gaussians2d_out, depth_out, count_dev, depth_keys_out,
                                                          depth_key_bias, depth_key_max);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_project_backward(const float* position, const float* log_scaling, const float* rotation_xyzw,
                         const float* alpha_logit, const int64_t* indexes, int64_t M, const float* T_camera_world,
                         const float* projection, const GsrRasterParamsC* params_host, const float* dL_dgaussians2d,
                         const float* dL_ddepth, float* d_position, float* d_log_scaling, float* d_rotation,
                         float* d_alpha_logit, int32_t accumulate, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || !params_host) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!position || !log_scaling || !rotation_xyzw || !alpha_logit || !indexes || !T_camera_world || !projection ||
      !dL_dgaussians2d || !d_position || !d_log_scaling || !d_rotation || !d_alpha_logit)
    return GSR_ERR_INVALID_ARGUMENT;
  if (accumulate)
    project_bwd_kernel<true><<<grid_for(M, 256), 256, 0, stream>>>(position, log_scaling, rotation_xyzw, alpha_logit,
                                                                  indexes, M, T_camera_world, projection,
                                                                  to_params(params_host), dL_dgaussians2d, dL_ddepth,
                                                                  d_position, d_log_scaling, d_rotation, d_alpha_logit);
  else
    project_bwd_kernel<false><<<grid_for(M, 256), 256, 0, stream>>>(position, log_scaling, rotation_xyzw, alpha_logit,
                                                                   indexes, M, T_camera_world, projection,
                                                                   to_params(params_host), dL_dgaussians2d, dL_ddepth,
                                                                   d_position, d_log_scaling, d_rotation, d_alpha_logit);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_depth_keys_from_positions(const float* position, const int64_t* indexes, int64_t M, const uint32_t* count_dev,
                                  const float* T_camera_world, const float* projection, uint32_t depth_key_bias,
                                  uint32_t depth_key_max, uint32_t* keys_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!position || !indexes || !T_camera_world || !projection || !keys_out) return GSR_ERR_INVALID_ARGUMENT;
  depth_keys_pos_kernel<<<grid_for(M, 256), 256, 0, stream>>>(position, indexes, M, count_dev, T_camera_world, projection,
                                                             keys_out, depth_key_bias, depth_key_max);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_project_sh_forward(const float* position, const float* log_scaling, const float* rotation_xyzw,
                           const float* alpha_logit, const float* sh_features, int32_t K, const int64_t* indexes, int64_t M,
                           const float* T_camera_world, const float* projection, const float* camera_pos,
                           const GsrRasterParamsC* params_host, float* rows_out, float* screen_scale_out,
                           float* jacobian_out, const uint32_t* count_dev, uint32_t* depth_keys_out,
                           uint32_t depth_key_bias, uint32_t depth_key_max, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || !params_host) return GSR_ERR_INVALID_ARGUMENT;
  if (K != 1 && K != 4 && K != 9 && K != 16) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!position || !log_scaling || !rotation_xyzw || !alpha_logit || !sh_features || !indexes || !T_camera_world ||
      !projection || !camera_pos || !rows_out || !screen_scale_out)
    return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(M, 256);
  const GsrRasterParams rp = to_params(params_host);
#define GSR_LAUNCH_PSF(KK, JJ)                                                                                          \
  project_sh_fwd_kernel<KK, JJ><<<g, 256, 0, stream>>>(position, log_scaling, rotation_xyzw, alpha_logit, sh_features,   \
                                                       indexes, M, T_camera_world, projection, camera_pos, rp, rows_out, \
                                                       screen_scale_out, jacobian_out, count_dev, depth_keys_out,        \
                                                       depth_key_bias, depth_key_max)
  const bool jj = jacobian_out != nullptr;
  switch (K) {
    case 1: if (jj) GSR_LAUNCH_PSF(1, true); else GSR_LAUNCH_PSF(1, false); break;
    case 4: if (jj) GSR_LAUNCH_PSF(4, true); else GSR_LAUNCH_PSF(4, false); break;
    case 9: if (jj) GSR_LAUNCH_PSF(9, true); else GSR_LAUNCH_PSF(9, false); break;
    default: if (jj) GSR_LAUNCH_PSF(16, true); else GSR_LAUNCH_PSF(16, false); break;
  }
#undef GSR_LAUNCH_PSF
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_pack_rows(const float* gaussians2d, const float* depth, const float* features, int64_t M, int32_t C,
                  const GsrRasterParamsC* params_host, float* rows_out, float* screen_scale_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || !params_host) return GSR_ERR_INVALID_ARGUMENT;
  const GsrRasterParams rp = to_params(params_host);
  if (C < 1 || C > 3) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!gaussians2d || !depth || !features || !rows_out || !screen_scale_out) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(M, 256);
  if (C == 1) pack_rows_kernel<1><<<g, 256, 0, stream>>>(gaussians2d, depth, features, M, rp, rows_out, screen_scale_out);
  else if (C == 2) pack_rows_kernel<2><<<g, 256, 0, stream>>>(gaussians2d, depth, features, M, rp, rows_out, screen_scale_out);
  else pack_rows_kernel<3><<<g, 256, 0, stream>>>(gaussians2d, depth, features, M, rp, rows_out, screen_scale_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_project_backward_rows(const float* position, const float* log_scaling, const float* rotation_xyzw,
                              const float* alpha_logit, const int64_t* indexes, int64_t M, const int32_t* inverse,
                              int64_t N, const float* T_camera_world, const float* projection,
                              const GsrRasterParamsC* params_host, const float* rows, const float* grad_rows,
                              const float* dL_dgaussians2d_extra, const float* dL_ddepth, const float* jacobian,
                              float* d_position, float* d_log_scaling, float* d_rotation, float* d_alpha_logit,
                              int32_t mode, float* d_colors_out, float* prune_cost_out, float* split_score_out,
                              float* visibility_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 0 || M > N || !params_host || mode < 0 || mode > 2) return GSR_ERR_INVALID_ARGUMENT;
  const bool geom = d_position != nullptr;
  if (geom && (!position || !log_scaling || !rotation_xyzw || !alpha_logit || !T_camera_world || !projection ||
               !d_log_scaling || !d_rotation || !d_alpha_logit || (M > 0 && !indexes)))
    return GSR_ERR_INVALID_ARGUMENT;
  if (mode == 2 && (!geom || (!inverse && M != N))) return GSR_ERR_INVALID_ARGUMENT;   // identity map only when all rows are visible
  if (mode != 2 && M == 0) return GSR_OK;
  if (mode == 2 && N == 0) return GSR_OK;
  if (M > 0 && !grad_rows) return GSR_ERR_INVALID_ARGUMENT;
  const GsrRasterParams rp = to_params(params_host);
#define GSR_LAUNCH_PBR(MODE, COUNT)                                                                                      \
  project_bwd_rows_kernel<MODE><<<grid_for(COUNT, 256), 256, 0, stream>>>(                                               \
      position, log_scaling, rotation_xyzw, alpha_logit, indexes, M, inverse, N, T_camera_world, projection, rp, rows,   \
      grad_rows, dL_dgaussians2d_extra, dL_ddepth, jacobian, d_position, d_log_scaling, d_rotation, d_alpha_logit,       \
      d_colors_out, prune_cost_out, split_score_out, visibility_out)
  if (mode == 2) GSR_LAUNCH_PBR(2, N);
  else if (mode == 1) GSR_LAUNCH_PBR(1, M);
  else GSR_LAUNCH_PBR(0, M);
#undef GSR_LAUNCH_PBR
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_sh_forward(const float* sh_features, const float* positions, const int64_t* indexes, int64_t M, int32_t K,
                   const float* camera_pos, float* colors_out, float* jacobian_out, const uint32_t* count_dev,
                   void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (K != 1 && K != 4 && K != 9 && K != 16) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!sh_features || !positions || !indexes || !camera_pos || !colors_out) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(M, 256);
  switch (K) {
    case 1:
      if (jacobian_out) sh_fwd_kernel<1, true><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, jacobian_out, count_dev);
      else sh_fwd_kernel<1, false><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, nullptr, count_dev);
      break;
    case 4:
      if (jacobian_out) sh_fwd_kernel<4, true><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, jacobian_out, count_dev);
      else sh_fwd_kernel<4, false><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, nullptr, count_dev);
      break;
    case 9:
      if (jacobian_out) sh_fwd_kernel<9, true><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, jacobian_out, count_dev);
      else sh_fwd_kernel<9, false><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, nullptr, count_dev);
      break;
    default:
      if (jacobian_out) sh_fwd_kernel<16, true><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, jacobian_out, count_dev);
      else sh_fwd_kernel<16, false><<<g, 256, 0, stream>>>(sh_features, positions, indexes, M, camera_pos, colors_out, nullptr, count_dev);
      break;
  }
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_sh_backward(const float* dL_dcolors, const float* sh_features, const float* positions, const int64_t* indexes,
                    int64_t M, int32_t K, const float* camera_pos, const float* jacobian, float* d_sh_features,
                    float* d_positions, int32_t accumulate, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (K != 1 && K != 4 && K != 9 && K != 16) return GSR_ERR_UNSUPPORTED;
  if (M == 0) return GSR_OK;
  if (!dL_dcolors || !sh_features || !positions || !indexes || !camera_pos || !d_sh_features)
    return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(M, 256);
  switch (K) {
    case 1:
      if (accumulate) sh_bwd_kernel<1, true><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      else sh_bwd_kernel<1, false><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      break;
    case 4:
      if (accumulate) sh_bwd_kernel<4, true><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      else sh_bwd_kernel<4, false><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      break;
    case 9:
      if (accumulate) sh_bwd_kernel<9, true><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      else sh_bwd_kernel<9, false><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      break;
    default:
      if (accumulate) sh_bwd_kernel<16, true><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      else sh_bwd_kernel<16, false><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, indexes, M, camera_pos, d_sh_features, d_positions, jacobian);
      break;
  }
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_inverse_map(const int64_t* indexes, int64_t M, int64_t N, int32_t* inverse_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 0 || M > N || N > 0x7FFFFFFFll) return GSR_ERR_INVALID_ARGUMENT;
  if (N == 0) return GSR_OK;
  if (!inverse_out || (M > 0 && !indexes)) return GSR_ERR_INVALID_ARGUMENT;
  inverse_map_kernel<<<grid_for(M + 1, 256), 256, 0, stream>>>(indexes, M, N, inverse_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_sh_backward_dense(const float* dL_dcolors, const float* sh_features, const float* positions,
                          const int32_t* inverse, int64_t M, int64_t N, int32_t K, const float* camera_pos,
                          const float* jacobian, float* d_sh_features, float* d_positions, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || N < 0 || M > N) return GSR_ERR_INVALID_ARGUMENT;
  if (K != 1 && K != 4 && K != 9 && K != 16) return GSR_ERR_UNSUPPORTED;
  if (N == 0) return GSR_OK;
  if (!inverse && M != N) return GSR_ERR_INVALID_ARGUMENT;          // identity map only when every row is visible
  if (!sh_features || !positions || !camera_pos || !d_sh_features || (M > 0 && !dL_dcolors)) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(N, 256);
  switch (K) {
    case 1: sh_bwd_dense_kernel<1><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, inverse, N, camera_pos, d_sh_features, d_positions, jacobian); break;
    case 4: sh_bwd_dense_kernel<4><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, inverse, N, camera_pos, d_sh_features, d_positions, jacobian); break;
    case 9: sh_bwd_dense_kernel<9><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, inverse, N, camera_pos, d_sh_features, d_positions, jacobian); break;
    default: sh_bwd_dense_kernel<16><<<g, 256, 0, stream>>>(dL_dcolors, sh_features, positions, inverse, N, camera_pos, d_sh_features, d_positions, jacobian); break;
  }
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_sh_backward_multi(const float* dL_dcolors_dense, int64_t dense_stride, const float* camera_positions,
                          int64_t camera_stride, int32_t num_cameras, const float* sh_features, const float* positions, int64_t N, int32_t K, float* d_sh_features,
                          float* d_positions, int32_t accumulate, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (N < 0 || num_cameras < 0 || dense_stride < 3 * N || camera_stride < 3) return GSR_ERR_INVALID_ARGUMENT;
  if (K != 1 && K != 4 && K != 9 && K != 16) return GSR_ERR_UNSUPPORTED;
  if (N == 0 || (num_cameras == 0 && accumulate)) return GSR_OK;
  if (!dL_dcolors_dense || !camera_positions || !sh_features || !positions || !d_sh_features) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned g = grid_for(N, 256);
  if (!accumulate && !d_positions && num_cameras > 0) {            // overwrite, no position term: the streaming form
    switch (K) {
      case 1: sh_bwd_multi_stream_kernel<1><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, positions, N, d_sh_features); break;
      case 4: sh_bwd_multi_stream_kernel<4><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, positions, N, d_sh_features); break;
      case 9: sh_bwd_multi_stream_kernel<9><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, positions, N, d_sh_features); break;
      default: sh_bwd_multi_stream_kernel<16><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, positions, N, d_sh_features); break;
    }
    GSR_CHECK_LAUNCH();
    return GSR_OK;
  }
  switch (K) {
    case 1: sh_bwd_multi_kernel<1><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, sh_features, positions, N, d_sh_features, d_positions, accumulate); break;
    case 4: sh_bwd_multi_kernel<4><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, sh_features, positions, N, d_sh_features, d_positions, accumulate); break;
    case 9: sh_bwd_multi_kernel<9><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, sh_features, positions, N, d_sh_features, d_positions, accumulate); break;
    default: sh_bwd_multi_kernel<16><<<g, 256, 0, stream>>>(dL_dcolors_dense, dense_stride, camera_positions, camera_stride, num_cameras, sh_features, positions, N, d_sh_features, d_positions, accumulate); break;
  }
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
