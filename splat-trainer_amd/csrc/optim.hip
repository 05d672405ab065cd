// Sparse, visibility-aware optimizer step over the rows a batch has seen.
//
// Replaces, behind ParameterClass.step(visibility=, indexes=, basis=) (reference call site
// splat_trainer/scene/mlp_scene.py:214-230; options :58-60; parameter groups config/scene/mlp.yaml:8-14), the Taichi
// kernels of taichi_splatting.optim, which are not part of the reference tree.  The arithmetic is specified in
// oracle/optim_oracle.py (published Adam / LaProp + the visibility weighting the reference's options name; PARITY
// UNPINNED for the visibility-aware part) and restated here:
//
//   opt_point_weights   once per step, thread per visible row: per-point step count t and running visibility, and the
//                       four per-row factors every group kernel needs (1/(w + smooth), rho/(1 - beta1^t),
//                       1/(1 - beta2^t), spare)
//   opt_step<TYPE>      once per parameter group: read grad, moments and parameter of the visible rows only, write the
//                       moments and the parameter back in place -- 5 to 7 accesses of 4 bytes per element, nothing else
//                       touches HBM (a torch formulation gathers, updates and scatters every column: ~3x the traffic
//                       and dozens of launches).  Narrow rows (D <= 4: position, log_scaling, rotation, alpha_logit)
//                       take one lane per row; wide rows (features) take 16 lanes per row so that a row is one
//                       contiguous burst, with a DPP-free 4-step xor reduction for the row norm of `vector` groups.
//
// HBM-bound integer/float streaming; no LDS, no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "gsr_device.h"

namespace {

enum : int { OPT_SCALAR = 0, OPT_VECTOR = 1, OPT_LOCAL_VECTOR = 2 };
enum : int { OPT_ADAM = 0, OPT_LAPROP = 1 };

__global__ __launch_bounds__(256) void opt_point_weights_kernel(const int64_t* __restrict__ indexes,
                                                                const float* __restrict__ visibility, int64_t M,
                                                                float* __restrict__ step, float* __restrict__ vis_avg,
                                                                float beta1, float beta2, float vis_beta,
                                                                float vis_smooth, int bias_correction,
                                                                float4* __restrict__ row_scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int64_t idx = indexes[i];
  const float t = step[idx] + 1.0f;
  step[idx] = t;
  float inv_w = 1.0f, rho = 1.0f;
  if (visibility) {
    const float w = visibility[i];
    const float avg = vis_beta * vis_avg[idx] + (1.0f - vis_beta) * w;
    vis_avg[idx] = avg;
    const float avg_hat = bias_correction ? avg / (1.0f - powf(vis_beta, t)) : avg;
    inv_w = 1.0f / (w + vis_smooth);
    rho = w / (avg_hat + vis_smooth);
  }
  const float bc1 = bias_correction ? 1.0f - powf(beta1, t) : 1.0f;
  const float bc2 = bias_correction ? 1.0f - powf(beta2, t) : 1.0f;
  row_scale[i] = make_float4(inv_w, rho / bc1, 1.0f / bc2, 0.0f);
}

struct OptArgs {
  float lr, beta1, beta2, eps, grad_clip;
  int algo;
};

// one element: returns the parameter decrement (before the basis rotation of local_vector groups)
__device__ __forceinline__ float opt_element(float g, float second, float& m, const OptArgs& a, float step_scale,
                                             float inv_bc2) {
  const float denom = sqrtf(second * inv_bc2) + a.eps;
  if (a.algo == OPT_LAPROP) {
    float u = g / denom;
    if (a.grad_clip > 0.0f) u = fminf(fmaxf(u, -a.grad_clip), a.grad_clip);
    m = a.beta1 * m + (1.0f - a.beta1) * u;
    return a.lr * step_scale * m;
  }
  m = a.beta1 * m + (1.0f - a.beta1) * g;
  return a.lr * step_scale * m / denom;
}

// D <= 4, one lane per visible row
template <int D, int TYPE>
__global__ __launch_bounds__(256) void opt_step_narrow_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                              float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                              const int64_t* __restrict__ indexes,
                                                              const float4* __restrict__ row_scale,
                                                              const float* __restrict__ basis, int64_t M, OptArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int64_t idx = indexes[i];
  const float4 rs = row_scale[i];
  float g[D], m[D], dec[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    g[j] = grad[idx * D + j] * rs.x;
    m[j] = exp_avg[idx * D + j];
  }
  float B[9];
  if (TYPE == OPT_LOCAL_VECTOR) {             // gradient with respect to the splat's local coordinates: B^T g
#pragma unroll
    for (int k = 0; k < 9; ++k) B[k] = basis[i * 9 + k];
    const float g0 = B[0] * g[0] + B[3] * g[1] + B[6] * g[2];
    const float g1 = B[1] * g[0] + B[4] * g[1] + B[7] * g[2];
    const float g2 = B[2] * g[0] + B[5] * g[1] + B[8] * g[2];
    g[0] = g0; g[1] = g1; g[2] = g2;
  }
  if (TYPE == OPT_SCALAR) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float v = a.beta2 * exp_avg_sq[idx * D + j] + (1.0f - a.beta2) * g[j] * g[j];
      exp_avg_sq[idx * D + j] = v;
      dec[j] = opt_element(g[j], v, m[j], a, rs.y, rs.z);
    }
  } else {
    float ss = 0.0f;
#pragma unroll
    for (int j = 0; j < D; ++j) ss += g[j] * g[j];
    const float v = a.beta2 * exp_avg_sq[idx] + (1.0f - a.beta2) * (ss / (float)D);
    exp_avg_sq[idx] = v;
#pragma unroll
    for (int j = 0; j < D; ++j) dec[j] = opt_element(g[j], v, m[j], a, rs.y, rs.z);
  }
  if (TYPE == OPT_LOCAL_VECTOR) {             // back to world coordinates: B dec
    const float d0 = B[0] * dec[0] + B[1] * dec[1] + B[2] * dec[2];
    const float d1 = B[3] * dec[0] + B[4] * dec[1] + B[5] * dec[2];
    const float d2 = B[6] * dec[0] + B[7] * dec[1] + B[8] * dec[2];
    dec[0] = d0; dec[1] = d1; dec[2] = d2;
  }
#pragma unroll
  for (int j = 0; j < D; ++j) {
    exp_avg[idx * D + j] = m[j];
    param[idx * D + j] -= dec[j];
  }
}

// any D, 16 lanes per visible row (a row is read as one contiguous burst)
template <int TYPE>
__global__ __launch_bounds__(256) void opt_step_wide_kernel(float* __restrict__ param, const float* __restrict__ grad,
                                                            float* __restrict__ exp_avg, float* __restrict__ exp_avg_sq,
                                                            const int64_t* __restrict__ indexes,
                                                            const float4* __restrict__ row_scale, int64_t M, int D,
                                                            OptArgs a) {
  const int sub = threadIdx.x & 15;
  const int64_t i = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const bool live = i < M;                    // rows of a 16-lane group are uniform: the xor reduction stays in-group
  const int64_t idx = live ? indexes[i] : 0;
  const float4 rs = live ? row_scale[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t row = idx * D;
  float v_row = 0.0f;
  if (TYPE == OPT_VECTOR) {
    float ss = 0.0f;
    if (live)
      for (int j = sub; j < D; j += 16) {
        const float g = grad[row + j] * rs.x;
        ss += g * g;
      }
    ss += __shfl_xor(ss, 8, 16);
    ss += __shfl_xor(ss, 4, 16);
    ss += __shfl_xor(ss, 2, 16);
    ss += __shfl_xor(ss, 1, 16);
    if (live) {
      v_row = a.beta2 * exp_avg_sq[idx] + (1.0f - a.beta2) * (ss / (float)D);
    }
  }
  if (!live) return;
  for (int j = sub; j < D; j += 16) {
    const float g = grad[row + j] * rs.x;
    float m = exp_avg[row + j];
    float v = v_row;
    if (TYPE == OPT_SCALAR) {
      v = a.beta2 * exp_avg_sq[row + j] + (1.0f - a.beta2) * g * g;
      exp_avg_sq[row + j] = v;
    }
    const float dec = opt_element(g, v, m, a, rs.y, rs.z);
    exp_avg[row + j] = m;
    param[row + j] -= dec;
  }
  if (TYPE == OPT_VECTOR && sub == 0) exp_avg_sq[idx] = v_row;
}

// basis[m] = R(normalize(q[i])) * diag(max(exp(log_scaling[i]), eps)), i = indexes[m] (or m): the splat's own frame
// the local_vector groups step in (splat_trainer/gaussians/split.py:16-20, mlp_scene.py:219-225) -- one launch instead
// of the dozen elementwise torch kernels of the expression, which at 3 M rows cost more than the optimizer step itself.
__global__ __launch_bounds__(256) void point_basis_kernel(const float* __restrict__ log_scaling,
                                                          const float* __restrict__ rotation,
                                                          const int64_t* __restrict__ indexes, int64_t M, float eps,
                                                          float* __restrict__ basis) {
  const int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const int64_t i = indexes ? indexes[m] : m;
  const float4 qv = *reinterpret_cast<const float4*>(rotation + 4 * i);
  const float n = fmaxf(sqrtf(qv.x * qv.x + qv.y * qv.y + qv.z * qv.z + qv.w * qv.w), 1e-12f);   // F.normalize
  const float x = qv.x / n, y = qv.y / n, z = qv.z / n, w = qv.w / n;
  const float s0 = fmaxf(expf(log_scaling[3 * i]), eps), s1 = fmaxf(expf(log_scaling[3 * i + 1]), eps),
              s2 = fmaxf(expf(log_scaling[3 * i + 2]), eps);
  float* b = basis + 9 * m;
  b[0] = (1.f - 2.f * (y * y + z * z)) * s0; b[1] = (2.f * (x * y - w * z)) * s1; b[2] = (2.f * (x * z + w * y)) * s2;
  b[3] = (2.f * (x * y + w * z)) * s0; b[4] = (1.f - 2.f * (x * x + z * z)) * s1; b[5] = (2.f * (y * z - w * x)) * s2;
  b[6] = (2.f * (x * z - w * y)) * s0; b[7] = (2.f * (y * z + w * x)) * s1; b[8] = (1.f - 2.f * (x * x + y * y)) * s2;
}

}  // namespace

extern "C" {

// Per-point step counts / running visibility and the per-row factors of one optimizer step.  `visibility` may be NULL
// (plain sparse Adam / LaProp: weight 1, no normalisation).  `indexes` must not contain duplicates.
int gsr_opt_point_weights(const int64_t* indexes, const float* visibility, int64_t M, float* step, float* vis_avg,
                          float beta1, float beta2, float vis_beta, float vis_smooth, int32_t bias_correction,
                          float* row_scale, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!indexes || !step || !row_scale || (visibility && !vis_avg)) return GSR_ERR_INVALID_ARGUMENT;
  const unsigned blocks = (unsigned)((M + 255) / 256);
  opt_point_weights_kernel<<<blocks, 256, 0, stream>>>(indexes, visibility, M, step, vis_avg, beta1, beta2, vis_beta,
                                                       vis_smooth, bias_correction,
                                                       reinterpret_cast<float4*>(row_scale));
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

// One parameter group.  type: 0 scalar (second moment per element), 1 vector (one second moment per row: mean of
// squares), 2 local_vector (D = 3; gradient and update expressed in the splat's basis, `basis` (M,3,3) row-major).
// algo: 0 Adam, 1 LaProp.  exp_avg is (N,D); exp_avg_sq is (N,D) for scalar groups and (N,) otherwise.
int gsr_opt_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, const int64_t* indexes,
                 const float* row_scale, const float* basis, int64_t M, int32_t D, int32_t type, int32_t algo, float lr,
                 float beta1, float beta2, float eps, float grad_clip, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0 || D < 1 || type < 0 || type > 2 || algo < 0 || algo > 1) return GSR_ERR_INVALID_ARGUMENT;
  if (type == OPT_LOCAL_VECTOR && (D != 3 || !basis)) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!param || !grad || !exp_avg || !exp_avg_sq || !indexes || !row_scale) return GSR_ERR_INVALID_ARGUMENT;
  OptArgs a{lr, beta1, beta2, eps, grad_clip, algo};
  const float4* rs = reinterpret_cast<const float4*>(row_scale);
  const unsigned nb = (unsigned)((M + 255) / 256);
#define GSR_OPT_NARROW(D_, T_)                                                                                         \
  opt_step_narrow_kernel<D_, T_><<<nb, 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, indexes, rs, basis, M, a)
  if (type == OPT_LOCAL_VECTOR) {
    GSR_OPT_NARROW(3, OPT_LOCAL_VECTOR);
  } else if (D <= 4) {
    if (type == OPT_SCALAR) {
      switch (D) {
        case 1: GSR_OPT_NARROW(1, OPT_SCALAR); break;
        case 2: GSR_OPT_NARROW(2, OPT_SCALAR); break;
        case 3: GSR_OPT_NARROW(3, OPT_SCALAR); break;
        default: GSR_OPT_NARROW(4, OPT_SCALAR); break;
      }
    } else {
      switch (D) {
        case 1: GSR_OPT_NARROW(1, OPT_VECTOR); break;
        case 2: GSR_OPT_NARROW(2, OPT_VECTOR); break;
        case 3: GSR_OPT_NARROW(3, OPT_VECTOR); break;
        default: GSR_OPT_NARROW(4, OPT_VECTOR); break;
      }
    }
  } else {
    const unsigned wb = (unsigned)((M + 15) / 16);
    if (type == OPT_SCALAR)
      opt_step_wide_kernel<OPT_SCALAR><<<wb, 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, indexes, rs, M, D, a);
    else
      opt_step_wide_kernel<OPT_VECTOR><<<wb, 256, 0, stream>>>(param, grad, exp_avg, exp_avg_sq, indexes, rs, M, D, a);
  }
#undef GSR_OPT_NARROW
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

int gsr_point_basis(const float* log_scaling, const float* rotation_xyzw, const int64_t* indexes, int64_t M, float eps,
                    float* basis_out, void* stream_) {
  hipStream_t stream = reinterpret_cast<hipStream_t>(stream_);
  if (M < 0) return GSR_ERR_INVALID_ARGUMENT;
  if (M == 0) return GSR_OK;
  if (!log_scaling || !rotation_xyzw || !basis_out) return GSR_ERR_INVALID_ARGUMENT;
  point_basis_kernel<<<(unsigned)((M + 255) / 256), 256, 0, stream>>>(log_scaling, rotation_xyzw, indexes, M, eps, basis_out);
  GSR_CHECK_LAUNCH();
  return GSR_OK;
}

}  // extern "C"
