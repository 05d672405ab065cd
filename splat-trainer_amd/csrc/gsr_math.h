// Per-splat maths shared by the HIP kernels (device) and the CPU unit-test shim (host).
// Pure functions, no memory access, no wave intrinsics.  Spec: oracle/torch_oracle.py header.
#pragma once
#include <math.h>

#if defined(__HIPCC__)
#define GSR_HD __host__ __device__ __forceinline__
#else
#define GSR_HD inline
#endif

struct GsrRasterParams {
  float alpha_threshold;   // 1/255
  float clamp_max_alpha;   // 0.99
  float T_eps;             // 1 - saturate_threshold
  float q_max;             // gaussian_scale^2
  float blur;              // blur_cov (+ aa_blur when antialias)
  int   antialias;         // 0/1
  int   tile_size;         // 16 (kernels are specialised for 16)
  float margin_px;         // margin_tiles * tile_size
};

struct GsrCam {            // loaded from the device-resident T_camera_world (4x4 row major) + projection
  float R[9];
  float t[3];
  float fx, fy, cx, cy;
};

GSR_HD GsrCam gsr_load_cam(const float* T, const float* proj) {
  GsrCam c;
  c.R[0] = T[0]; c.R[1] = T[1]; c.R[2] = T[2];  c.t[0] = T[3];
  c.R[3] = T[4]; c.R[4] = T[5]; c.R[5] = T[6];  c.t[1] = T[7];
  c.R[6] = T[8]; c.R[7] = T[9]; c.R[8] = T[10]; c.t[2] = T[11];
  c.fx = proj[0]; c.fy = proj[1]; c.cx = proj[2]; c.cy = proj[3];
  return c;
}

// Rounding is PINNED in the forward projection (gsr_to_camera, gsr_quat_to_rot, gsr_project_one): every fused
// multiply-add is written out and nothing else may be contracted.  The projection is inlined into several kernels (K1,
// stand-alone K2, fused K2 + K3 with and without the saved Jacobian) that must return the same bits -- a frame's image,
// and with it the controller scores and densification masks, must not depend on which call form or mode rendered it.
GSR_HD void gsr_to_camera(const GsrCam& c, float px, float py, float pz, float& x, float& y, float& z) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  x = fmaf(c.R[0], px, fmaf(c.R[1], py, fmaf(c.R[2], pz, c.t[0])));
  y = fmaf(c.R[3], px, fmaf(c.R[4], py, fmaf(c.R[5], pz, c.t[1])));
  // The depth is formed in double and rounded once: it is the SORT KEY of the frame, and at millions of splats
  // neighbouring depths are an fp32 ulp apart -- a depth that is off by one ulp orders nearly coplanar splats differently
  // from the (fp64) specification.  Products of two floats are exact in double, so this is the correctly rounded value
  // of R p + t (up to double rounding, probability ~1e-9 per splat) and the depth ORDER can be checked against a stable
  // argsort of the oracle's depths (tests/test_gpu_render.py, config 3).  Three DFMAs per splat in HBM-bound kernels.
  z = (float)fma((double)c.R[6], (double)px, fma((double)c.R[7], (double)py, fma((double)c.R[8], (double)pz, (double)c.t[2])));
}

// centre-in-frustum test (K1).  The expanded image bound keeps x/z, y/z bounded for K2.
GSR_HD bool gsr_in_view(const GsrCam& c, float px, float py, float pz, int W, int H,
                        float near_p, float far_p, float margin) {
  float x, y, z;
  gsr_to_camera(c, px, py, pz, x, y, z);
  if (!(z > near_p && z < far_p)) return false;
  float u = c.fx * x / z + c.cx;
  float v = c.fy * y / z + c.cy;
  return (u > -margin) && (u < (float)W + margin) && (v > -margin) && (v < (float)H + margin);
}

GSR_HD void gsr_quat_to_rot(const float qn[4], float Rq[9]) {   // xyzw, already normalised
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  const float x = qn[0], y = qn[1], z = qn[2], w = qn[3];
  Rq[0] = fmaf(-2.f, fmaf(y, y, z * z), 1.f); Rq[1] = 2.f * fmaf(x, y, -(w * z));       Rq[2] = 2.f * fmaf(x, z, w * y);
  Rq[3] = 2.f * fmaf(x, y, w * z);            Rq[4] = fmaf(-2.f, fmaf(x, x, z * z), 1.f); Rq[5] = 2.f * fmaf(y, z, -(w * x));
  Rq[6] = 2.f * fmaf(x, z, -(w * y));         Rq[7] = 2.f * fmaf(y, z, w * x);           Rq[8] = fmaf(-2.f, fmaf(x, x, y * y), 1.f);
}

struct GsrProjected {
  float u, v, A, B, C, opacity, depth, s_major, s_minor;
};

// K2 forward for one splat (rounding pinned, see gsr_to_camera).
GSR_HD GsrProjected gsr_project_one(const GsrCam& c, const GsrRasterParams& rp, const float p[3],
                                    const float ls[3], const float q[4], float logit) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  GsrProjected o;
  float x, y, z;
  gsr_to_camera(c, p[0], p[1], p[2], x, y, z);
  const float iz = 1.f / z;
  o.u = fmaf(c.fx * x, iz, c.cx);
  o.v = fmaf(c.fy * y, iz, c.cy);
  o.depth = z;

  float qn[4];
  const float inv = 1.f / sqrtf(fmaf(q[0], q[0], fmaf(q[1], q[1], fmaf(q[2], q[2], q[3] * q[3]))));
  for (int i = 0; i < 4; ++i) qn[i] = q[i] * inv;
  float Rq[9];
  gsr_quat_to_rot(qn, Rq);
  const float s[3] = {expf(ls[0]), expf(ls[1]), expf(ls[2])};

  // Wm = J * R_cw  (2x3)
  const float j00 = c.fx * iz, j02 = -c.fx * x * iz * iz, j11 = c.fy * iz, j12 = -c.fy * y * iz * iz;
  float Wm[6];
  for (int k = 0; k < 3; ++k) {
    Wm[k]     = fmaf(j00, c.R[k], j02 * c.R[6 + k]);
    Wm[3 + k] = fmaf(j11, c.R[3 + k], j12 * c.R[6 + k]);
  }
  // Tm = Wm * (Rq diag(s))  (2x3)
  float Tm[6];
  for (int cc = 0; cc < 3; ++cc) {
    const float m0 = Rq[cc] * s[cc], m1 = Rq[3 + cc] * s[cc], m2 = Rq[6 + cc] * s[cc];
    Tm[cc]     = fmaf(Wm[0], m0, fmaf(Wm[1], m1, Wm[2] * m2));
    Tm[3 + cc] = fmaf(Wm[3], m0, fmaf(Wm[4], m1, Wm[5] * m2));
  }
  const float a0 = fmaf(Tm[0], Tm[0], fmaf(Tm[1], Tm[1], Tm[2] * Tm[2]));
  const float b0 = fmaf(Tm[0], Tm[3], fmaf(Tm[1], Tm[4], Tm[2] * Tm[5]));
  const float c0 = fmaf(Tm[3], Tm[3], fmaf(Tm[4], Tm[4], Tm[5] * Tm[5]));
  const float a = a0 + rp.blur, b = b0, cc2 = c0 + rp.blur;
  const float det = fmaf(a, cc2, -(b * b));
  const float idet = 1.f / det;
  o.A = cc2 * idet; o.B = -b * idet; o.C = a * idet;
  float op = 1.f / (1.f + expf(-logit));
  if (rp.antialias) {
    const float rho = fmaf(a0, c0, -(b0 * b0)) * idet;
    op *= sqrtf(fmaxf(rho, 0.f));
  }
  o.opacity = op;
  const float mid = 0.5f * (a + cc2);
  const float rad = sqrtf(fmaxf(fmaf(mid, mid, -det), 0.f));
  o.s_major = sqrtf(mid + rad);
  o.s_minor = sqrtf(fmaxf(mid - rad, 0.f));
  return o;
}

struct GsrProjectGrad {
  float dp[3], dls[3], dq[4], dlogit;
};

// K2 backward for one splat: g = dL/d[u v A B C opacity], gdepth = dL/ddepth.
GSR_HD GsrProjectGrad gsr_project_one_bwd(const GsrCam& c, const GsrRasterParams& rp, const float p[3],
                                          const float ls[3], const float q[4], float logit,
                                          const float g[6], float gdepth) {
  GsrProjectGrad o;
  float x, y, z;
  gsr_to_camera(c, p[0], p[1], p[2], x, y, z);
  float iz = 1.f / z, iz2 = iz * iz;

  float qlen2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
  float inv = 1.f / sqrtf(qlen2);
  float qn[4];
  for (int i = 0; i < 4; ++i) qn[i] = q[i] * inv;
  float Rq[9];
  gsr_quat_to_rot(qn, Rq);
  float s[3] = {expf(ls[0]), expf(ls[1]), expf(ls[2])};

  float j00 = c.fx * iz, j02 = -c.fx * x * iz2, j11 = c.fy * iz, j12 = -c.fy * y * iz2;
  float Wm[6];
  for (int k = 0; k < 3; ++k) {
    Wm[k]     = j00 * c.R[k]     + j02 * c.R[6 + k];
    Wm[3 + k] = j11 * c.R[3 + k] + j12 * c.R[6 + k];
  }
  float Mw[9];
  for (int r = 0; r < 3; ++r)
    for (int cc = 0; cc < 3; ++cc) Mw[3 * r + cc] = Rq[3 * r + cc] * s[cc];
  float Tm[6];
  for (int cc = 0; cc < 3; ++cc) {
    Tm[cc]     = Wm[0] * Mw[cc] + Wm[1] * Mw[3 + cc] + Wm[2] * Mw[6 + cc];
    Tm[3 + cc] = Wm[3] * Mw[cc] + Wm[4] * Mw[3 + cc] + Wm[5] * Mw[6 + cc];
  }
  float a0 = Tm[0] * Tm[0] + Tm[1] * Tm[1] + Tm[2] * Tm[2];
  float b0 = Tm[0] * Tm[3] + Tm[1] * Tm[4] + Tm[2] * Tm[5];
  float c0 = Tm[3] * Tm[3] + Tm[4] * Tm[4] + Tm[5] * Tm[5];
  float a = a0 + rp.blur, b = b0, cc2 = c0 + rp.blur;
  float det = a * cc2 - b * b;
  float idet = 1.f / det, idet2 = idet * idet;

  // conic (A,B,C) = (c, -b, a)/det  ->  (a, b, c)
  float gA = g[2], gB = g[3], gC = g[4];
  float ga = gA * (-cc2 * cc2 * idet2) + gB * (b * cc2 * idet2)            + gC * (-b * b * idet2);
  float gb = gA * (2.f * b * cc2 * idet2) + gB * (-(a * cc2 + b * b) * idet2) + gC * (2.f * a * b * idet2);
  float gc = gA * (-b * b * idet2)     + gB * (a * b * idet2)             + gC * (-a * a * idet2);

  float sig = 1.f / (1.f + expf(-logit));
  float gsig = g[5];
  if (rp.antialias) {
    float det0 = a0 * c0 - b0 * b0;
    float rho = det0 * idet;
    if (rho > 0.f) {
      float r = sqrtf(rho);
      float g_rho = g[5] * sig / (2.f * r);
      float g_det0 = g_rho * idet;
      float g_det = -g_rho * det0 * idet2;
      ga += g_det0 * c0 + g_det * cc2;
      gb += -2.f * b0 * g_det0 - 2.f * b * g_det;
      gc += g_det0 * a0 + g_det * a;
      gsig = g[5] * r;
    } else {
      gsig = 0.f;
    }
  }
  o.dlogit = gsig * sig * (1.f - sig);

  // cov -> Tm
  float gT[6];
  for (int cc = 0; cc < 3; ++cc) {
    gT[cc]     = 2.f * ga * Tm[cc] + gb * Tm[3 + cc];
    gT[3 + cc] = 2.f * gc * Tm[3 + cc] + gb * Tm[cc];
  }
  // Tm = Wm Mw
  float gW[6], gM[9];
  for (int k = 0; k < 3; ++k) {
    gW[k]     = gT[0] * Mw[3 * k] + gT[1] * Mw[3 * k + 1] + gT[2] * Mw[3 * k + 2];
    gW[3 + k] = gT[3] * Mw[3 * k] + gT[4] * Mw[3 * k + 1] + gT[5] * Mw[3 * k + 2];
    for (int cc = 0; cc < 3; ++cc) gM[3 * k + cc] = Wm[k] * gT[cc] + Wm[3 + k] * gT[3 + cc];
  }
  // Mw = Rq diag(s)
  float gR[9];
  for (int cc = 0; cc < 3; ++cc) {
    float gs = gM[cc] * Rq[cc] + gM[3 + cc] * Rq[3 + cc] + gM[6 + cc] * Rq[6 + cc];
    o.dls[cc] = gs * s[cc];
    gR[cc] = gM[cc] * s[cc]; gR[3 + cc] = gM[3 + cc] * s[cc]; gR[6 + cc] = gM[6 + cc] * s[cc];
  }
  // rotmat -> unit quaternion (xyzw)
  {
    float qx = qn[0], qy = qn[1], qz = qn[2], qw = qn[3];
    float gx = 2.f * (qy * (gR[1] + gR[3]) + qz * (gR[2] + gR[6]) - 2.f * qx * (gR[4] + gR[8]) + qw * (gR[7] - gR[5]));
    float gy = 2.f * (qx * (gR[1] + gR[3]) + qz * (gR[5] + gR[7]) - 2.f * qy * (gR[0] + gR[8]) + qw * (gR[2] - gR[6]));
    float gz = 2.f * (qx * (gR[2] + gR[6]) + qy * (gR[5] + gR[7]) - 2.f * qz * (gR[0] + gR[4]) + qw * (gR[3] - gR[1]));
    float gw = 2.f * (qz * (gR[3] - gR[1]) + qy * (gR[2] - gR[6]) + qx * (gR[7] - gR[5]));
    float dot = qx * gx + qy * gy + qz * gz + qw * gw;    // through q / |q|
    o.dq[0] = (gx - qx * dot) * inv;
    o.dq[1] = (gy - qy * dot) * inv;
    o.dq[2] = (gz - qz * dot) * inv;
    o.dq[3] = (gw - qw * dot) * inv;
  }
  // Wm = J R_cw -> J
  float gJ00 = gW[0] * c.R[0] + gW[1] * c.R[1] + gW[2] * c.R[2];
  float gJ02 = gW[0] * c.R[6] + gW[1] * c.R[7] + gW[2] * c.R[8];
  float gJ11 = gW[3] * c.R[3] + gW[4] * c.R[4] + gW[5] * c.R[5];
  float gJ12 = gW[3] * c.R[6] + gW[4] * c.R[7] + gW[5] * c.R[8];
  float iz3 = iz2 * iz;
  float gxc = gJ02 * (-c.fx * iz2) + g[0] * c.fx * iz;
  float gyc = gJ12 * (-c.fy * iz2) + g[1] * c.fy * iz;
  float gzc = gJ00 * (-c.fx * iz2) + gJ02 * (2.f * c.fx * x * iz3)
            + gJ11 * (-c.fy * iz2) + gJ12 * (2.f * c.fy * y * iz3)
            - g[0] * c.fx * x * iz2 - g[1] * c.fy * y * iz2 + gdepth;
  o.dp[0] = c.R[0] * gxc + c.R[3] * gyc + c.R[6] * gzc;
  o.dp[1] = c.R[1] * gxc + c.R[4] * gyc + c.R[7] * gzc;
  o.dp[2] = c.R[2] * gxc + c.R[5] * gyc + c.R[8] * gzc;
  return o;
}

// dL/dposition through the view direction from the colour gradient g and the saved d colour / d position (row-major 3x3).
GSR_HD void gsr_jac_apply(const float g[3], const float* J, float out[3]) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  out[0] = g[0] * J[0] + g[1] * J[3] + g[2] * J[6];
  out[1] = g[0] * J[1] + g[1] * J[4] + g[2] * J[7];
  out[2] = g[0] * J[2] + g[1] * J[5] + g[2] * J[8];
}

// ------------------------------------------------------------------ SH basis, degrees 0..3
#define GSR_SH_C0 0.28209479177387814f
#define GSR_SH_C1 0.4886025119029199f

// Every product and sum below is rounded on its own (no fma contraction): the forward kernel exists in several
// instantiations (with / without the Jacobian the backward pass saves) that must return bit-identical colours -- a
// data-parallel run evaluates the colours without it, a single-GPU run with it, and both must see the same image.
template <int K>
GSR_HD void gsr_sh_basis(float x, float y, float z, float* Y) {
#if defined(__clang__)
#pragma clang fp contract(off)
#endif
  Y[0] = GSR_SH_C0;
  if (K > 1) {
    Y[1] = -GSR_SH_C1 * y; Y[2] = GSR_SH_C1 * z; Y[3] = -GSR_SH_C1 * x;
  }
  if (K > 4) {
    float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    Y[4] = 1.0925484305920792f * xy;
    Y[5] = -1.0925484305920792f * yz;
    Y[6] = 0.31539156525252005f * (2.f * zz - xx - yy);
    Y[7] = -1.0925484305920792f * xz;
    Y[8] = 0.5462742152960396f * (xx - yy);
    if (K > 9) {
      Y[9]  = -0.5900435899266435f * y * (3.f * xx - yy);
      Y[10] = 2.890611442640554f * xy * z;
      Y[11] = -0.4570457994644658f * y * (4.f * zz - xx - yy);
      Y[12] = 0.3731763325901154f * z * (2.f * zz - 3.f * xx - 3.f * yy);
      Y[13] = -0.4570457994644658f * x * (4.f * zz - xx - yy);
      Y[14] = 1.445305721320277f * z * (xx - yy);
      Y[15] = -0.5900435899266435f * x * (xx - 3.f * yy);
    }
  }
}

// Polynomial partial derivatives of the basis wrt the (unit) direction components; the caller chains
// them through the normalisation d = v/|v|.
template <int K>
GSR_HD void gsr_sh_basis_grad(float x, float y, float z, float* dx, float* dy, float* dz) {
  dx[0] = dy[0] = dz[0] = 0.f;
  if (K > 1) {
    dx[1] = 0.f;         dy[1] = -GSR_SH_C1; dz[1] = 0.f;
    dx[2] = 0.f;         dy[2] = 0.f;        dz[2] = GSR_SH_C1;
    dx[3] = -GSR_SH_C1;  dy[3] = 0.f;        dz[3] = 0.f;
  }
  if (K > 4) {
    const float c0 = 1.0925484305920792f, c1 = -1.0925484305920792f, c2 = 0.31539156525252005f,
                c3 = -1.0925484305920792f, c4 = 0.5462742152960396f;
    dx[4] = c0 * y;         dy[4] = c0 * x;         dz[4] = 0.f;
    dx[5] = 0.f;            dy[5] = c1 * z;         dz[5] = c1 * y;
    dx[6] = -2.f * c2 * x;  dy[6] = -2.f * c2 * y;  dz[6] = 4.f * c2 * z;
    dx[7] = c3 * z;         dy[7] = 0.f;            dz[7] = c3 * x;
    dx[8] = 2.f * c4 * x;   dy[8] = -2.f * c4 * y;  dz[8] = 0.f;
    if (K > 9) {
      const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
      const float e0 = -0.5900435899266435f, e1 = 2.890611442640554f, e2 = -0.4570457994644658f,
                  e3 = 0.3731763325901154f, e4 = -0.4570457994644658f, e5 = 1.445305721320277f,
                  e6 = -0.5900435899266435f;
      dx[9]  = 6.f * e0 * xy;                dy[9]  = e0 * (3.f * xx - 3.f * yy);       dz[9]  = 0.f;
      dx[10] = e1 * yz;                      dy[10] = e1 * xz;                          dz[10] = e1 * xy;
      dx[11] = -2.f * e2 * xy;               dy[11] = e2 * (4.f * zz - xx - 3.f * yy);  dz[11] = 8.f * e2 * yz;
      dx[12] = -6.f * e3 * xz;               dy[12] = -6.f * e3 * yz;                   dz[12] = e3 * (6.f * zz - 3.f * xx - 3.f * yy);
      dx[13] = e4 * (4.f * zz - 3.f * xx - yy); dy[13] = -2.f * e4 * xy;                dz[13] = 8.f * e4 * xz;
      dx[14] = 2.f * e5 * xz;                dy[14] = -2.f * e5 * yz;                   dz[14] = e5 * (xx - yy);
      dx[15] = e6 * (3.f * xx - 3.f * yy);   dy[15] = -6.f * e6 * xy;                   dz[15] = 0.f;
    }
  }
}

// The same partial derivatives, one basis function at a time (k is a compile-time constant after unrolling, the switch
// folds away): lets a caller consume dY_k right where it forms it instead of holding all 3K values in registers.
template <int K>
GSR_HD void gsr_sh_basis_grad_at(int k, float x, float y, float z, float& ax, float& ay, float& az) {
  const float c0 = 1.0925484305920792f, c1 = -1.0925484305920792f, c2 = 0.31539156525252005f,
              c3 = -1.0925484305920792f, c4 = 0.5462742152960396f;
  const float e0 = -0.5900435899266435f, e1 = 2.890611442640554f, e2 = -0.4570457994644658f,
              e3 = 0.3731763325901154f, e4 = -0.4570457994644658f, e5 = 1.445305721320277f,
              e6 = -0.5900435899266435f;
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  switch (k) {
    case 1: ax = 0.f; ay = -GSR_SH_C1; az = 0.f; break;
    case 2: ax = 0.f; ay = 0.f; az = GSR_SH_C1; break;
    case 3: ax = -GSR_SH_C1; ay = 0.f; az = 0.f; break;
    case 4: ax = c0 * y; ay = c0 * x; az = 0.f; break;
    case 5: ax = 0.f; ay = c1 * z; az = c1 * y; break;
    case 6: ax = -2.f * c2 * x; ay = -2.f * c2 * y; az = 4.f * c2 * z; break;
    case 7: ax = c3 * z; ay = 0.f; az = c3 * x; break;
    case 8: ax = 2.f * c4 * x; ay = -2.f * c4 * y; az = 0.f; break;
    case 9: ax = 6.f * e0 * xy; ay = e0 * (3.f * xx - 3.f * yy); az = 0.f; break;
    case 10: ax = e1 * yz; ay = e1 * xz; az = e1 * xy; break;
    case 11: ax = -2.f * e2 * xy; ay = e2 * (4.f * zz - xx - 3.f * yy); az = 8.f * e2 * yz; break;
    case 12: ax = -6.f * e3 * xz; ay = -6.f * e3 * yz; az = e3 * (6.f * zz - 3.f * xx - 3.f * yy); break;
    case 13: ax = e4 * (4.f * zz - 3.f * xx - yy); ay = -2.f * e4 * xy; az = 8.f * e4 * xz; break;
    case 14: ax = 2.f * e5 * xz; ay = -2.f * e5 * yz; az = e5 * (xx - yy); break;
    case 15: ax = e6 * (3.f * xx - 3.f * yy); ay = -6.f * e6 * xy; az = 0.f; break;
    default: ax = ay = az = 0.f; break;
  }
}

// ------------------------------------------------------------------ tile binning maths (K4)
struct GsrExtent {            // half-open tile rectangle [x0,x1) x [y0,y1) and the effective support
  int x0, x1, y0, y1;
  float qmax;                 // d^T conic d <= qmax can reach alpha >= alpha_threshold
};

// Conservative support of a projected splat: q <= min(q_max, 2 ln(opacity / alpha_threshold)) (inflated a
// hair so that fp32 rounding in the per-pixel test can never include a pixel the binning excluded).
GSR_HD GsrExtent gsr_splat_extent(float u, float v, float A, float B, float C, float op,
                                  const GsrRasterParams& rp, int tiles_x, int tiles_y) {
  GsrExtent e;
  e.x0 = e.x1 = e.y0 = e.y1 = 0;
  e.qmax = 0.f;
  if (!(op >= rp.alpha_threshold)) return e;
  float qop = 2.f * logf(op / rp.alpha_threshold) * 1.0001f + 1e-4f;
  float qmax = fminf(rp.q_max * 1.00001f, qop);
  float det = A * C - B * B;
  if (!(det > 0.f)) return e;
  float hx = sqrtf(qmax * C / det) + 1e-3f;     // cov_xx = C/det, cov_yy = A/det
  float hy = sqrtf(qmax * A / det) + 1e-3f;
  const float ts = 16.f;
  // pixel centres of tile column k span [16k + .5, 16k + 15.5]
  float fx0 = floorf((u - hx - 0.5f) / ts), fx1 = floorf((u + hx - 0.5f) / ts) + 1.f;
  float fy0 = floorf((v - hy - 0.5f) / ts), fy1 = floorf((v + hy - 0.5f) / ts) + 1.f;
  e.x0 = (int)fminf(fmaxf(fx0, 0.f), (float)tiles_x);
  e.x1 = (int)fminf(fmaxf(fx1, 0.f), (float)tiles_x);
  e.y0 = (int)fminf(fmaxf(fy0, 0.f), (float)tiles_y);
  e.y1 = (int)fminf(fmaxf(fy1, 0.f), (float)tiles_y);
  e.qmax = qmax;
  return e;
}

// min over a segment of the quadratic q restricted to a line: fixed offset d_fixed on one axis,
// free offset in [lo, hi] on the other.  q = Kf d_fixed^2 + 2 B d_fixed d + Kv d^2.
GSR_HD float gsr_edge_min_q(float Kf, float Bc, float Kv, float d_fixed, float lo, float hi) {
  float d = -Bc * d_fixed / Kv;
  d = fminf(fmaxf(d, lo), hi);
  return Kf * d_fixed * d_fixed + 2.f * Bc * d_fixed * d + Kv * d * d;
}

// Does the ellipse {q <= qmax} reach the rectangle [rx0,rx1] x [ry0,ry1] (given relative to the mean)?
GSR_HD bool gsr_rect_hit(float A, float B, float C, float qmax, float rx0, float rx1, float ry0, float ry1) {
  if (rx0 <= 0.f && rx1 >= 0.f && ry0 <= 0.f && ry1 >= 0.f) return true;
  float m = gsr_edge_min_q(A, B, C, rx0, ry0, ry1);
  m = fminf(m, gsr_edge_min_q(A, B, C, rx1, ry0, ry1));
  m = fminf(m, gsr_edge_min_q(C, B, A, ry0, rx0, rx1));
  m = fminf(m, gsr_edge_min_q(C, B, A, ry1, rx0, rx1));
  return m <= qmax;
}

// ... the rectangle of pixel centres of tile (tx, ty)?
GSR_HD bool gsr_tile_hit(float u, float v, float A, float B, float C, float qmax, int tx, int ty) {
  return gsr_rect_hit(A, B, C, qmax, 16.f * tx + 0.5f - u, 16.f * tx + 15.5f - u, 16.f * ty + 0.5f - v,
                      16.f * ty + 15.5f - v);
}

// ... the pixel centres of the upper (bit 0: rows 0-7) / lower (bit 1: rows 8-15) half of tile (tx, ty)?
// The composite kernels evaluate a tile half as one unit and skip a half whose bit is clear.
GSR_HD unsigned gsr_tile_half_mask(float u, float v, float A, float B, float C, float qmax, int tx, int ty) {
  const float rx0 = 16.f * tx + 0.5f - u, rx1 = 16.f * tx + 15.5f - u;
  const float ry = 16.f * ty + 0.5f - v;
  unsigned m = gsr_rect_hit(A, B, C, qmax, rx0, rx1, ry, ry + 7.f) ? 1u : 0u;
  m |= gsr_rect_hit(A, B, C, qmax, rx0, rx1, ry + 8.f, ry + 15.f) ? 2u : 0u;
  return m;
}
