// CPU unit-test shim: exposes the per-splat maths of gsr_math.h (the very source the HIP kernels compile)
// to the `-m "not gpu"` tests.  It is NOT a product path: nothing in the package calls it, it renders
// nothing, and the rasterizer fails loudly without libgsplat_hip.so.
#include <stdint.h>
#include <string.h>

#include "gsr_math.h"

extern "C" {

// params: 8 x 4 bytes laid out as GsrRasterParams
void hm_project_forward(const float* T, const float* proj, const void* params, int64_t M, const float* pos,
                        const float* ls, const float* rot, const float* logit, float* g2d, float* depth,
                        float* sscale) {
  GsrRasterParams rp;
  memcpy(&rp, params, sizeof(rp));
  GsrCam cam = gsr_load_cam(T, proj);
  for (int64_t m = 0; m < M; ++m) {
    GsrProjected o = gsr_project_one(cam, rp, pos + 3 * m, ls + 3 * m, rot + 4 * m, logit[m]);
    float* g = g2d + 6 * m;
    g[0] = o.u; g[1] = o.v; g[2] = o.A; g[3] = o.B; g[4] = o.C; g[5] = o.opacity;
    depth[m] = o.depth;
    sscale[2 * m] = o.s_major; sscale[2 * m + 1] = o.s_minor;
  }
}

void hm_project_backward(const float* T, const float* proj, const void* params, int64_t M, const float* pos,
                         const float* ls, const float* rot, const float* logit, const float* dg2d,
                         const float* ddepth, float* dpos, float* dls, float* drot, float* dlogit) {
  GsrRasterParams rp;
  memcpy(&rp, params, sizeof(rp));
  GsrCam cam = gsr_load_cam(T, proj);
  for (int64_t m = 0; m < M; ++m) {
    GsrProjectGrad o = gsr_project_one_bwd(cam, rp, pos + 3 * m, ls + 3 * m, rot + 4 * m, logit[m], dg2d + 6 * m,
                                           ddepth[m]);
    for (int k = 0; k < 3; ++k) { dpos[3 * m + k] = o.dp[k]; dls[3 * m + k] = o.dls[k]; }
    for (int k = 0; k < 4; ++k) drot[4 * m + k] = o.dq[k];
    dlogit[m] = o.dlogit;
  }
}

void hm_in_view(const float* T, const float* proj, int64_t N, const float* pos, int W, int H, float near_p,
                float far_p, float margin, uint8_t* mask) {
  GsrCam cam = gsr_load_cam(T, proj);
  for (int64_t i = 0; i < N; ++i)
    mask[i] = gsr_in_view(cam, pos[3 * i], pos[3 * i + 1], pos[3 * i + 2], W, H, near_p, far_p, margin) ? 1 : 0;
}

void hm_sh_basis(int K, int64_t M, const float* dirs, float* Y) {
  for (int64_t m = 0; m < M; ++m) {
    const float* d = dirs + 3 * m;
    float* y = Y + (int64_t)K * m;
    switch (K) {
      case 1: gsr_sh_basis<1>(d[0], d[1], d[2], y); break;
      case 4: gsr_sh_basis<4>(d[0], d[1], d[2], y); break;
      case 9: gsr_sh_basis<9>(d[0], d[1], d[2], y); break;
      default: gsr_sh_basis<16>(d[0], d[1], d[2], y); break;
    }
  }
}

// per-splat tile hit mask over the whole tile grid: hits[m * tiles_x * tiles_y + ty * tiles_x + tx]
void hm_tile_hits(const void* params, int64_t M, const float* g2d, int tiles_x, int tiles_y, uint8_t* hits) {
  GsrRasterParams rp;
  memcpy(&rp, params, sizeof(rp));
  const int64_t nt = (int64_t)tiles_x * tiles_y;
  for (int64_t m = 0; m < M; ++m) {
    const float* g = g2d + 6 * m;
    GsrExtent e = gsr_splat_extent(g[0], g[1], g[2], g[3], g[4], g[5], rp, tiles_x, tiles_y);
    for (int ty = e.y0; ty < e.y1; ++ty)
      for (int tx = e.x0; tx < e.x1; ++tx)
        if (gsr_tile_hit(g[0], g[1], g[2], g[3], g[4], e.qmax, tx, ty)) hits[m * nt + ty * tiles_x + tx] = 1;
  }
}

}  // extern "C"
