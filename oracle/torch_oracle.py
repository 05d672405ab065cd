"""CPU oracle for the Gaussian-splat rasterizer path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module; the product path (``splat-trainer_amd/``) never does and fails loudly when the HIP
library is missing.

PARITY UNPINNED.  The arithmetic of this path lives in ``taichi-splatting >= 0.31.0``
(/root/reference/pyproject.toml:16), which is neither vendored in the reference tree nor
installed, and the reference holds no golden images/grads for it.  This file is
therefore a restatement of the *published* 3D-Gaussian-splatting algorithm (Kerbl et al. 2023:
EWA projection, front-to-back alpha compositing with alpha clamp / threshold / transmittance
stop) anchored on the reference's own call sites:

  * boundary + argument meaning ........ splat_trainer/scene/mlp_scene.py:372-427
  * camera convention (OpenCV pinhole) .. splat_trainer/trainer/trainer.py:291-301,
                                          splat_trainer/visibility/query_points.py:27-31,73-84
  * Gaussian parametrisation ............ splat_trainer/gaussians/split.py:16-20
                                          (basis = R(q) diag(exp(log_s)); Sigma = basis basis^T),
                                          quaternion xyzw: splat_trainer/scene/io.py:102-104
  * opacity = sigmoid(alpha_logit) ...... splat_trainer/scene/mlp_scene.py:193
  * raster options ...................... splat_trainer/trainer/trainer.py:305-310
  * consumers of the per-point outputs .. splat_trainer/controller/point_state.py:34-57
  * SH basis (pinned by golden vectors) . splat_trainer/scene/mlp/rsh.py:11-158
  * SH -> rgb offset 0.5 ................ splat_trainer/util/misc.py:41-49,
                                          splat_trainer/scene/transfer_sh.py:30-31

What *is* pinned by reference code: the SH basis (tests/golden/rsh_deg0_4.npz, generated from
rsh.py) and the controller maths consuming the per-point outputs (tests/golden/misc_vectors.json,
generated from util/misc.py).

Maths spec (the HIP kernels implement exactly this; written here with differentiable torch ops so
autograd provides the gradient oracle, any float dtype, fp64 for gradcheck):

  cull      in view  <=>  near < z < far  and  -m < u < W+m  and  -m < v < H+m,   m = margin_tiles*tile_size
  project   p_c = R_cw p + t;  (u,v) = (fx x/z + cx, fy y/z + cy);  depth = z
            Sigma = R(q) diag(exp(2 log_s)) R(q)^T,  q = xyzw normalised
            J = [[fx/z, 0, -fx x/z^2], [0, fy/z, -fy y/z^2]];  cov = J R_cw Sigma R_cw^T J^T
            antialias: cov_b = cov + (aa_blur + blur_cov) I, opacity *= sqrt(max(det cov / det cov_b, 0))
            else      : cov_b = cov + blur_cov I
            conic = cov_b^-1 = (A, B, C);  opacity = sigmoid(alpha_logit) [* aa factor]
            screen_scale = sqrt(eigenvalues(cov_b)) (major, minor)
            gaussians2d row = [u, v, A, B, C, opacity]
  raster    splats sorted by (depth, index) ascending.  For pixel centre x = (i+.5, j+.5), d = x - (u,v):
            q = A dx^2 + 2 B dx dy + C dy^2;  skip unless q <= gaussian_scale^2
            alpha = min(clamp_max_alpha, opacity * exp(-q/2));  skip unless alpha >= alpha_threshold
            C += T alpha f;  T *= (1 - alpha);  stop the pixel once T < 1 - saturate_threshold
            (support is defined per pixel, so the result does not depend on how tiles are binned)
  per point visibility = sum_px T alpha
            prune_cost = sum_px |dL_px/dalpha| alpha        (first-order loss change if the point is removed)
            split_score = sum_px || dL_px/d(u,v) ||_2       (per-pixel, so opposite signs do not cancel)
            with d alpha / d(opacity G) = 0 where the clamp is active (what autograd does).
  sh        colour_c = 0.5 + sum_k sh[c,k] Y_k(normalize(p - cam_pos)),  Y_k ordering k = n(n+1)+m (rsh.py)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


# ----------------------------------------------------------------------------- SH basis
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


def sh_basis(dirs: torch.Tensor, degree: int) -> torch.Tensor:
  """Real SH basis of unit vectors, (..., (degree+1)^2), index n(n+1)+m.

  Closed forms of the 3DGS paper's SH evaluation; checked against the reference's
  splat_trainer/scene/mlp/rsh.py (rsh_cart_0..3) through tests/golden/rsh_deg0_4.npz."""
  assert 0 <= degree <= 3, "oracle restates degrees 0..3"
  x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
  out = [torch.full_like(x, SH_C0)]
  if degree >= 1:
    out += [-SH_C1 * y, SH_C1 * z, -SH_C1 * x]
  if degree >= 2:
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    out += [SH_C2[0] * xy, SH_C2[1] * yz, SH_C2[2] * (2.0 * zz - xx - yy),
            SH_C2[3] * xz, SH_C2[4] * (xx - yy)]
  if degree >= 3:
    out += [SH_C3[0] * y * (3.0 * xx - yy), SH_C3[1] * xy * z,
            SH_C3[2] * y * (4.0 * zz - xx - yy),
            SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy),
            SH_C3[4] * x * (4.0 * zz - xx - yy), SH_C3[5] * z * (xx - yy),
            SH_C3[6] * x * (xx - 3.0 * yy)]
  return torch.stack(out, dim=-1)


def evaluate_sh_at(sh_features: torch.Tensor, positions: torch.Tensor, indexes: torch.Tensor,
                   camera_pos: torch.Tensor) -> torch.Tensor:
  """(N,3,K), (N,3), (M,), (3,) -> (M,3).  Call site: splat_trainer/scene/transfer_sh.py:49."""
  K = sh_features.shape[2]
  degree = int(round(math.sqrt(K))) - 1
  d = positions[indexes] - camera_pos
  d = d / d.norm(dim=-1, keepdim=True)
  Y = sh_basis(d, degree)                                  # (M, K)
  return (sh_features[indexes] * Y[:, None, :]).sum(-1) + 0.5


# ----------------------------------------------------------------------------- cull + project
def _camera(T_camera_world: torch.Tensor, projection: torch.Tensor):
  R = T_camera_world[:3, :3]
  t = T_camera_world[:3, 3]
  fx, fy, cx, cy = projection.unbind(0)
  return R, t, fx, fy, cx, cy


def frustum_cull(position, T_camera_world, projection, image_size, near, far, margin_px) -> torch.Tensor:
  """Indices (ascending, int64) of the points whose centre is inside the margin-expanded frustum."""
  R, t, fx, fy, cx, cy = _camera(T_camera_world, projection)
  W, H = image_size
  with torch.no_grad():
    pc = position @ R.t() + t
    z = pc[:, 2]
    zs = torch.where(z > 0, z, torch.ones_like(z))
    u = fx * pc[:, 0] / zs + cx
    v = fy * pc[:, 1] / zs + cy
    m = float(margin_px)
    mask = (z > near) & (z < far) & (u > -m) & (u < W + m) & (v > -m) & (v < H + m)
  return mask.nonzero().squeeze(1)


def quat_to_rotmat_xyzw(q: torch.Tensor) -> torch.Tensor:
  q = q / q.norm(dim=-1, keepdim=True)
  x, y, z, w = q.unbind(-1)
  return torch.stack([
      1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
      2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(*q.shape[:-1], 3, 3)


def project(position, log_scaling, rotation, alpha_logit, indexes, T_camera_world, projection, config
            ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
  """-> gaussians2d (M,6) [u v A B C opacity], depth (M,1), screen_scale (M,2).  Differentiable."""
  R, t, fx, fy, cx, cy = _camera(T_camera_world, projection)
  p = position[indexes]
  pc = p @ R.t() + t
  x, y, z = pc.unbind(-1)
  u = fx * x / z + cx
  v = fy * y / z + cy

  Rq = quat_to_rotmat_xyzw(rotation[indexes])
  S = torch.exp(log_scaling[indexes])
  Mw = Rq * S[:, None, :]                                  # basis = R diag(s)   (split.py:16-20)
  zero = torch.zeros_like(z)
  J = torch.stack([fx / z, zero, -fx * x / (z * z),
                   zero, fy / z, -fy * y / (z * z)], dim=-1).reshape(-1, 2, 3)
  Tm = (J @ R) @ Mw                                        # (M,2,3)
  cov = Tm @ Tm.transpose(1, 2)
  a0, b0, c0 = cov[:, 0, 0], cov[:, 0, 1], cov[:, 1, 1]

  opacity = torch.sigmoid(alpha_logit[indexes].squeeze(-1))
  blur = config.blur_cov + (config.aa_blur if config.antialias else 0.0)
  a, b, c = a0 + blur, b0, c0 + blur
  det = a * c - b * b
  if config.antialias:
    det0 = a0 * c0 - b0 * b0
    opacity = opacity * torch.sqrt(torch.clamp(det0 / det, min=0.0))
  A, B, C = c / det, -b / det, a / det

  mid = 0.5 * (a + c)
  rad = torch.sqrt(torch.clamp(mid * mid - det, min=0.0))
  screen_scale = torch.stack([torch.sqrt(mid + rad), torch.sqrt(torch.clamp(mid - rad, min=0.0))], dim=-1)

  g2d = torch.stack([u, v, A, B, C, opacity], dim=-1)
  return g2d, z.unsqueeze(-1), screen_scale.detach()


# ----------------------------------------------------------------------------- tile lists (conservative)
def _tile_lists(g2d: torch.Tensor, depth: torch.Tensor, image_size, config):
  """Conservative per-tile candidate lists in (depth, index) order.

  The raster spec is per pixel, so any superset of the true overlaps gives the same result;
  this uses the axis-aligned bounding box of the gaussian_scale-sigma ellipse."""
  W, H = image_size
  ts = config.tile_size
  tw, th = (W + ts - 1) // ts, (H + ts - 1) // ts
  with torch.no_grad():
    g = g2d.detach().double()
    u, v, A, B, C = g[:, 0], g[:, 1], g[:, 2], g[:, 3], g[:, 4]
    det = A * C - B * B
    hx = config.gaussian_scale * torch.sqrt(torch.clamp(C / det, min=0.0)) + 1e-3
    hy = config.gaussian_scale * torch.sqrt(torch.clamp(A / det, min=0.0)) + 1e-3
    # pixel centres of tile column k span [k*ts + .5, k*ts + ts - .5]
    x0 = torch.clamp(torch.floor((u - hx - 0.5) / ts), 0, tw).long()
    x1 = torch.clamp(torch.floor((u + hx - 0.5) / ts) + 1, 0, tw).long()
    y0 = torch.clamp(torch.floor((v - hy - 0.5) / ts), 0, th).long()
    y1 = torch.clamp(torch.floor((v + hy - 0.5) / ts) + 1, 0, th).long()
    nx, ny = (x1 - x0).clamp_min(0), (y1 - y0).clamp_min(0)
    cnt = nx * ny
    order = torch.argsort(depth.detach().reshape(-1), stable=True)   # ties -> ascending index
    cnt_o = cnt[order]
    total = int(cnt_o.sum().item())
    splat = torch.repeat_interleave(order, cnt_o)                    # depth-ordered instances
    start = torch.cumsum(cnt_o, 0) - cnt_o
    local = torch.arange(total) - torch.repeat_interleave(start, cnt_o)
    nxs = nx[splat].clamp_min(1)
    tx = x0[splat] + local % nxs
    ty = y0[splat] + local // nxs
    tile = ty * tw + tx
    tile_sorted, perm = torch.sort(tile, stable=True)                # keeps depth order inside a tile
    splat_sorted = splat[perm]
    counts = torch.bincount(tile_sorted, minlength=tw * th)
    starts = torch.cumsum(counts, 0) - counts
  return splat_sorted, starts, counts, tw, th


@dataclass
class RasterOutputs:
  image: torch.Tensor                    # (H, W, C)  differentiable
  final_T: torch.Tensor                  # (H, W)
  visibility: torch.Tensor               # (M,)
  median_depth: Optional[torch.Tensor]   # (H, W)
  prune_cost: Optional[torch.Tensor] = None
  split_score: Optional[torch.Tensor] = None
  num_overlaps: int = 0
  pixel_margin: Optional[torch.Tensor] = None   # (H, W)  smallest decision margin along the pixel's list (want_margins)
  splat_margin: Optional[torch.Tensor] = None   # (M,)    smallest pixel margin among the pixels the splat touches
  splat_own_margin: Optional[torch.Tensor] = None   # (M,) smallest margin of the splat's OWN decisions (it is the candidate)


# A pixel's result is a sum of discrete decisions, one per splat of its list: inside the support and above the alpha
# threshold (q <= qlim = min(q_max, 2 ln(opacity / threshold))), transmittance still above T_eps, alpha below the clamp
# (the clamp only switches the gradient through G), and which of two splats comes first.  An fp32 implementation within
# rounding of a boundary may decide the other way and move the pixel -- and the sums of every splat composited there -- by
# one minimal contribution.  ``want_margins`` reports how close the fp64 walk came to each boundary IN UNITS OF WHAT ONE
# fp32 ULP OF ROUNDING IN THE OPERANDS MOVES THE TESTED QUANTITY BY:
#   q vs qlim   one ulp in the pixel offsets (coordinates of magnitude |pixel|, plus |pixel - principal point| when the caller
#               names it: the mean is c + f x / z) and in qlim, and the rounding an fp32 CONIC
#               carries -- it is the inverse of the 2x2 screen covariance, so one ulp in the covariance's entries (relative
#               to its larger eigenvalue) is kappa ulps in the conic's, kappa = lambda_max / lambda_min of the conic:
#               dq1 = eps32 (2 (|t_x| |x| + |t_y| |y|) + kappa (|A| dx^2 + 2 |B dx dy| + |C| dy^2) + qlim),  t = conic d
#               (kappa is 1.5-4 on the benchmark scenes and reaches thousands in the fuzz sweep's needles: before round 4's
#               extended sweep the term had no kappa, and 6 gradient rows of 1600 random scenes -- all on splats with kappa
#               of 575-2430 whose few visible pixels lie at the rim of their support -- sat 9-77 "ulps" from a boundary)
#   T vs T_eps  a product of rounded factors 1 - alpha_i, i in front of the entry: each contributes 2 eps32 and, where the clamp
#               is inactive, alpha_i / (1 - alpha_i) (eps32 + dq1_i / 2):  dT1 = T (eps32 + sum_i f_i)
#               (until round 4's mid-size sweep: eps32 (j + 2) T, which understates T's rounding behind nearly opaque splats
#               a hundredfold -- one scene of 130 showed a saturation flip at "30 ulps" that the oracle's own fp32 run makes too)
#   alpha_raw vs the clamp:  da1 = alpha_raw (eps32 + dq1 / 2)
#   depth order of two list neighbours that could both contribute (inside their supports, the pixel alive at the first):
#               relative gap / eps32
#   (``loss_clamp`` = (lo, hi), the caller's: a channel of the composited pixel vs the bounds its loss clamps the image to --
#    the clamp passes the channel's gradient only inside them; unit = the rounding of the sum, see the code)
# so that a parity test can demand that every entry outside its tolerance sits within a STATED number of ulps of a boundary
# (tests/helpers.py: compare_explained).
MARGIN_EPS32 = 2.0 ** -23


def _composite_batch(g2d, feats, depth, idx, valid, pix, config, dL_dimage=None, want_median=False,
                     pix_valid=None, want_margins=False, loss_clamp=None, principal=None):
  """idx (B,L) splat ids (padded), valid (B,L), pix (B,P,2) pixel centres, pix_valid (B,P) inside-image mask.

  Returns image (B,P,C), final_T (B,P), w (B,P,L) weights, plus -- when dL_dimage (B,P,C) is given --
  the per-(pixel,splat) analytic |dL/dalpha| alpha and ||dL/dmean|| terms (no autograd)."""
  g = g2d[idx]                                             # (B,L,6)
  f = feats[idx]                                           # (B,L,C)
  dx = pix[:, :, None, 0] - g[:, None, :, 0]
  dy = pix[:, :, None, 1] - g[:, None, :, 1]
  A, Bc, Cc, op = g[:, None, :, 2], g[:, None, :, 3], g[:, None, :, 4], g[:, None, :, 5]
  q = A * dx * dx + 2.0 * Bc * dx * dy + Cc * dy * dy
  G = torch.exp(-0.5 * q)
  a_raw = op * G
  alpha = torch.clamp(a_raw, max=config.clamp_max_alpha)
  with torch.no_grad():
    contrib = valid[:, None, :] & (q <= config.gaussian_scale ** 2) & (alpha >= config.alpha_threshold)
    if pix_valid is not None:
      contrib = contrib & pix_valid[:, :, None]          # pixels of a partial tile that lie outside the image
  alpha = torch.where(contrib, alpha, torch.zeros_like(alpha))
  one_m = 1.0 - alpha
  T_incl = torch.cumprod(one_m, dim=2)
  T_excl = torch.cat([torch.ones_like(T_incl[:, :, :1]), T_incl[:, :, :-1]], dim=2)
  with torch.no_grad():
    live = T_excl >= config.transmittance_eps               # pixel stops once T < eps
  w = torch.where(live, alpha * T_excl, torch.zeros_like(alpha))
  image = torch.einsum('bpl,blc->bpc', w, f)
  with torch.no_grad():
    alpha_live = torch.where(live, alpha, torch.zeros_like(alpha))
    final_T = torch.prod(1.0 - alpha_live, dim=2)
    median = None
    if want_median:
      # depth of the first splat at which the accumulated opacity reaches one half (T_after < 0.5)
      T_after = torch.where(live, T_incl, torch.ones_like(T_incl))
      crossed = (T_after < 0.5) & live & contrib
      first = torch.where(crossed.any(dim=2), crossed.float().argmax(dim=2), torch.zeros_like(crossed[:, :, 0], dtype=torch.long))
      dsel = torch.gather(depth.reshape(-1)[idx][:, None, :].expand(-1, pix.shape[1], -1), 2, first[:, :, None]).squeeze(2)
      median = torch.where(crossed.any(dim=2), dsel, torch.zeros_like(dsel))

  extra = None
  if dL_dimage is not None:
    with torch.no_grad():
      gc = torch.einsum('bpc,blc->bpl', dL_dimage, f)       # g . c_i
      wg = w * gc
      suffix = torch.flip(torch.cumsum(torch.flip(wg, dims=[2]), dim=2), dims=[2]) - wg   # sum_{j>i}
      dL_dalpha = torch.where(live & contrib, T_excl * gc - suffix / one_m, torch.zeros_like(gc))
      prune = dL_dalpha.abs() * alpha
      unclamped = (a_raw <= config.clamp_max_alpha)
      dL_dG = torch.where(unclamped, dL_dalpha * op, torch.zeros_like(gc))
      gx = dL_dG * G * (A * dx + Bc * dy)
      gy = dL_dG * G * (Bc * dx + Cc * dy)
      split = torch.sqrt(gx * gx + gy * gy)
      extra = (prune, split)
  margins = None
  if want_margins:
    with torch.no_grad():
      inf = torch.full_like(q, float("inf"))
      e32 = MARGIN_EPS32
      qlim = torch.minimum(torch.full_like(op, config.gaussian_scale ** 2),
                           2.0 * torch.log((op / config.alpha_threshold).clamp_min(1e-300)))
      listed = valid[:, None, :] & (qlim > 0)
      if pix_valid is not None:
        listed = listed & pix_valid[:, :, None]
      inside = q <= qlim
      near_live = T_excl >= config.transmittance_eps * (1.0 - 1e-3)
      tx, ty = (A * dx + Bc * dy).abs(), (Bc * dx + Cc * dy).abs()
      xs, ys = pix[:, :, None, 0].abs().clamp_min(1.0), pix[:, :, None, 1].abs().clamp_min(1.0)
      if principal is not None:
        # the mean's coordinate is a SUM, u = c_x + f_x x / z: next to the image's left / top border it is a small difference
        # of two numbers of magnitude c_x, good to an ulp of THOSE (seen in the mid-size sweep: u = 4.3 off by 30 of its own ulps)
        xs = xs + (pix[:, :, None, 0] - float(principal[0])).abs()
        ys = ys + (pix[:, :, None, 1] - float(principal[1])).abs()
      half, det = 0.5 * (A + Cc), (A * Cc - Bc * Bc).clamp_min(1e-300)
      l1 = half + (half * half - det).clamp_min(0).sqrt()
      kappa = (l1 * l1 / det).clamp_min(1.0)                                # lambda_max / lambda_min of the conic
      dq1 = e32 * (2.0 * (tx * xs + ty * ys) + kappa * (A.abs() * dx * dx + 2.0 * (Bc * dx * dy).abs() + Cc.abs() * dy * dy)
                   + qlim.abs())
      m = torch.where(listed & near_live, (q - qlim).abs() / dq1.clamp_min(1e-300), inf)
      # T in front of list position j is a product of rounded factors 1 - alpha_i: each carries the rounding of the
      # subtraction and of the multiplication that folds it in (2 eps32) and -- where the clamp is inactive -- the rounding of
      # alpha_i = opacity exp(-q_i / 2) itself (eps32 + dq1_i / 2 relative) MAGNIFIED by alpha_i / (1 - alpha_i): behind a
      # stack of nearly opaque splats T is good to hundreds of ulps, not to (j + 2).  Entries that do not contribute are
      # exact factors of one.
      rel_alpha = e32 + 0.5 * dq1
      fr = torch.where(alpha > 0, 2.0 * e32 + torch.where(a_raw > config.clamp_max_alpha, torch.zeros_like(q),
                                                           alpha / one_m.clamp_min(1e-300) * rel_alpha), torch.zeros_like(q))
      dT1 = (torch.cumsum(fr, dim=2) - fr + e32) * T_excl.clamp_min(config.transmittance_eps * 0.5)
      m = torch.minimum(m, torch.where(listed & inside, (T_excl - config.transmittance_eps).abs() / dT1, inf))
      da1 = a_raw * (e32 + 0.5 * dq1)
      m = torch.minimum(m, torch.where(listed & inside & live, (a_raw - config.clamp_max_alpha).abs() / da1.clamp_min(1e-300), inf))
      # list neighbours that both contribute and whose depths an fp32 key cannot tell apart
      d = depth.reshape(-1)[idx]                                          # (B, L)
      gap = ((d[:, 1:] - d[:, :-1]).abs() / d[:, 1:].abs().clamp_min(1e-300)) / e32
      # (both are CANDIDATES on a pixel still alive at the first of them -- not "both contribute": behind a nearly opaque
      # first one the second is dead, and alive in the other order; seen in the held-out sweep, seed 3763: a saturated thumbnail
      # whose two largest splats are 0.1 ulp apart in depth)
      cand = listed & inside
      both = cand[:, :, 1:] & cand[:, :, :-1] & near_live[:, :, :-1]
      mg = torch.where(both, gap[:, None, :].expand_as(both), inf[:, :, 1:])
      m[:, :, 1:] = torch.minimum(m[:, :, 1:], mg)
      m[:, :, :-1] = torch.minimum(m[:, :, :-1], mg)
      mpx = m.min(dim=2).values                                           # (B, P)
      if loss_clamp is not None:
        # the CALLER's decision: a loss on image.clamp(lo, hi) passes a channel's gradient only inside [lo, hi], so a channel
        # within rounding of a bound switches the gradient of everything composited on that pixel.  One unit = what the
        # rounding counted above does to the sum: every term w_i c_i carries w's relative rounding (T's and alpha's) plus the
        # rounding of the accumulation.
        rel_w = dT1 / T_excl.clamp_min(1e-300) + da1 / a_raw.clamp_min(1e-300) + e32
        dC1 = torch.einsum('bpl,blc->bpc', w.detach() * rel_w, f.detach().abs())
        img = image.detach()
        dist = torch.minimum((img - loss_clamp[0]).abs(), (img - loss_clamp[1]).abs())
        mpx = torch.minimum(mpx, (dist / dC1.clamp_min(1e-300)).min(dim=2).values)
      # a splat is touched by a pixel's flip when it contributes there (everything behind the flipped splat moves with T)
      # or is itself the candidate (its own entry is the close one)
      sm = torch.minimum(torch.where(w > 0, mpx[:, :, None].expand_as(m), inf), m).min(dim=1).values   # (B, L)
      margins = (mpx, sm, m.min(dim=1).values)
  if want_margins:
    return image, final_T, w, median, extra, margins
  return image, final_T, w, median, extra


def rasterize(g2d: torch.Tensor, depth: torch.Tensor, feats: torch.Tensor, image_size, config,
              dL_dimage: Optional[torch.Tensor] = None, want_median: bool = False,
              tile_batch: int = 64, tiles: Optional[torch.Tensor] = None, lists=None,
              want_margins: bool = False, loss_clamp=None, principal=None) -> RasterOutputs:
  """Tile-batched compositing of projected splats.  Differentiable wrt g2d and feats (autograd).

  ``dL_dimage`` (H,W,C): when given, also returns the per-point heuristics prune_cost / split_score
  (they are functions of the incoming image gradient).  ``tiles``: optional subset of tile ids to
  render (others stay zero) -- used for the bounded CPU-baseline sample in bench.py."""
  W, H = image_size
  ts = config.tile_size
  M = g2d.shape[0]
  C = feats.shape[1]
  dtype, dev = g2d.dtype, g2d.device
  # ``lists``: the result of _tile_lists() when the caller renders several tile subsets of one frame
  splat_sorted, starts, counts, tw, th = lists if lists is not None else _tile_lists(g2d, depth, image_size, config)

  image = torch.zeros(th * ts, tw * ts, C, dtype=dtype, device=dev)
  final_T = torch.ones(th * ts, tw * ts, dtype=dtype, device=dev)
  median = torch.zeros(th * ts, tw * ts, dtype=dtype, device=dev) if want_median else None
  vis = torch.zeros(M, dtype=dtype, device=dev)
  prune = torch.zeros(M, dtype=dtype, device=dev) if dL_dimage is not None else None
  split = torch.zeros(M, dtype=dtype, device=dev) if dL_dimage is not None else None
  pmargin = torch.full((th * ts, tw * ts), float("inf"), dtype=dtype, device=dev) if want_margins else None
  smargin = torch.full((M,), float("inf"), dtype=dtype, device=dev) if want_margins else None
  omargin = torch.full((M,), float("inf"), dtype=dtype, device=dev) if want_margins else None
  gpad = None
  if dL_dimage is not None:
    gpad = torch.zeros(th * ts, tw * ts, C, dtype=dtype, device=dev)
    gpad[:H, :W] = dL_dimage.to(dtype)

  oy, ox = torch.meshgrid(torch.arange(ts), torch.arange(ts), indexing='ij')
  oy, ox = oy.reshape(-1), ox.reshape(-1)                  # (P,)

  tile_ids = torch.arange(tw * th) if tiles is None else tiles.long()
  tile_ids = tile_ids[counts[tile_ids] > 0]
  # batch tiles of similar list length together to limit padding
  tile_ids = tile_ids[torch.argsort(counts[tile_ids], stable=True)]
  image_parts = []
  for b0 in range(0, tile_ids.numel(), tile_batch):
    tb = tile_ids[b0:b0 + tile_batch]
    L = int(counts[tb].max().item())
    ar = torch.arange(L)
    valid = ar[None, :] < counts[tb][:, None]
    pos = (starts[tb][:, None] + ar[None, :]).clamp_max(max(splat_sorted.numel() - 1, 0))
    idx = torch.where(valid, splat_sorted[pos], torch.zeros_like(pos))
    ty, tx = tb // tw, tb % tw
    py = ty[:, None] * ts + oy[None, :]
    px = tx[:, None] * ts + ox[None, :]
    pix = torch.stack([px.to(dtype) + 0.5, py.to(dtype) + 0.5], dim=-1)
    gB = gpad[py, px] if gpad is not None else None
    res = _composite_batch(g2d, feats, depth, idx, valid, pix, config, gB, want_median,
                           pix_valid=(px < W) & (py < H), want_margins=want_margins, loss_clamp=loss_clamp,
                           principal=principal)
    img, fT, w, med, extra = res[:5]
    image_parts.append((py, px, img))
    with torch.no_grad():
      if want_margins:
        pmargin[py, px] = res[5][0]
        import warnings
        with warnings.catch_warnings():
          warnings.simplefilter("ignore")                   # (index_reduce_ is flagged "beta")
          smargin.index_reduce_(0, idx.reshape(-1), res[5][1].reshape(-1), "amin", include_self=True)
          omargin.index_reduce_(0, idx.reshape(-1), res[5][2].reshape(-1), "amin", include_self=True)
      final_T[py, px] = fT
      if want_median:
        median[py, px] = med
      vis.index_add_(0, idx.reshape(-1), w.sum(dim=1).reshape(-1))
      if extra is not None:
        prune.index_add_(0, idx.reshape(-1), extra[0].sum(dim=1).reshape(-1))
        split.index_add_(0, idx.reshape(-1), extra[1].sum(dim=1).reshape(-1))

  if image_parts:
    py = torch.cat([p[0].reshape(-1) for p in image_parts])
    px = torch.cat([p[1].reshape(-1) for p in image_parts])
    vals = torch.cat([p[2].reshape(-1, C) for p in image_parts])
    image = image.index_put((py, px), vals)               # differentiable scatter (each pixel once)
  return RasterOutputs(image=image[:H, :W], final_T=final_T[:H, :W], visibility=vis,
                       median_depth=median[:H, :W] if want_median else None,
                       prune_cost=prune, split_score=split,
                       num_overlaps=int(counts.sum().item()),
                       pixel_margin=pmargin[:H, :W] if want_margins else None, splat_margin=smargin,
                       splat_own_margin=omargin)


def rasterize_dense(g2d, depth, feats, image_size, config, dL_dimage=None, want_median=False) -> RasterOutputs:
  """Every splat against every pixel, no tiles: the small-case definition the tiled form must equal."""
  W, H = image_size
  M, C = g2d.shape[0], feats.shape[1]
  dtype = g2d.dtype
  order = torch.argsort(depth.detach().reshape(-1), stable=True)
  ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
  pix = torch.stack([xs.reshape(-1).to(dtype) + 0.5, ys.reshape(-1).to(dtype) + 0.5], dim=-1)[None]
  idx = order[None, :]
  valid = torch.ones(1, M, dtype=torch.bool)
  gB = dL_dimage.reshape(1, H * W, C).to(dtype) if dL_dimage is not None else None
  img, fT, w, med, extra = _composite_batch(g2d, feats, depth, idx, valid, pix, config, gB, want_median)
  vis = torch.zeros(M, dtype=dtype).index_add_(0, order, w.detach().sum(dim=1).reshape(-1))
  prune = split = None
  if extra is not None:
    prune = torch.zeros(M, dtype=dtype).index_add_(0, order, extra[0].sum(dim=1).reshape(-1))
    split = torch.zeros(M, dtype=dtype).index_add_(0, order, extra[1].sum(dim=1).reshape(-1))
  return RasterOutputs(image=img.reshape(H, W, C), final_T=fT.reshape(H, W), visibility=vis,
                       median_depth=med.reshape(H, W) if want_median else None,
                       prune_cost=prune, split_score=split, num_overlaps=0)


# ----------------------------------------------------------------------------- one-call form
def render(position, log_scaling, rotation, alpha_logit, features, T_camera_world, projection,
           image_size, near, far, config, use_sh: bool = False, want_median: bool = False,
           tiles: Optional[torch.Tensor] = None):
  """cull -> project -> (SH) -> rasterize.  Returns (RasterOutputs, g2d, depth, screen_scale, indexes)."""
  idx = frustum_cull(position, T_camera_world, projection, image_size, near, far,
                     config.margin_tiles * config.tile_size)
  g2d, depth, screen_scale = project(position, log_scaling, rotation, alpha_logit, idx,
                                     T_camera_world, projection, config)
  if use_sh:
    R = T_camera_world[:3, :3]
    cam_pos = -(R.t() @ T_camera_world[:3, 3])
    feats = evaluate_sh_at(features, position, idx, cam_pos)
  else:
    feats = features[idx]
  out = rasterize(g2d, depth, feats, image_size, config, want_median=want_median, tiles=tiles)
  return out, g2d, depth, screen_scale, idx
