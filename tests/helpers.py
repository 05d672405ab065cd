"""Shared helpers for the parity tests (oracle = oracle/torch_oracle.py; product = splat_trainer_amd)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

from oracle import torch_oracle as oracle  # noqa: E402  (test infrastructure only)
from oracle import optim_oracle as oracle_optim  # noqa: E402,F401


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
  """max |a-b| relative to the reference tensor's max magnitude (the 1e-4-rel criterion of BASELINE.json,
  measured against the tensor scale so that near-zero entries do not dominate)."""
  a, b = a.detach().double().cpu(), b.detach().double().cpu()
  scale = max(b.abs().max().item(), 1e-30)
  return (a - b).abs().max().item() / scale


def frac_above(a: torch.Tensor, b: torch.Tensor, tol: float) -> float:
  """Fraction of entries whose error exceeds tol x the reference's max magnitude."""
  a, b = a.detach().double().cpu(), b.detach().double().cpu()
  scale = max(b.abs().max().item(), 1e-30)
  return ((a - b).abs() > tol * scale).double().mean().item()


# Observed parity numbers of this session: (label, key, max error relative to the tensor's max magnitude, entries above
# tol, entries, tol, entries violating the PER-ENTRY criterion).  tests/conftest.py prints the table at the end of the run
# and writes it to gpurun_out/parity_observed.txt so that every bound asserted in the tests can be read against what was
# actually measured.
PARITY_LOG = []


def per_entry_violations(a_: torch.Tensor, b_: torch.Tensor, tol: float) -> int:
  """Entries with |a - b| > tol |b| + 1e-2 tol max|b|: the error relative to the ENTRY itself (a gradient row a hundred
  times smaller than the tensor's largest must still be right to ``tol`` of its own size), with an absolute floor two
  orders below the tensor-relative criterion for entries that are (nearly) zero."""
  if not a_.numel():
    return 0
  scale = max(b_.abs().max().item(), 1e-30)
  return int(((a_ - b_).abs() > tol * b_.abs() + 1e-2 * tol * scale).sum().item())


def observe(label: str, key: str, a: torch.Tensor, b: torch.Tensor, tol: float):
  """Records and returns (max rel err, fraction of entries above tol) -- both relative to the tensor's largest magnitude;
  the per-entry figure (``per_entry_violations``) is logged next to them."""
  a_, b_ = a.detach().double().cpu(), b.detach().double().cpu()
  scale = max(b_.abs().max().item(), 1e-30)
  err = (a_ - b_).abs() / scale
  worst = err.max().item() if err.numel() else 0.0
  above = int((err > tol).sum().item())
  PARITY_LOG.append((label, key, worst, above, err.numel(), tol, per_entry_violations(a_, b_, tol)))
  return worst, above / max(err.numel(), 1)


# A pixel within fp32 rounding of a discrete contribute / skip boundary (q = 9, alpha = 1/255, T = 1e-4) takes the other
# branch than the fp64 oracle and moves that pixel -- and the per-point sums and gradients of the one or two splats
# involved -- by one minimal contribution.  WHICH scenes show such a flip depends on every rounding upstream (it moved
# when the forward projection's fma pattern was pinned in round 3), so the tests state the rule, not a list of scenes:
# every entry within tol, or the exceptions are isolated -- few (a share of the tensor, or 3 entries of a small one) and
# small (1e-2 of the tensor's largest magnitude).
FLIP_SHARE, FLIP_SIZE = 4e-3, 1e-2


def clean_or_isolated_flip(label: str, key: str, a: torch.Tensor, b: torch.Tensor, tol: float) -> bool:
  """Asserts the rule above for one tensor; returns True when a flip was seen."""
  worst, frac = observe(label, key, a, b, tol)
  n = a.numel()
  assert frac * n <= max(3.0, FLIP_SHARE * n) + 0.5 and worst < max(tol, FLIP_SIZE), (label, key, frac, worst)
  return frac > 0


GRAD_KEYS = ("d_position", "d_log_scaling", "d_rotation", "d_alpha_logit", "d_feature")
POINT_KEYS = ("visibility", "prune_cost", "split_score", "screen_scale", "depth")


def compare_to_oracle(label: str, hip: dict, orc: dict, tol: float = 1e-4, pixel_flips: float = 0.0,
                      point_flips: float = 0.0, worst_pixel: float = 0.0, worst_point: float = 0.0,
                      keys=("image", "final_T") + POINT_KEYS + GRAD_KEYS):
  """HIP vs oracle: EVERY entry within ``tol`` of the oracle, relative to the tensor's max magnitude (BASELINE.json's
  1e-4 criterion) -- the default, and what all scenes up to tens of thousands of splats satisfy with two orders of
  margin (observed max 3e-6, profiles/r02_parity_observed.txt).

  Only the full-size scenes pass allowances: among 6e7 contributing (pixel, splat) pairs a few dozen lie within fp32
  rounding of a discrete contribute/skip boundary (q = 9, alpha = 1/255, T = 1e-4) and take the other branch than the
  fp64 oracle.  ``pixel_flips`` / ``point_flips`` bound the SHARE of entries above ``tol`` (images / per-point sums and
  gradients, which a flipped pixel also moves), ``worst_*`` their size.  Every observed number is logged (PARITY_LOG)."""
  assert torch.equal(hip["idx"].cpu(), orc["idx"])
  for k in keys:
    image_like = k in ("image", "final_T", "median")
    worst, frac = observe(label, k, hip[k], orc[k], tol)
    assert frac <= (pixel_flips if image_like else point_flips), (label, k, frac, worst)
    assert worst < max(tol, worst_pixel if image_like else worst_point), (label, k, worst)


def small_scene(n=400, w=64, h=48, sh_degree=0, seed=3, sigma_px=3.0):
  import splat_trainer_amd.synthetic as syn
  return syn.scene_a(n, w, h, sh_degree=sh_degree, seed=seed, sigma_px=sigma_px)


def oracle_render_and_grads(g, cam, config, use_sh, target=0.5, want_median=False, dtype=torch.float64,
                            loss_scale=1.0):
  """Oracle forward + MSE loss + autograd backward + analytic per-point heuristics.  fp64 by default, so that
  the comparison measures the fp32 HIP path's error alone (not HIP error + the oracle's own fp32 rounding)."""
  pos = g.position.clone().to(dtype).requires_grad_(True)
  ls = g.log_scaling.clone().to(dtype).requires_grad_(True)
  rot = g.rotation.clone().to(dtype).requires_grad_(True)
  al = g.alpha_logit.clone().to(dtype).requires_grad_(True)
  feat = g.feature.clone().to(dtype).requires_grad_(True)
  T = cam.T_camera_world.to(dtype)
  proj = cam.projection.to(dtype)
  out, g2d, depth, sscale, idx = oracle.render(pos, ls, rot, al, feat, T, proj, cam.image_size, cam.near_plane,
                                                cam.far_plane, config, use_sh=use_sh, want_median=want_median)
  image = out.image
  image.retain_grad()
  g2d.retain_grad()
  loss = ((image.clamp(0, 1) - target) ** 2).mean() * loss_scale
  loss.backward()
  # heuristics need the incoming image gradient
  if use_sh:
    R = T[:3, :3]
    feats = oracle.evaluate_sh_at(feat.detach(), pos.detach(), idx, -(R.t() @ T[:3, 3]))
  else:
    feats = feat.detach()[idx]
  heur = oracle.rasterize(g2d.detach(), depth.detach(), feats, cam.image_size, config, dL_dimage=image.grad)
  return dict(image=image.detach(), final_T=out.final_T, visibility=out.visibility, median=out.median_depth,
              g2d=g2d.detach(), d_g2d=g2d.grad, depth=depth.detach(), screen_scale=sscale, idx=idx, loss=loss.detach(),
              d_position=pos.grad, d_log_scaling=ls.grad, d_rotation=rot.grad, d_alpha_logit=al.grad,
              d_feature=feat.grad, prune_cost=heur.prune_cost, split_score=heur.split_score,
              num_overlaps=out.num_overlaps)


def oracle_render_and_grads_chunked(g, cam, config, use_sh, target=0.5, dtype=torch.float64, loss_scale=1.0,
                                    chunk_tiles=1024):
  """Same outputs as oracle_render_and_grads for FULL-SIZE scenes: the composite stage runs tile chunk by tile chunk with
  a backward per chunk (the MSE is a sum over pixels), so the autograd graph of only one chunk is alive at a time; the
  per-point heuristics come from a second, gradient-free pass per chunk fed with that chunk's image gradient."""
  leaves = [t.clone().to(dtype).requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  pos, ls, rot, al, feat = leaves
  T, proj = cam.T_camera_world.to(dtype), cam.projection.to(dtype)
  W, H = cam.image_size
  idx = oracle.frustum_cull(pos, T, proj, cam.image_size, cam.near_plane, cam.far_plane,
                            config.margin_tiles * config.tile_size)
  g2d, depth, sscale = oracle.project(pos, ls, rot, al, idx, T, proj, config)
  if use_sh:
    R = T[:3, :3]
    feats = oracle.evaluate_sh_at(feat, pos, idx, -(R.t() @ T[:3, 3]))
  else:
    feats = feat[idx]
  g2d_d, feats_d = g2d.detach().requires_grad_(True), feats.detach().requires_grad_(True)
  lists = oracle._tile_lists(g2d_d, depth.detach(), cam.image_size, config)
  n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
  C = feats.shape[1]
  M = idx.shape[0]
  image = torch.zeros(H, W, C, dtype=dtype)
  final_T = torch.ones(H, W, dtype=dtype)
  vis, prune, split = (torch.zeros(M, dtype=dtype) for _ in range(3))
  overlaps = 0
  for t0 in range(0, n_tiles, chunk_tiles):
    tiles = torch.arange(t0, min(t0 + chunk_tiles, n_tiles))
    out = oracle.rasterize(g2d_d, depth.detach(), feats_d, cam.image_size, config, tiles=tiles, lists=lists)
    img = out.image
    img.retain_grad()
    loss = ((img.clamp(0, 1) - target) ** 2).sum() / (W * H * C) * loss_scale
    # pixels outside this chunk are constant zeros in `img`: they add a constant to the loss and nothing to any gradient
    loss.backward()
    with torch.no_grad():
      heur = oracle.rasterize(g2d_d.detach(), depth.detach(), feats_d.detach(), cam.image_size, config, tiles=tiles,
                              lists=lists, dL_dimage=img.grad)
      image += out.image.detach()
      final_T = torch.minimum(final_T, out.final_T)
      vis += out.visibility
      prune += heur.prune_cost
      split += heur.split_score
    overlaps = out.num_overlaps
  torch.autograd.backward([g2d, feats], [g2d_d.grad, feats_d.grad])
  return dict(image=image, final_T=final_T, visibility=vis, g2d=g2d.detach(), d_g2d=g2d_d.grad, depth=depth.detach(),
              screen_scale=sscale, idx=idx, d_position=pos.grad, d_log_scaling=ls.grad, d_rotation=rot.grad,
              d_alpha_logit=al.grad, d_feature=feat.grad, prune_cost=prune, split_score=split, num_overlaps=overlaps)


def hip_render_and_grads(g, cam, config, use_sh, target=0.5, want_median=False, device="cuda", loss_scale=1.0):
  import splat_trainer_amd as sta
  gd = sta.Gaussians3D(*(t.clone().to(device).requires_grad_(True) for t in
                         (g.position, g.rotation, g.log_scaling, g.alpha_logit, g.feature)))
  camd = cam.to(device)
  r = sta.render_gaussians(gd, camd, config, use_sh=use_sh, render_median_depth=want_median)
  loss = ((r.image.clamp(0, 1) - target) ** 2).mean() * loss_scale
  loss.backward()
  return dict(rendering=r, image=r.image.detach(), final_T=r.final_transmittance, visibility=r.points.visibility,
              median=r.median_depth_image, depth=r.points.depths.detach(), screen_scale=r.points.screen_scale,
              idx=r.points.idx, loss=loss.detach(), d_position=gd.position.grad, d_log_scaling=gd.log_scaling.grad,
              d_rotation=gd.rotation.grad, d_alpha_logit=gd.alpha_logit.grad, d_feature=gd.feature.grad,
              prune_cost=r.points.prune_cost, split_score=r.points.split_score, num_overlaps=r.num_overlaps)
