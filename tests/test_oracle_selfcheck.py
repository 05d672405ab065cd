"""Internal consistency of the CPU oracle: the tiled form equals the dense per-pixel definition, the
gradients pass fp64 gradcheck, and the analytic per-pixel heuristics agree with autograd."""
import torch

from helpers import oracle, small_scene
from splat_trainer_amd import RasterConfig


def _projected(n=120, w=40, h=36, seed=5, dtype=torch.float64, sh_degree=0):
  g, cam = small_scene(n, w, h, sh_degree=sh_degree, seed=seed, sigma_px=2.5)
  cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  T, proj = cam.T_camera_world.to(dtype), cam.projection.to(dtype)
  idx = oracle.frustum_cull(g.position.to(dtype), T, proj, cam.image_size, cam.near_plane, cam.far_plane, 48)
  g2d, depth, ss = oracle.project(g.position.to(dtype), g.log_scaling.to(dtype), g.rotation.to(dtype),
                                  g.alpha_logit.to(dtype), idx, T, proj, cfg)
  feats = (g.feature[:, :, 0].to(dtype) * 0.28 + 0.5)[idx]
  return g, cam, cfg, g2d, depth, feats


def test_tiled_equals_dense():
  g, cam, cfg, g2d, depth, feats = _projected()
  gimg = torch.randn(cam.image_size[1], cam.image_size[0], 3, dtype=torch.float64)
  a = oracle.rasterize(g2d, depth, feats, cam.image_size, cfg, dL_dimage=gimg, want_median=True, tile_batch=7)
  b = oracle.rasterize_dense(g2d, depth, feats, cam.image_size, cfg, dL_dimage=gimg, want_median=True)
  for name in ("image", "final_T", "visibility", "median_depth", "prune_cost", "split_score"):
    x, y = getattr(a, name), getattr(b, name)
    assert torch.allclose(x, y, rtol=1e-10, atol=1e-12), name
  assert a.image.abs().max() > 0.1 and a.visibility.max() > 0


def test_weights_partition_unity():
  """With feature == 1 the image is the accumulated opacity: image + final_T == 1 at every pixel."""
  g, cam, cfg, g2d, depth, feats = _projected()
  ones = torch.ones(g2d.shape[0], 1, dtype=torch.float64)
  out = oracle.rasterize(g2d, depth, ones, cam.image_size, cfg)
  assert torch.allclose(out.image[..., 0] + out.final_T, torch.ones_like(out.final_T), atol=1e-12)


def test_gradcheck_project_and_raster_fp64():
  g, cam = small_scene(10, 16, 16, seed=11, sigma_px=3.0)
  cfg = RasterConfig()
  T, proj = cam.T_camera_world.double(), cam.projection.double()
  idx = torch.arange(10)
  feat = torch.rand(10, 3, dtype=torch.float64)

  def f(pos, ls, rot, al, ft):
    g2d, depth, _ = oracle.project(pos, ls, rot, al, idx, T, proj, cfg)
    return oracle.rasterize(g2d, depth, ft, cam.image_size, cfg).image

  args = [t.double().clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit)]
  args.append(feat.requires_grad_(True))
  assert torch.autograd.gradcheck(f, args, eps=1e-6, atol=1e-5, rtol=1e-3, nondet_tol=0.0)


def test_gradcheck_antialias_fp64():
  g, cam = small_scene(6, 16, 16, seed=12, sigma_px=1.0)
  cfg = RasterConfig(antialias=True, blur_cov=0.0)
  T, proj = cam.T_camera_world.double(), cam.projection.double()
  idx = torch.arange(6)

  def f(pos, ls, rot, al):
    g2d, depth, _ = oracle.project(pos, ls, rot, al, idx, T, proj, cfg)
    return g2d, depth

  args = [t.double().clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit)]
  assert torch.autograd.gradcheck(f, args, eps=1e-6, atol=1e-6, rtol=1e-3)


def test_analytic_pixel_terms_match_autograd():
  """sum over pixels of the SIGNED analytic dL/d(u,v) and dL/dalpha-derived terms equals autograd: validates
  the closed forms behind prune_cost / split_score (which sum magnitudes and so have no autograd twin)."""
  g, cam, cfg, g2d, depth, feats = _projected(n=60, w=32, h=32, seed=9)
  g2d = g2d.detach().requires_grad_(True)
  out = oracle.rasterize_dense(g2d, depth, feats, cam.image_size, cfg)
  gimg = torch.randn_like(out.image)
  (out.image * gimg).sum().backward()

  W, H = cam.image_size
  order = torch.argsort(depth.reshape(-1), stable=True)
  ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
  pix = torch.stack([xs.reshape(-1).double() + .5, ys.reshape(-1).double() + .5], -1)[None]

  # re-run the batch with a signed variant: patch abs/sqrt out by recomputing from the returned pieces
  gd = g2d.detach()
  idx = order[None]
  valid = torch.ones(1, gd.shape[0], dtype=torch.bool)
  img, fT, w, med, extra = oracle._composite_batch(gd, feats, depth, idx, valid, pix, cfg,
                                                   gimg.reshape(1, H * W, 3))
  prune, split = extra
  # |dL/dalpha| alpha summed must bound |sum alpha dL/dalpha| = |opacity * dL/dopacity| for unclamped splats
  d_op = g2d.grad[:, 5]
  lhs = (gd[:, 5] * d_op).abs()[order]
  assert (prune.sum(1).reshape(-1) + 1e-9 >= lhs - 1e-9).all()
  # split (sum of magnitudes) bounds the magnitude of the summed gradient
  mag = g2d.grad[:, :2].norm(dim=1)[order]
  assert (split.sum(1).reshape(-1) + 1e-9 >= mag - 1e-9).all()
  assert split.sum() > 0 and prune.sum() > 0


def test_margins_know_the_boundaries_they_are_asked_about():
  """Hand-built cases for the decision margins (want_margins): a pixel exactly on the rim of a support has margin ~0 there and
  a large one elsewhere; a needle's rim margin shrinks by its conic's condition number; a channel sitting on the bound of the
  caller's clamp is flagged only when ``loss_clamp`` says the loss clamps there."""
  cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  dt = torch.float64
  # one round splat centred on pixel centre (8.5, 8.5), sigma 2 px: conic 1/4; the pixel 6 px to the right sits exactly on q = 9
  g2d = torch.tensor([[8.5, 8.5, 0.25, 0.0, 0.25, 0.8]], dtype=dt)
  depth = torch.tensor([[1.0]], dtype=dt)
  feats = torch.tensor([[0.3, 0.6, 0.9]], dtype=dt)
  out = oracle.rasterize(g2d, depth, feats, (32, 16), cfg, want_margins=True)
  assert out.pixel_margin[8, 14] < 1e-6                    # q == 9 == qlim exactly
  assert out.pixel_margin[8, 8] > 1e4 and out.pixel_margin[8, 11] > 1e4
  assert out.splat_own_margin[0] < 1e-6
  # the same distance from the rim in q, round splat vs needle (condition 100): the conic part of the unit grows 100x
  # (6.25 -> 625 of 49 -> 668 in all, with the offset and q_lim parts unchanged): the margin shrinks 13.6x
  round_ = torch.tensor([[8.5, 8.5, 0.25, 0.0, 0.25, 0.8]], dtype=dt)
  needle = torch.tensor([[8.5, 8.5, 0.25, 0.0, 0.0025, 0.8]], dtype=dt)
  m_round = oracle.rasterize(round_, depth, feats, (32, 16), cfg, want_margins=True).pixel_margin[8, 13]    # dx = 5: q = 6.25
  m_needle = oracle.rasterize(needle, depth, feats, (32, 16), cfg, want_margins=True).pixel_margin[8, 13]
  assert 12 < m_round / m_needle < 15, (m_round, m_needle)
  # a channel exactly on the clamp's upper bound at the centre pixel: alpha = opacity there, colour = 1 / opacity
  feats1 = torch.tensor([[0.5, 1.0 / 0.8, 0.2]], dtype=dt)
  plain = oracle.rasterize(round_, depth, feats1, (32, 16), cfg, want_margins=True)
  clamped = oracle.rasterize(round_, depth, feats1, (32, 16), cfg, want_margins=True, loss_clamp=(0.0, 1.0))
  assert plain.pixel_margin[8, 8] > 1e4 and clamped.pixel_margin[8, 8] < 4.0
  assert clamped.splat_margin[0] < 4.0 and plain.image[8, 8, 1] == clamped.image[8, 8, 1]
  # far from both bounds nothing changes
  assert clamped.pixel_margin[8, 11] > 1e3


def test_margins_charge_the_mean_as_a_sum_and_ties_to_every_candidate_pixel():
  """(a) With the principal point named, a pixel next to the left border is charged the rounding of c_x, not of its own small
  coordinate; (b) two splats 0.1 ulp apart in depth: every pixel inside both supports and alive at the first is within that of a
  boundary -- also where the second one is dead behind the nearly opaque first."""
  cfg = RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  dt = torch.float64
  depth = torch.tensor([[1.0]], dtype=dt)
  feats = torch.tensor([[0.3, 0.6, 0.9]], dtype=dt)
  g2d = torch.tensor([[2.5, 8.5, 0.25, 0.0, 0.25, 0.8]], dtype=dt)          # centred on pixel (2, 8); (7, 8) has q = 6.25
  plain = oracle.rasterize(g2d, depth, feats, (512, 16), cfg, want_margins=True).pixel_margin[8, 7]
  named = oracle.rasterize(g2d, depth, feats, (512, 16), cfg, want_margins=True, principal=(256.0, 8.0)).pixel_margin[8, 7]
  assert 5.0 < plain / named < 40.0, (plain, named)
  # (b)
  two = torch.tensor([[8.5, 8.5, 0.02, 0.0, 0.02, 0.99999], [9.5, 8.5, 0.02, 0.0, 0.02, 0.9]], dtype=dt)
  depths = torch.tensor([[1.0], [1.0 + 0.1 * 2.0 ** -23]], dtype=dt)
  f2 = torch.tensor([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]], dtype=dt)
  cfg2 = RasterConfig(compute_visibility=True, compute_point_heuristic=True, clamp_max_alpha=0.99999)
  out = oracle.rasterize(two, depths, f2, (24, 16), cfg2, want_margins=True)
  assert out.final_T[8, 8] < cfg2.transmittance_eps                        # the first one saturates its centre pixel: the second is dead there
  assert out.image[8, 8, 1] == 0
  assert out.pixel_margin[8, 8] < 0.2 and out.pixel_margin[8, 12] < 0.2   # ... and both pixels are within the tie's 0.1 ulp
  assert out.splat_own_margin.max() < 0.2
