"""The reference's own call pattern around the boundary, restated with this build's imports
(splat_trainer/scene/mlp_scene.py:410-427 render, :268-288 reg_loss, :241-244 add_rendering;
trainer.py:500-514 evaluate_backward_with; controller/point_state.py:34-50): project_to_image ->
colour MLP on the gathered features (fp16 autocast) -> render_projected -> dataclasses.replace -> losses that
use rendering.points.visible.{opacity, depths} differentiably -> loss.backward() -> per-point outputs read
AFTER backward.  Every parameter gradient (Gaussian parameters AND the MLP weights) is compared with the same
flow run through the CPU oracle."""
from dataclasses import replace

import pytest
import torch
import torch.nn as nn

import splat_trainer_amd as sta
from helpers import oracle, rel_err, small_scene
from splat_trainer_amd.controller_math import PointState

pytestmark = pytest.mark.gpu


def saturate(t, gain=6.0, k=1.0):                    # util/misc.py:68-69 (golden-tested in test_oracle_golden)
  return (1 - 1 / torch.exp(gain * t)).pow(k)


class TinyColorModel(nn.Module):
  """Stand-in for scene/color_model.py: per-point feature (+ view direction) -> 3 colour channels."""

  def __init__(self, nfeat):
    super().__init__()
    torch.manual_seed(0)
    self.net = nn.Sequential(nn.Linear(nfeat + 3, 16), nn.SiLU(), nn.Linear(16, 3))

  def forward(self, feature, position, cam_pos):
    d = torch.nn.functional.normalize(position - cam_pos, dim=1)
    return torch.sigmoid(self.net(torch.cat([feature, d], dim=1)))


def reg_loss(points_visible, log_scaling_all):
  """mlp_scene.py:268-288 (scale / opacity / aspect terms, visibility-weighted)."""
  log_scale = log_scaling_all[points_visible.idx]
  scale = torch.exp(log_scale)
  norm_scale = scale.pow(2).sum(1) / points_visible.depths.pow(2).squeeze(-1)
  opacity_term = saturate(points_visible.opacity, gain=4.0, k=2.0) * norm_scale
  aspect = scale.max(1).values / scale.min(1).values
  w = points_visible.visibility
  return 0.1 * (norm_scale * w).mean() + 0.1 * (opacity_term * w).mean() + 0.01 * (aspect * w).mean()


@pytest.mark.parametrize("autocast,tol", [(False, 2e-4), (True, 3e-2)])
def test_reference_call_pattern_matches_oracle_flow(autocast, tol):
  """autocast=False isolates the rasterizer (tight tolerance); autocast=True is the reference's fp16 colour MLP,
  whose half-precision rounding (not the rasterizer) sets the error level."""
  g, cam = small_scene(600, 96, 72, sh_degree=0, seed=21, sigma_px=3.0)
  nfeat = 8
  torch.manual_seed(1)
  point_feature = torch.randn(600, nfeat)
  target = torch.rand(72, 96, 3)
  cfg_opts = dict(antialias=False, compute_visibility=True, compute_point_heuristic=True, blur_cov=0.3)   # trainer.py:305-310

  # ---------------- HIP path, written the way MLPScene.render is
  dev = "cuda"
  params = {n: getattr(g, n).clone().to(dev).requires_grad_(True) for n in ("position", "rotation", "log_scaling", "alpha_logit")}
  feature = point_feature.clone().to(dev).requires_grad_(True)
  model = TinyColorModel(nfeat).to(dev)
  camd = cam.to(dev)
  options = dict(cfg_opts)
  config = sta.pop_raster_config(options)                                             # scene/util.py:11-22
  gaussians = sta.Gaussians3D(feature=feature, **params)
  with torch.enable_grad():
    gaussians2d, depth, indexes = sta.project_to_image(gaussians, camd, config)     # mlp_scene.py:415
    with torch.autocast(device_type="cuda", dtype=torch.float16, enabled=autocast):   # mlp_scene.py:362
      colors = model(feature[indexes], params["position"][indexes], camd.camera_position)
    rendering = sta.render_projected(indexes, gaussians2d, colors, depth, camd, config, **options)   # :418
    rendering = replace(rendering, points=rendering.points.replace(attributes=colors),
                        image=rendering.image[..., :3].clamp(0, 1))                 # :422-424
    assert rendering.points.num_visible > 0                                         # trainer.py:507
    loss = torch.nn.functional.mse_loss(rendering.image, target.to(dev)) + \
        torch.nn.functional.l1_loss(rendering.image, target.to(dev)) + reg_loss(rendering.points.visible, params["log_scaling"])
    loss.backward()                                                                  # trainer.py:512
  state = PointState.new_zeros(600, dev)
  state.add_rendering(rendering)                                                     # point_state.py:34-50, after backward
  assert state.split_score.abs().sum() > 0 and state.prune_cost.abs().sum() > 0 and state.points_in_view.sum() > 0

  # ---------------- the same flow through the oracle (fp64, CPU)
  op = {n: getattr(g, n).clone().double().requires_grad_(True) for n in params}
  ofeature = point_feature.clone().double().requires_grad_(True)
  omodel = TinyColorModel(nfeat).double()
  T, proj = cam.T_camera_world.double(), cam.projection.double()
  oidx = oracle.frustum_cull(op["position"], T, proj, cam.image_size, cam.near_plane, cam.far_plane, 48)
  og2d, odepth, oss = oracle.project(op["position"], op["log_scaling"], op["rotation"], op["alpha_logit"], oidx, T, proj, config)
  ocolors = omodel(ofeature[oidx], op["position"][oidx], cam.camera_position.double())
  out = oracle.rasterize(og2d, odepth, ocolors, cam.image_size, config)
  oimage = out.image.clamp(0, 1)
  vis_rows = (out.visibility > 0).nonzero().squeeze(1)

  class V:                                                                           # points.visible view
    idx, depths, opacity, visibility = oidx[vis_rows], odepth[vis_rows], og2d[vis_rows, 5], out.visibility[vis_rows]
  oloss = torch.nn.functional.mse_loss(oimage, target.double()) + torch.nn.functional.l1_loss(oimage, target.double()) + \
      reg_loss(V, op["log_scaling"])
  oloss.backward()

  assert torch.equal(indexes.cpu(), oidx)
  assert abs(loss.item() - oloss.item()) < max(tol, 1e-5) * abs(oloss.item())
  for n in params:
    assert rel_err(params[n].grad, op[n].grad) < tol, (n, rel_err(params[n].grad, op[n].grad))
  assert rel_err(feature.grad, ofeature.grad) < tol
  for (na, pa), (nb, pb) in zip(model.named_parameters(), omodel.named_parameters()):
    assert rel_err(pa.grad, pb.grad) < tol, na
  # the alpha_logit gradient has a component that arrives only through points.visible.opacity (reg_loss)
  assert params["alpha_logit"].grad.abs().max() > 0


def test_two_rank_sh_factor_exchange_matches_all_reduce():
  """Two ranks (gloo, both on this GPU: the box has one) run bench.py's data-parallel step with the colour-gradient
  factor exchange and compare the summed gradients with a plain all-reduce of the full gradient buffer."""
  import json, os, subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  env = dict(os.environ, BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", "29533", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
         "--workload", "c1", "--backend", "gloo", "--collective", "sh_factor", "--check-collective"]
  out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300, cwd=root)
  assert out.returncode == 0, out.stderr[-2000:]
  line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
  res = json.loads(line)
  assert res["n_gpus"] == 2 and res["config"]["cameras_per_step"] == 2
  assert res["config"]["collective_check_rel_err"] < 1e-5


def test_prefetched_depth_order_is_used_only_for_the_tensor_it_was_made_for():
  """project_to_image(prefetch=d) enqueues the depth sort early; render_projected must take it only when it is handed
  the very same, unmodified depth tensor -- otherwise it sorts again (a stale order would composite in the wrong order)."""
  g, cam = small_scene(n=1500, w=96, h=64, sh_degree=0, seed=11)
  dev = "cuda"
  g, cam = g.to(dev), cam.to(dev)
  cfg = sta.RasterConfig()
  with torch.no_grad():
    d = {}
    g2d, depth, idx = sta.project_to_image(g, cam, cfg, prefetch=d)
    assert "depth_order" in d and d["depth_order"][1] is depth
    feats = torch.rand(idx.shape[0], 3, device=dev)
    ref = sta.render_projected(idx, g2d, feats, depth, cam, cfg).image                       # no prefetch: sorts itself
    same = sta.render_projected(idx, g2d, feats, depth, cam, cfg, _depth_order=d["depth_order"]).image
    assert torch.equal(same, ref)
    # a different depth tensor (reversed order of the splats in depth) with the stale prefetch handed in
    flipped = (depth.max() + depth.min() - depth).contiguous()
    want = sta.render_projected(idx, g2d, feats, flipped, cam, cfg).image
    got = sta.render_projected(idx, g2d, feats, flipped, cam, cfg, _depth_order=d["depth_order"]).image
    assert torch.equal(got, want) and not torch.equal(want, ref)
    # the same tensor modified in place after the prefetch: version counter differs -> ignored as well
    depth.copy_(flipped)
    got2 = sta.render_projected(idx, g2d, feats, depth, cam, cfg, _depth_order=d["depth_order"]).image
    assert torch.equal(got2, want)


def test_test_split_script_flow():
  """splat_trainer/scripts/test_split.py:21-38 with this build's imports: random camera + 5 random Gaussians -> render;
  split every Gaussian in two (split.py:87-113) -> render again.  The split keeps the picture close to the original."""
  from splat_trainer_amd.harness import split_gaussians_uniform
  sta.TaichiQueue.init(arch=None, debug=True)
  gen = torch.Generator().manual_seed(5)
  camera = sta.random_camera(image_size=(640, 480), generator=gen)
  gaussians = sta.random_3d_gaussians(5, camera, alpha_range=(0.5, 1.0), scale_factor=0.2, generator=gen)
  device = torch.device("cuda:0")
  camera, gaussians = camera.to(device), gaussians.to(device)
  with torch.no_grad():
    image = sta.render_gaussians(gaussians, camera_params=camera).image
    assert image.shape == (480, 640, 3) and torch.isfinite(image).all() and float(image.max()) > 0.05
    split = split_gaussians_uniform(gaussians.to_tensordict(), k=2, random_axis=False)
    image2 = sta.render_gaussians(sta.Gaussians3D.from_tensordict(split), camera_params=camera).image
  assert split["position"].shape[0] == 10
  covered = (image.sum(-1) > 0.02)
  covered2 = (image2.sum(-1) > 0.02)
  inter = (covered & covered2).sum().item(); union = (covered | covered2).sum().item()
  assert inter / union > 0.7                                    # same footprint, two smaller blobs per parent


def test_transfer_sh_flow():
  """splat_trainer/scene/transfer_sh.py:17-113 with this build's ``evaluate_sh_at``: per-point SH coefficients are
  fitted (Adam, base lr 0.1, higher orders lr/10 + weight decay) to view-dependent colours of the visible points of
  each camera, visibility-weighted MSE + 0.1 L1 on the base colour.  The fit must converge."""
  from splat_trainer_amd import synthetic
  dev = "cuda"
  g, cams = synthetic.scene_b(4000, 160, 120, sh_degree=2, seed=9, num_cameras=8)
  g = g.to(dev)
  cams = [c.to(dev) for c in cams]
  cfg = sta.RasterConfig(compute_visibility=True)
  positions = g.position
  true_sh = g.feature * 0.5                                                    # the "colour model" to be transferred

  def query_visibility(cam):                                                   # mlp_scene: render with visibility
    with torch.no_grad():
      r = sta.render_gaussians(g, cam, cfg, use_sh=True)
    vis = r.points.visibility > 0
    return r.points.idx[vis], r.points.visibility[vis]

  def eval_colors(idx, cam):
    with torch.no_grad():
      return sta.evaluate_sh_at(true_sh, positions, idx, cam.camera_position).clamp(0, 1)

  n = positions.shape[0]
  base_sh = torch.nn.Parameter(torch.randn(n, 3, 1, device=dev, generator=torch.Generator(device=dev).manual_seed(0)))
  higher_sh = torch.nn.Parameter(torch.zeros(n, 3, 8, device=dev))
  opt = torch.optim.Adam([dict(params=[base_sh], lr=0.1), dict(params=[higher_sh], lr=0.01, weight_decay=1e-4)],
                         betas=(0.9, 0.999))
  sh0 = 0.282094791773878
  losses = []
  for epoch in range(6):
    for cam in cams:
      opt.zero_grad()
      idx, vis = query_visibility(cam)
      colors = eval_colors(idx, cam)
      with torch.enable_grad():
        pred = sta.evaluate_sh_at(torch.cat([base_sh, higher_sh], dim=2), positions, idx, cam.camera_position).clamp(0, 1)
        mse = torch.nn.functional.mse_loss(pred, colors, reduction="none")
        rgb = torch.nn.functional.l1_loss((base_sh.squeeze(2) * sh0 + 0.5)[idx], colors)
        v = vis.unsqueeze(1)
        loss = (mse * v).sum() / v.sum() + 0.1 * rgb
        loss.backward()
      opt.step()
      losses.append(float(loss.detach()))
  assert losses[-1] < 0.1 * losses[0] and all(torch.isfinite(p).all() for p in (base_sh, higher_sh))


@pytest.mark.parametrize("form", ["three_call", "one_call"])
def test_fused_gradient_buffers_declared_uninitialised_get_every_term(form):
  """GradOut(geometry_uninitialized=True, feature_uninitialized=True): buffers full of garbage, no zero-fill by the
  caller.  In the three-call form autograd runs the SH backward (which ADDS the colour gradient's position term) before
  the projection's backward (which would overwrite every row): every term must still arrive, as with plain autograd."""
  g, cam = small_scene(3000, 160, 120, sh_degree=2, seed=21, sigma_px=3.0)
  cam = cam.to("cuda")
  cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
  target = torch.full((120, 160, 3), 0.4, device="cuda")

  def run(grad_out):
    params = [t.clone().cuda().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
    scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3],
                            feature=params[4])
    go = None
    if grad_out:
      bufs = [torch.full_like(p, 123.0) for p in params]               # garbage the backward pass must not keep
      go = sta.GradOut(position=bufs[0], log_scaling=bufs[1], rotation=bufs[2], alpha_logit=bufs[3], feature=bufs[4],
                       feature_uninitialized=True, geometry_uninitialized=True)
    with torch.enable_grad():
      if form == "one_call":
        r = sta.render_gaussians(scene, cam, cfg, use_sh=True, grad_out=go)
      else:
        g2d, depth, idx = sta.project_to_image(scene, cam, cfg, grad_out=go)
        sh_out = (go.feature, go.position, go) if go is not None else None
        feats = sta.evaluate_sh_at(params[4], params[0], idx, cam.camera_position, grad_out=sh_out)
        r = sta.render_projected(idx, g2d, feats, depth, cam, cfg)
      sta.clamped_mse_loss(r.image, target).backward()
    return bufs if grad_out else [p.grad for p in params]

  want, got = run(False), run(True)
  for name, a, b in zip(("position", "log_scaling", "rotation", "alpha_logit", "feature"), got, want):
    assert rel_err(a, b) < 1e-6, name
  assert want[0].abs().max() > 0
