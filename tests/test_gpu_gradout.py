"""The fused gradient buffers' contract (renderer.GradOut): whichever call form the caller uses, whichever exchange
delivers the sums, and whether a camera is the first of its batch (buffers declared uninitialised: some backward node
overwrites them) or a later one (accumulation), every term of every gradient must arrive -- checked against plain
autograd (no GradOut at all), with the owner object in debug mode recording who initialised what.
The reference accumulates into .grad over the cameras of a batch and zeroes only in scene.step (trainer.py:500-514)."""
import pytest
import torch

import splat_trainer_amd as sta
from helpers import small_scene
from splat_trainer_amd.controller_math import PointState
from splat_trainer_amd.distributed import CameraShardedStep

pytestmark = pytest.mark.gpu
CFG = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
NAMES = ("position", "log_scaling", "rotation", "alpha_logit", "feature")


def _scene():
  g, cam = small_scene(2500, 160, 120, sh_degree=2, seed=21, sigma_px=3.0)
  cams = [cam, sta.CameraParams(cam.T_camera_world.clone(), cam.projection * 1.0, cam.image_size, cam.near_plane, cam.far_plane),
          sta.CameraParams(cam.T_camera_world.clone(), cam.projection * 1.0, cam.image_size, cam.near_plane, cam.far_plane)]
  cams[1].T_camera_world[0, 3] += 0.06
  cams[2].T_camera_world[1, 3] -= 0.04
  cams[2].T_camera_world[2, 3] += 0.8                  # further away: part of the scene leaves the frustum margin
  return g, [c.to("cuda") for c in cams]


def _leaves(g):
  return [t.clone().cuda().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]


def _render(form, params, cam, grad_out=None, collector=None):
  scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])
  if form == "one_call":
    return sta.render_gaussians(scene, cam, CFG, use_sh=True, grad_out=grad_out, sh_collector=collector)
  g2d, depth, idx = sta.project_to_image(scene, cam, CFG, grad_out=grad_out)
  sh_out = collector if collector is not None else ((grad_out.feature, grad_out.position, grad_out) if grad_out is not None else None)
  feats = sta.evaluate_sh_at(params[4], params[0], idx, cam.camera_position, grad_out=sh_out)
  return sta.render_projected(idx, g2d, feats, depth, cam, CFG)


def _loss(r, k):
  w = torch.linspace(0.5, 1.5, r.image.shape[1], device="cuda")[None, :, None] * (1.0 + 0.3 * k)
  return (((r.image.clamp(0, 1) - 0.4) ** 2) * w).mean() * 50.0


def _autograd_sums(g, cams):
  """Plain autograd, one-call form, gradient sums after each camera of the batch."""
  params = _leaves(g)
  sums = []
  for k, cam in enumerate(cams):
    _loss(_render("one_call", params, cam), k).backward()
    sums.append([p.grad.clone() for p in params])
  return sums


def _close(got, want, what):
  for n, a, b in zip(NAMES, got, want):
    scale = b.abs().max().clamp_min(1e-20)
    err = ((a - b).abs().max() / scale).item()
    assert err < 3e-5, (what, n, err)


@pytest.mark.parametrize("form", ["one_call", "three_call"])
@pytest.mark.parametrize("mode", ["single", "dp_factor", "dp_factor_light", "dp_all_reduce"])
def test_every_gradient_term_arrives_in_every_form_and_mode(form, mode):
  g, cams = _scene()
  want = _autograd_sums(g, cams)
  params = _leaves(g)
  if mode == "single":
    bufs = [torch.full_like(p, float("nan")) for p in params]              # garbage: nothing may be read before it is written
    go = sta.GradOut(**dict(zip(NAMES, bufs)), debug=True).begin_batch()
    for k, cam in enumerate(cams):
      _loss(_render(form, params, cam, grad_out=go), k).backward()
      _close(bufs, want[k], f"{form} single, after camera {k}")
      if k == 0:
        first = list(go.log)
    assert go.finish_batch() == []
    # the first camera's nodes initialised every buffer exactly once; later cameras only accumulate
    written = [n for node, action, names in first if action in ("overwrite", "zero-fill") for n in names]
    assert sorted(written) == sorted(NAMES), first
    assert all(action == "accumulate" for node, action, names in go.log[len(first):]), go.log[len(first):]
    assert all(p.grad is None for p in params)                             # autograd itself received nothing
    return
  kind = "all_reduce" if mode == "dp_all_reduce" else "sh_factor"
  for count in (1, len(cams)):                                             # a batch of one camera (the first), then all
    dp = CameraShardedStep(params, 1, 0, mode=kind, exchange_when_single=True, with_stats=False)
    dp.grad_out.debug = True
    state = PointState.new_zeros(params[0].shape[0], "cuda") if mode == "dp_factor_light" else None

    def render_backward(j, cam, grad_out, collector):
      r = _render(form, params, cam, grad_out=grad_out, collector=collector)
      _loss(r, j).backward()
      return r

    for _ in range(2):                                                     # twice: the per-batch state must reset
      dp.run(cams[:count], render_backward, point_state=state)
    _close([dp.grads[n] for n in NAMES], want[count - 1], f"{form} {mode}, batch of {count}")
    assert not dp.grad_out.geometry_uninitialized and not dp.grad_out.feature_uninitialized


def test_protocol_violations_raise_in_debug_mode():
  g, cams = _scene()
  params = _leaves(g)
  bufs = [torch.zeros_like(p) for p in params]
  go = sta.GradOut(**dict(zip(NAMES, bufs)), debug=True).begin_batch()
  assert go.claim_overwrite("a", ("feature",)) is True
  assert go.claim_overwrite("b", ("feature",)) is False                    # already written: b must accumulate
  go.begin_batch()
  go._fresh.discard("position")                                            # position holds a term, the other three do not
  bufs[1].fill_(7.0)
  assert go.claim_overwrite("c", go.GEOMETRY) is False                      # mixed: the fresh ones are zero-filled, c accumulates
  assert bufs[1].abs().max() == 0
  go.begin_batch(feature=True, geometry=False)
  go.claim_overwrite("d", ("feature",))
  go._fresh.add("feature")                                                 # a caller re-declaring mid-batch ...
  with pytest.raises(sta.GsplatHipError):
    go.claim_overwrite("e", ("feature",))                                  # ... makes a second overwrite: caught
  go.begin_batch()
  assert sorted(go.finish_batch()) == sorted(NAMES) and go.finish_batch() == []
