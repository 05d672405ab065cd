"""The data-parallel exchanges of distributed.CameraShardedStep on one GPU -- the default (sharded replay: SUM and MAX
all-reduce, all-gather of the colour-factor blocks, all-to-all of the score slices, all-gather of the slice state) and the
round-3 form (dense per-camera blocks) -- with RCCL's collectives on a ONE-rank group: packing, the SH-gradient rebuild
over all cameras and the in-order replay of the controller statistics, against the sequential per-camera loop."""
import os

import pytest
import torch
import torch.distributed as dist

import splat_trainer_amd as sta
from helpers import small_scene
from splat_trainer_amd import synthetic
from splat_trainer_amd.controller_math import PointState
from splat_trainer_amd.densify import (dp_block_floats, dp_finish, dp_pack, dp_pack_sharded, dp_replay, dp_replay_slice,
                                       dp_slice_len)
from splat_trainer_amd.distributed import CameraShardedStep

pytestmark = pytest.mark.gpu
CFG = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)


@pytest.fixture(scope="module")
def one_rank_group():
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29500 + os.getpid() % 2000))
  try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
  except Exception as e:   # noqa: BLE001 -- an environment without a usable RCCL rendezvous is not a defect of the path
    pytest.skip(f"one-rank RCCL group unavailable here: {e}")
  yield
  dist.destroy_process_group()


def _scene(culled: bool):
  if culled:
    g, cams = synthetic.scene_b(20_000, 320, 240, sh_degree=2, seed=2, num_cameras=3, radius=1.2)   # frustum cull active
  else:
    g, cam = small_scene(3000, 160, 120, sh_degree=2, seed=9, sigma_px=3.0)
    cams = [cam, sta.CameraParams(cam.T_camera_world.clone(), cam.projection * 1.0, cam.image_size, cam.near_plane, cam.far_plane)]
    cams[1].T_camera_world[0, 3] += 0.05
  return g, [c.to("cuda") for c in cams]


def _run(g, cams, batches=2, **step_options):
  params = [t.clone().cuda().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  scene = sta.Gaussians3D(position=params[0], rotation=params[2], log_scaling=params[1], alpha_logit=params[3], feature=params[4])
  target = torch.full((cams[0].image_size[1], cams[0].image_size[0], 3), 0.5, device="cuda")

  def render_backward(j, cam, grad_out, collector):
    with torch.enable_grad():
      r = sta.render_gaussians(scene, cam, CFG, use_sh=True, grad_out=grad_out, sh_collector=collector)
      sta.clamped_mse_loss(r.image, target).backward()
    return r

  dp = CameraShardedStep(params, 1, 0, **step_options)
  state = PointState.new_zeros(params[0].shape[0], "cuda")
  for _ in range(batches):                               # two batches: everything per-batch must reset in between
    dp.run(cams, render_backward, point_state=state)
  return {k: v.clone() for k, v in dp.grads.items()}, state, dp.visible.clone()


@pytest.mark.parametrize("sharded", [True, False])
@pytest.mark.parametrize("culled", [False, True])
def test_dense_exchange_on_one_rank_equals_the_sequential_loop(one_rank_group, culled, sharded):
  g, cams = _scene(culled)
  want_g, want_s, want_v = _run(g, cams)                                  # no exchange: the reference's own loop
  got_g, got_s, got_v = _run(g, cams, exchange_when_single=True, sharded_replay=sharded)   # through RCCL (one rank)
  for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view"):
    assert torch.equal(getattr(got_s, f), getattr(want_s, f)), f          # what the densify masks are made of: bit-exact
  assert torch.allclose(got_s.visibility, want_s.visibility, rtol=1e-6, atol=1e-7)
  assert torch.allclose(got_v, want_v, rtol=1e-6, atol=1e-7)
  for k in want_g:                                                        # same sums, different association order
    scale = want_g[k].abs().max().clamp_min(1e-20)
    assert (got_g[k] - want_g[k]).abs().max() / scale < 2e-5, k
  assert want_s.points_in_view.max() >= 2 and want_s.split_score.abs().max() > 0


def test_config3_batch_goes_through_the_exchange(one_rank_group):
  """BASELINE.json configs[2] at full size -- 3M Gaussians, 1080p, SH degree 3, the batch of 8 orbit cameras -- through
  the default exchange on the one-rank RCCL group: 8 dense 72 MB camera blocks packed and all-gathered, the 156 MB
  all-reduce, the SH gradient rebuilt from the factors of all 8 cameras and their controller scores replayed in camera
  order, against the sequential loop over the same cameras.  Everything the densification masks are made of must be
  bit-identical; gradients agree up to the association order of the sums."""
  g, cams = synthetic.scene_b(3_000_000, 1920, 1080, sh_degree=3, seed=1, num_cameras=8)
  cams = [c.to("cuda") for c in cams]
  want_g, want_s, want_v = _run(g, cams, batches=1)
  got_g, got_s, got_v = _run(g, cams, batches=1, exchange_when_single=True)
  for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view"):
    assert torch.equal(getattr(got_s, f), getattr(want_s, f)), f
  assert torch.allclose(got_s.visibility, want_s.visibility, rtol=1e-5, atol=1e-6)
  assert torch.allclose(got_v, want_v, rtol=1e-5, atol=1e-6)
  for k in want_g:
    scale = want_g[k].abs().max().clamp_min(1e-20)
    assert (got_g[k] - want_g[k]).abs().max() / scale < 2e-5, k
  assert int(want_s.points_in_view.max()) == 8 and want_s.split_score.abs().max() > 0
  # the masks themselves, through the reference's controller rule (target_controller.py:150-160 via controller_math)
  from splat_trainer_amd.controller_math import find_split_prune_indexes
  a = find_split_prune_indexes(want_s, 0.25, 3_300_000, min_views=5, max_scale_px=200.0)
  b = find_split_prune_indexes(got_s, 0.25, 3_300_000, min_views=5, max_scale_px=200.0)
  assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and bool(a[0].any())


def test_dp_pack_and_replay_match_their_cpu_forms():
  n, m = 1000, 640
  gen = torch.Generator().manual_seed(3)
  idx = torch.randperm(n, generator=gen)[:m].sort().values
  d = dict(dcol=torch.randn(m, 3, generator=gen), split=torch.rand(m, generator=gen), prune=torch.rand(m, generator=gen),
           scale=torch.rand(m, 2, generator=gen) * 30, cam=torch.randn(3, generator=gen))
  vis = torch.rand(m, generator=gen)
  vis[::3] = 0.0
  blocks, all_sums = [], []
  for dev in ("cpu", "cuda"):
    b = torch.full((2, dp_block_floats(n)), 7.0, device=dev)              # garbage: every float must be (re)written
    sums = torch.ones(2 * n, device=dev)
    dp_pack(b[0], n, idx.to(dev), d["dcol"].to(dev), d["split"].to(dev), d["prune"].to(dev), d["scale"].to(dev), d["cam"].to(dev),
            visibility=vis.to(dev), sums=sums)
    all_sums.append(sums)
    full = torch.arange(n, device=dev)
    dp_pack(b[1], n, full, torch.ones(n, 3, device=dev), torch.full((n,), 0.5, device=dev), torch.full((n,), 0.25, device=dev),
            torch.full((n, 2), 2.0, device=dev), d["cam"].to(dev))
    blocks.append(b)
  assert torch.equal(torch.nan_to_num(blocks[0], nan=-1.0), torch.nan_to_num(blocks[1].cpu(), nan=-1.0))
  assert torch.isnan(blocks[1][0, 3 * n + 3:5 * n + 3]).sum() == 2 * (n - m)
  assert torch.equal(all_sums[0], all_sums[1].cpu())
  assert all_sums[0][:n].sum() == pytest.approx(n + float(vis.sum()), rel=1e-5) and all_sums[0][n:].sum() == n + int((vis > 0).sum())
  slots = torch.tensor([1, 0, 1], dtype=torch.int32)
  states = []
  for b, sums, dev in zip(blocks, all_sums, ("cpu", "cuda")):
    st = PointState.new_zeros(n, dev)
    dp_replay(st, b, slots.to(dev), n, sums=sums)
    states.append(st)
  assert torch.equal(states[0].points_in_view, states[1].points_in_view.cpu())
  assert torch.equal(states[0].visibility, states[1].visibility.cpu())
  for f in ("split_score", "prune_cost", "max_scale_px"):
    assert torch.allclose(getattr(states[0], f), getattr(states[1], f).cpu(), rtol=2e-5, atol=2e-6), f   # expf/logf: host libm vs device


def test_sharded_pack_replay_finish_match_their_cpu_forms():
  """densify.dp_pack_sharded / dp_replay_slice / dp_finish (the sharded exchange's three launches) against their torch
  forms, with 4 ranks' worth of slices, two camera slots per rank and a point count that does not divide evenly."""
  n, G, cpr = 1003, 4, 2
  L = dp_slice_len(n, G)
  gen = torch.Generator().manual_seed(4)
  cams = []
  for c in range(3):                                                        # cameras 0, 1, 2 of slot 0 / 1 (one slot unused)
    m = 500 + 100 * c
    idx = torch.randperm(n, generator=gen)[:m].sort().values
    vis = torch.rand(m, generator=gen)
    vis[::4] = 0.0
    cams.append(dict(idx=idx, dcol=torch.randn(m, 3, generator=gen), split=torch.rand(m, generator=gen),
                     prune=torch.rand(m, generator=gen), scale=torch.rand(m, 2, generator=gen) * 30,
                     cam=torch.randn(3, generator=gen), vis=vis))
  full = dict(idx=torch.arange(n), dcol=torch.ones(n, 3), split=torch.full((n,), 0.5), prune=torch.full((n,), 0.25),
              scale=torch.full((n, 2), 2.0), cam=torch.zeros(3), vis=torch.ones(n))
  out = {}
  for dev in ("cpu", "cuda"):
    factors = torch.full((cpr, 3 * n + 3), 7.0, device=dev)                 # garbage: every float must be (re)written
    scores = torch.full((G, cpr, 2, L), 7.0, device=dev)
    smax, sums = torch.zeros(n, device=dev), torch.ones(2 * n, device=dev)
    for slot, d in enumerate((cams[0], full)):
      dp_pack_sharded(factors[slot], scores, smax, n, slot, d["idx"].to(dev), d["dcol"].to(dev), d["split"].to(dev),
                      d["prune"].to(dev), d["scale"].to(dev), d["cam"].to(dev), visibility=d["vis"].to(dev), sums=sums)
    state = PointState.new_zeros(n, dev)
    state.split_score.copy_(torch.linspace(-1, 1, n))
    state.prune_cost.copy_(torch.linspace(2, 0, n))
    gathered = torch.zeros(G, 2, L, device=dev)
    for r in range(G):      # every "rank" replays its slice from the same send buffer (as if all ranks had sent this one)
      recv = scores[r:r + 1].expand(G, cpr, 2, L).contiguous()
      dp_replay_slice(state, recv, r, 2 * G, n, gathered[r])
    dp_finish(state, gathered, n, scale_max=smax, sums=sums)
    out[dev] = (factors, scores, smax, sums, state)
  a, b = out["cpu"], out["cuda"]
  # (padding cells behind point N of the last slice are never read: compare what carries points)
  sc_a = a[1].permute(1, 2, 0, 3).reshape(cpr, 2, G * L)[:, :, :n]
  sc_b = b[1].cpu().permute(1, 2, 0, 3).reshape(cpr, 2, G * L)[:, :, :n]
  assert torch.equal(a[0], b[0].cpu()) and torch.equal(torch.nan_to_num(sc_a, nan=-1.0), torch.nan_to_num(sc_b, nan=-1.0))
  assert torch.isnan(sc_a[0]).sum() == 2 * (n - 500) and not torch.isnan(sc_a[1]).any()
  assert torch.equal(a[2], b[2].cpu()) and torch.equal(a[3], b[3].cpu())
  assert torch.equal(a[4].points_in_view, b[4].points_in_view.cpu()) and torch.equal(a[4].max_scale_px, b[4].max_scale_px.cpu())
  for f in ("split_score", "prune_cost", "visibility"):
    assert torch.allclose(getattr(a[4], f), getattr(b[4], f).cpu(), rtol=2e-5, atol=2e-6), f        # expf/logf: host libm vs device
  assert (a[4].split_score != torch.linspace(-1, 1, n)).all()               # every point was replayed (camera `full`)


def test_early_factor_gather_changes_nothing(one_rank_group):
  """The colour-factor all-gather started from inside the last backward pass of the batch (factor blocks packed from the
  gradient rows right behind K7 + the per-splat reduction, gsr_frame_backward_stages) against the same exchange issued
  after the backward pass: every gradient and the whole controller state bit-identical -- with the frustum cull active,
  and with three cameras on one rank (the gather leaves with the third)."""
  g, cams = _scene(True)
  a = _run(g, cams, exchange_when_single=True, early_gather=True)
  b = _run(g, cams, exchange_when_single=True, early_gather=False)
  for k in a[0]:
    assert torch.equal(a[0][k], b[0][k]), k
  for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view", "visibility"):
    assert torch.equal(getattr(a[1], f), getattr(b[1], f)), f
  assert torch.equal(a[2], b[2])
