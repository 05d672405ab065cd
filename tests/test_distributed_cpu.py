"""The N > 1 path on CPU: world_size-2 gloo processes shard the cameras of a batch, accumulate gradients
into one flat buffer, sum them with one collective, and replay per-camera point statistics in camera order.
The renderer here is the CPU oracle (tests may use it); the product's kernels are covered by the -m gpu
tests -- this file tests the sharding / collective / ordering logic of splat-trainer_amd/distributed.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _scene():
  import splat_trainer_amd.synthetic as syn
  return syn.scene_b(300, 64, 48, sh_degree=1, seed=3, num_cameras=4, sigma_px=3.0)


def _render_loss(params, cam, cfg):
  from oracle import torch_oracle as oracle
  pos, ls, rot, al, feat = params
  out, g2d, depth, ss, idx = oracle.render(pos, ls, rot, al, feat, cam.T_camera_world, cam.projection, cam.image_size,
                                           cam.near_plane, cam.far_plane, cfg, use_sh=True)
  loss = ((out.image.clamp(0, 1) - 0.5) ** 2).mean()
  stats = dict(idx=idx, visibility=out.visibility, screen_scale_max=ss.max(1).values)
  return loss, stats


def _worker(rank, world, port, out_path):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  torch.set_num_threads(2)
  import splat_trainer_amd as sta
  from splat_trainer_amd.distributed import GradBucket, evaluate_backward_sharded, shard_cameras
  g, cams = _scene()
  cfg = sta.RasterConfig(compute_visibility=True)
  params = [t.clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  bucket = GradBucket(params, world)
  assert shard_cameras(len(cams), rank, world) == [j for j in range(len(cams)) if j % world == rank]
  stats = evaluate_backward_sharded(params, cams, lambda j, cam: _render_loss(params, cam, cfg), bucket=bucket,
                                    mode="reduce_scatter", device="cpu")      # gloo falls back to all_reduce
  assert [s["camera"] for s in stats] == list(range(len(cams)))
  torch.save(dict(grads=[p.grad.clone() for p in params], vis=[s["visibility"] for s in stats],
                  idx=[s["idx"] for s in stats]), f"{out_path}.{rank}")
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_gloo_matches_sequential(tmp_path):
  import splat_trainer_amd as sta
  world, port = 2, 29000 + (os.getpid() % 2000)
  out = str(tmp_path / "res")
  mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
  r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
  # sequential single-process reference: the reference's loop (trainer.py:500-514)
  g, cams = _scene()
  cfg = sta.RasterConfig(compute_visibility=True)
  params = [t.clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  seq_stats = []
  for cam in cams:
    loss, stats = _render_loss(params, cam, cfg)
    loss.backward()
    seq_stats.append(stats)
  for a, b, p in zip(r0["grads"], r1["grads"], params):
    assert torch.equal(a, b)                                   # every rank ends with the same summed gradient
    assert torch.allclose(a, p.grad, rtol=1e-5, atol=1e-9)
  for j, s in enumerate(seq_stats):                            # stats replayed in camera order on every rank
    assert torch.equal(r0["idx"][j], s["idx"]) and torch.equal(r1["idx"][j], s["idx"])
    assert torch.allclose(r0["vis"][j], s["visibility"]) and torch.allclose(r1["vis"][j], s["visibility"])


def test_grad_bucket_layout():
  from splat_trainer_amd.distributed import GradBucket
  ps = [torch.randn(7, 3, requires_grad=True), torch.randn(7, 4, requires_grad=True), torch.randn(7, 3, 4, requires_grad=True)]
  b = GradBucket(ps, world_size=8, extra=7)
  assert b.flat.numel() % 8 == 0
  for p, v in zip(ps, b.views):
    assert p.grad is v and v.shape == p.shape and (v.data_ptr() - b.flat.data_ptr()) % 256 == 0
  (ps[0].sum() * 2 + ps[2].sum()).backward()
  assert torch.all(b.views[0] == 2) and torch.all(b.views[2] == 1) and torch.all(b.views[1] == 0)
  assert ps[0].grad.data_ptr() == b.views[0].data_ptr()      # autograd accumulated in place
  b.zero()
  assert b.flat.abs().sum() == 0 and b.extra.shape == (7,)
  # zero(except_views): the named slots keep their contents (the SH backward overwrites them), everything else is cleared
  b.flat.fill_(3.0)
  b.zero(except_views=(2,))
  assert torch.all(b.views[2] == 3) and torch.all(b.views[0] == 0) and torch.all(b.views[1] == 0) and torch.all(b.extra == 0)
  b.flat.fill_(3.0)
  b.zero(except_views=(0, 1))
  assert torch.all(b.views[0] == 3) and torch.all(b.views[1] == 3) and torch.all(b.views[2] == 0) and torch.all(b.extra == 0)
  assert b.all_reduce() is None and b.all_reduce(async_op=True) is None          # no process group: nothing to do


def _factor_worker(rank, world, port, out_path):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from oracle import torch_oracle as oracle
  from splat_trainer_amd.distributed import gather_sh_factors, shard_cameras
  from splat_trainer_amd.sh import ShFactorCollector
  g, cams = _scene()
  n = g.position.shape[0]
  col = ShFactorCollector()
  mine = shard_cameras(len(cams), rank, world)
  for j in mine:                                  # what _SHFn.backward records: (visible rows, d_colour, camera position)
    gen = torch.Generator().manual_seed(50 + j)
    idx = torch.randperm(n, generator=gen)[: n // 2].sort().values
    col.items.append((idx, torch.randn(idx.numel(), 3, generator=gen), cams[j].camera_position.float()))
  block = gather_sh_factors(col, list(range(len(mine))), len(mine), n)      # (cameras, N+1, 3): last row = camera position
  G_all, cams_all = block[:, :n], block[:, n]
  # rebuild d_sh = sum_c g_c (x) Y(dir_c) the way csrc/geometry.hip: sh_bwd_multi_kernel does, with the oracle's basis
  d_sh = torch.zeros(n, 3, 4)
  for c in range(G_all.shape[0]):
    d = torch.nn.functional.normalize(g.position - cams_all[c], dim=1)
    d_sh += G_all[c][:, :, None] * oracle.sh_basis(d, 1)[:, None, :]
  torch.save(dict(d_sh=d_sh, G=G_all, cams=cams_all), f"{out_path}.{rank}")
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sh_factor_gather_rebuilds_the_summed_sh_gradient(tmp_path):
  """distributed.gather_sh_factors over gloo: both ranks end with the same rank-major factor block, and the rebuilt
  coefficient gradient equals the sum of the per-camera autograd gradients of evaluate_sh_at (oracle)."""
  from oracle import torch_oracle as oracle
  world, port = 2, 31000 + (os.getpid() % 2000)
  out = str(tmp_path / "fac")
  mp.spawn(_factor_worker, args=(world, port, out), nprocs=world, join=True)
  r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
  assert torch.equal(r0["G"], r1["G"]) and torch.equal(r0["cams"], r1["cams"]) and torch.equal(r0["d_sh"], r1["d_sh"])
  g, cams = _scene()
  n = g.position.shape[0]
  order = [0, 2, 1, 3]                                              # rank-major: rank 0 holds cameras 0, 2
  for slot, j in enumerate(order):
    assert torch.allclose(r0["cams"][slot], cams[j].camera_position.float())
  sh = g.feature.clone().requires_grad_(True)
  for j in range(4):
    gen = torch.Generator().manual_seed(50 + j)
    idx = torch.randperm(n, generator=gen)[: n // 2].sort().values
    dcol = torch.randn(idx.numel(), 3, generator=gen)
    col = oracle.evaluate_sh_at(sh, g.position, idx, cams[j].camera_position)
    col.backward(dcol)
  assert torch.allclose(r0["d_sh"], sh.grad, rtol=1e-4, atol=1e-6)


def _stats_worker(rank, world, port, out_path, num_cameras):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from splat_trainer_amd.controller_math import PointState, find_split_prune_indexes
  from splat_trainer_amd.distributed import (exchange_counts, gather_point_stats, gather_sh_factors, replay_point_stats,
                                             shard_cameras)
  from splat_trainer_amd.sh import ShFactorCollector
  n = 500
  mine = shard_cameras(num_cameras, rank, world)
  cpr = (num_cameras + world - 1) // world
  local = [_camera_stats(j, n) for j in mine]
  got = gather_point_stats(local, num_cameras, device="cpu")                    # rank 1 may hold no camera at all
  state = replay_point_stats(PointState.new_zeros(n, "cpu"), got)
  masks = find_split_prune_indexes(state, 0.2, 560, min_views=1, max_scale_px=50.0)
  # the factor gather, dense and packed, with the same counts
  col = ShFactorCollector()
  for j in mine:
    d = _camera_stats(j, n)
    col.items.append((d["idx"], d["dcol"], d["cam"]))
  counts = exchange_counts([(j, it[0].shape[0]) for j, it in zip(mine, col.items)], cpr, "cpu")
  dense = gather_sh_factors(col, list(range(len(mine))), cpr, n, device="cpu")
  packed = gather_sh_factors(col, list(range(len(mine))), cpr, n, device="cpu", visible_max=max(m for _, m in counts))
  fields = ("prune_cost", "split_score", "max_scale_px", "points_in_view", "visibility")
  # the LIGHT exchange of CameraShardedStep.run(point_state=...): max / sum / count as reductions, EMA inputs gathered
  import splat_trainer_amd as sta
  from splat_trainer_amd.distributed import CameraShardedStep
  params = [torch.zeros(n, d, requires_grad=True) for d in (3, 3, 4, 1)] + [torch.zeros(n, 3, 4, requires_grad=True)]
  dp = CameraShardedStep(params, world, rank, mode="all_reduce", fused_grad_out=False)     # the stand-in render writes no gradients

  def fake_render(j, cam, grad_out, collector):
    d = _camera_stats(j, n)
    pts = sta.RenderedPoints(idx=d["idx"], depths=torch.zeros(d["idx"].shape[0], 1), opacity=torch.zeros(d["idx"].shape[0]),
                             screen_scale=torch.stack([d["screen_scale_max"], 0.5 * d["screen_scale_max"]], dim=1),
                             visibility=d["visibility"], prune_cost=d["prune_cost"], split_score=d["split_score"])
    return sta.Rendering(image=None, camera=None, points=pts)

  light = PointState.new_zeros(n, "cpu")
  for _ in range(2):                                             # two batches: the reductions must reset in between
    assert dp.run(list(range(num_cameras)), fake_render, point_state=light) == []
  fields = ("prune_cost", "split_score", "max_scale_px", "points_in_view", "visibility")
  # the DEFAULT exchange (dense per-camera blocks, no counts): blocks packed, gathered and replayed; the rebuild kernel
  # that consumes the colour-gradient part is HIP-only and checked on the GPU (tests/test_gpu_render.py)
  from splat_trainer_amd.densify import dp_replay
  dd = CameraShardedStep(params, world, rank)
  loc = [dict(camera=j, idx=d["idx"], split_score=d["split_score"], prune_cost=d["prune_cost"],
              screen_scale=torch.stack([d["screen_scale_max"], 0.5 * d["screen_scale_max"]], dim=1))
         for j, d in ((j, _camera_stats(j, n)) for j in mine)]
  blocks = dd.all_gather_blocks(dd.pack_camera_blocks(num_cameras, loc, col.items))
  replayed = PointState.new_zeros(n, "cpu")
  for _ in range(2):
    dp_replay(replayed, blocks, dd.camera_slots(num_cameras, "cpu"), n)
  # the DEFAULT exchange since round 4 (sharded replay), end to end through CameraShardedStep.run on gloo: gradient
  # all-reduce, MAX all-reduce of the screen scale, all-gather of the colour-factor blocks, all-to-all of the score slices,
  # in-order replay on the owned slice, all-gather of the slice state (the SH rebuild kernel itself is HIP-only)
  ds = CameraShardedStep(params, world, rank)

  def fake_render_dp(j, cam, grad_out, collector):
    d = _camera_stats(j, n)
    if grad_out.geometry_uninitialized:                 # the first backward pass of a batch writes every row
      for t in (grad_out.position, grad_out.log_scaling, grad_out.rotation, grad_out.alpha_logit):
        t.zero_()
      grad_out.geometry_uninitialized = False
    grad_out.position[d["idx"]] += float(j + 1)
    if collector.on_rows is not None:                   # what the fused node's backward pass does between its two halves:
      rows = torch.zeros(d["idx"].shape[0], 16)         # the packed gradient rows carry the colour gradient in columns 8..10
      rows[:, 8:11] = d["dcol"]
      collector.on_rows(d["idx"], rows, d["cam"])       # -> factor block packed, and (last local camera) the gather started
    collector.items.append((d["idx"], d["dcol"], d["cam"]))
    pts = sta.RenderedPoints(idx=d["idx"], depths=torch.zeros(d["idx"].shape[0], 1), opacity=torch.zeros(d["idx"].shape[0]),
                             screen_scale=torch.stack([d["screen_scale_max"], 0.5 * d["screen_scale_max"]], dim=1),
                             visibility=d["visibility"], prune_cost=d["prune_cost"], split_score=d["split_score"])
    return sta.Rendering(image=None, camera=None, points=pts)

  sharded = PointState.new_zeros(n, "cpu")
  for _ in range(2):
    assert ds.run(list(range(num_cameras)), fake_render_dp, point_state=sharded) == []
  torch.save(dict(stats=got, masks=masks, state={f: getattr(state, f) for f in fields}, dense=dense, packed=packed,
                  counts=counts, light={f: getattr(light, f) for f in fields}, visible=dp.visible.clone(),
                  blocks=blocks, replayed={f: getattr(replayed, f) for f in fields},
                  sharded={f: getattr(sharded, f) for f in fields}, sharded_visible=ds.visible.clone(),
                  sharded_blocks=ds.last_blocks.clone(), sharded_dpos=ds.grads["position"].clone()),
             f"{out_path}.{rank}")
  dist.barrier()
  dist.destroy_process_group()


def _camera_stats(j, n):
  gen = torch.Generator().manual_seed(100 + j)
  m = n // 2 + 17 * j                                           # cameras see different numbers of points
  idx = torch.randperm(n, generator=gen)[:m].sort().values
  vis = torch.rand(m, generator=gen)
  vis[torch.rand(m, generator=gen) < 0.2] = 0.0
  return dict(camera=j, idx=idx, screen_scale_max=40 * torch.rand(m, generator=gen) + 1, visibility=vis,
              split_score=torch.rand(m, generator=gen) * (vis > 0), prune_cost=torch.rand(m, generator=gen) * (vis > 0),
              dcol=torch.randn(m, 3, generator=gen), cam=torch.randn(3, generator=gen))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("num_cameras", [1, 3, 4])
def test_two_rank_point_stats_and_packed_factors(tmp_path, num_cameras):
  """distributed.gather_point_stats over gloo (one packed all_gather_into_tensor + counts; with 1 or 3 cameras a rank
  holds an empty slot or no camera at all): every rank receives every camera's statistics unchanged and in camera order,
  the PointState replayed from them -- and the split/prune masks -- equal the sequential loop's bit for bit, and the
  packed colour-factor gather scatters to exactly the dense block."""
  from splat_trainer_amd.controller_math import PointState, find_split_prune_indexes
  from splat_trainer_amd.distributed import STAT_FIELDS, replay_point_stats
  world, port = 2, 33000 + (os.getpid() % 2000) + num_cameras
  out = str(tmp_path / "stats")
  mp.spawn(_stats_worker, args=(world, port, out, num_cameras), nprocs=world, join=True)
  r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
  n = 500
  seq = [_camera_stats(j, n) for j in range(num_cameras)]
  for r in (r0, r1):
    assert [d["camera"] for d in r["stats"]] == list(range(num_cameras))
    for d, want in zip(r["stats"], seq):
      assert torch.equal(d["idx"], want["idx"])
      for f in STAT_FIELDS:
        assert torch.equal(d[f], want[f]), f
    assert torch.equal(r["dense"], r["packed"])
  state = replay_point_stats(PointState.new_zeros(n, "cpu"), seq)            # the sequential loop (trainer.py:500-514)
  masks = find_split_prune_indexes(state, 0.2, 560, min_views=1, max_scale_px=50.0)
  for r in (r0, r1):
    for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view", "visibility"):
      assert torch.equal(r["state"][f], getattr(state, f)), f
    assert torch.equal(r["masks"][0], masks[0]) and torch.equal(r["masks"][1], masks[1])
  if num_cameras > 1:
    assert masks[0].any() and masks[1].any()
  # light exchange, two batches: everything that enters the masks is bit-identical to the sequential loop; the summed
  # visibility (logged only) is reduced in rank order instead of camera order
  twice = replay_point_stats(replay_point_stats(PointState.new_zeros(n, "cpu"), seq), seq)
  for r in (r0, r1):
    for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view"):
      assert torch.equal(r["light"][f], getattr(twice, f)), f
    assert torch.allclose(r["light"]["visibility"], twice.visibility, rtol=1e-6, atol=1e-7)
    one = replay_point_stats(PointState.new_zeros(n, "cpu"), seq)
    assert torch.allclose(r["visible"], one.visibility, rtol=1e-6, atol=1e-7)      # the scene's accumulator of one batch
    m2 = find_split_prune_indexes(PointState(**r["light"]), 0.2, 560, min_views=1, max_scale_px=50.0)
    want = find_split_prune_indexes(twice, 0.2, 560, min_views=1, max_scale_px=50.0)
    assert torch.equal(m2[0], want[0]) and torch.equal(m2[1], want[1])
  # the sharded exchange (default): everything that enters the masks bit-identical to the sequential loop on both ranks,
  # the gathered colour-factor blocks equal to the dense factor gather, the all-reduced gradient the sum over all cameras
  want_dpos = torch.zeros(n, 3)
  for j, d in enumerate(seq):
    want_dpos[d["idx"]] += float(j + 1)
  for r in (r0, r1):
    for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view"):
      assert torch.equal(r["sharded"][f], getattr(twice, f)), f
    assert torch.allclose(r["sharded"]["visibility"], twice.visibility, rtol=1e-6, atol=1e-7)
    assert torch.allclose(r["sharded_visible"], one.visibility, rtol=1e-6, atol=1e-7)
    assert torch.equal(r["sharded_blocks"].reshape(-1, n + 1, 3), r["dense"])
    assert torch.equal(r["sharded_dpos"], want_dpos)
  # dense per-camera blocks: identical on both ranks, the factor part equals the dense factor gather, and the replayed
  # order-dependent state equals the sequential loop's bit for bit
  for r in (r0, r1):
    b = r["blocks"]
    assert b.shape == (r["dense"].shape[0], 6 * n + 3)
    assert torch.equal(b[:, :3 * n + 3].reshape(-1, n + 1, 3), r["dense"])
    assert torch.equal(torch.nan_to_num(b, nan=-1.0), torch.nan_to_num(r0["blocks"], nan=-1.0))
    for f in ("prune_cost", "split_score", "max_scale_px"):
      assert torch.equal(r["replayed"][f], getattr(twice, f)), f


def _idle_rank_worker(rank, world, port, out_path):
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("gloo", rank=rank, world_size=world)
  import splat_trainer_amd as sta
  from splat_trainer_amd.distributed import CameraShardedStep
  n = 40
  params = [torch.zeros(n, d, requires_grad=True) for d in (3, 3, 4, 1)] + [torch.zeros(n, 3, 4, requires_grad=True)]
  dp = CameraShardedStep(params, world, rank, mode="all_reduce", with_stats=False)      # fused_grad_out: the default
  dp.bucket.flat.fill_(float("nan"))                        # what an uninitialised buffer may hold

  def fake_render(j, cam, grad_out, collector):
    # the renderer's first backward pass of a batch: writes every row of every buffer and clears both flags
    assert grad_out.geometry_uninitialized and grad_out.feature_uninitialized
    for k, name in enumerate(("position", "log_scaling", "rotation", "alpha_logit", "feature")):
      getattr(grad_out, name).fill_(float(k + 1))
    grad_out.geometry_uninitialized = grad_out.feature_uninitialized = False
    idx = torch.arange(n)
    z = torch.zeros(n)
    return sta.Rendering(image=None, camera=None, points=sta.RenderedPoints(
        idx=idx, depths=z[:, None], opacity=z, screen_scale=torch.ones(n, 2), visibility=torch.ones(n), prune_cost=z,
        split_score=z))

  dp.run([0], fake_render)                                  # ONE camera, two ranks: rank 1 renders nothing
  torch.save({k: v.clone() for k, v in dp.grads.items()}, f"{out_path}.{rank}")
  assert not dp.grad_out.geometry_uninitialized and not dp.grad_out.feature_uninitialized
  dist.barrier()
  dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rank_without_camera_contributes_zeros_to_every_gradient_slot(tmp_path):
  """CameraShardedStep(mode="all_reduce", fused_grad_out=True) with fewer cameras than ranks: the idle rank's buffers --
  all five slots, the feature gradient included -- were declared uninitialised and never written, so they must be
  zero-filled before the all-reduce adds them to the other ranks' gradients."""
  world, port = 2, 35000 + (os.getpid() % 2000)
  out = str(tmp_path / "idle")
  mp.spawn(_idle_rank_worker, args=(world, port, out), nprocs=world, join=True)
  r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
  for k, name in enumerate(("position", "log_scaling", "rotation", "alpha_logit", "feature")):
    assert torch.all(r0[name] == float(k + 1)), name
    assert torch.equal(r0[name], r1[name]), name
