#!/usr/bin/env python
"""Benchmark of the rasterizer hot path: fwd+bwd Gaussians/s (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic input: every rank renders ITS camera of the
batch (project -> SH colour -> tile bin/sort -> composite), takes the MSE loss of the clamped image against a constant
image (fused pixel-loss kernels, loss.py), back-propagates to the Gaussian parameters, updates the controller's point
statistics for the camera (PointState.add_rendering, the reference's per-camera consumer, trainer.py:514), and at N > 1
the parameter gradients (+ the per-point visibility accumulator) and the per-camera statistics are exchanged over RCCL
(distributed.CameraShardedStep) and the statistics of ALL cameras replayed in camera order on every rank.  Parameters are replicated; per-GPU
work is fixed as N grows (weak scaling).  value = Gaussians x cameras / second over the whole job, inputs
resident in HBM when the timed region starts.

Workloads:  c2 = Scene A, 500k Gaussians, 1920x1080, SH deg 3 (default, BASELINE configs[1]);
c3 = Scene B, 3M Gaussians, 1080p, SH 3, 8 orbit cameras (the default at --gpus > 1);  c1 = Scene A 10k / 256^2 / SH0;
c5 = Scene B 10M / 4K;  c5culled = the same from an orbit of radius 1.2 (a third of the points survive the frustum cull).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0     # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 TB/s achievable)
FP32_VECTOR_PEAK_TFLOPS = 157.3   # MI355X spec, packed fp32 FMA on every lane (MI355X_MICROARCH.md)
# Issue cost of one wave64 VALU instruction per SIMD at >= 6 waves per SIMD, nanoseconds of wall time, measured on the
# MI355X with tools/valu_rate.hip (profiles/r02_valu_issue_costs.txt): the chip lowers its clock under a full VALU load,
# so the costs are quoted in time, not cycles.  plain = v_fma/v_mul/v_add/v_cmp/v_cndmask, packed = v_pk_{fma,mul,add}_f32,
# trans = v_exp/v_rcp/v_sqrt.
VALU_ISSUE_NS = {"valu": 1.35, "valu_packed": 1.95, "valu_trans": 4.1}

WORKLOADS = {
    "c1": dict(scene="A", n=10_000, w=256, h=256, sh=0),
    "c2": dict(scene="A", n=500_000, w=1920, h=1080, sh=3),
    "c3": dict(scene="B", n=3_000_000, w=1920, h=1080, sh=3),
    "c5": dict(scene="B", n=10_000_000, w=3840, h=2160, sh=3),
    # c5's culled variant: orbit radius 1.2 around the unit ball, the frustum cull removes ~2/3
    "c5culled": dict(scene="B", n=10_000_000, w=3840, h=2160, sh=3, radius=1.2),
}


def yaw(T: torch.Tensor, deg: float) -> torch.Tensor:
  """World->camera with the camera turned by ``deg`` about its y axis (per-rank cameras for Scene A)."""
  a = math.radians(deg)
  R = torch.tensor([[math.cos(a), 0., math.sin(a), 0.], [0., 1., 0., 0.], [-math.sin(a), 0., math.cos(a), 0.],
                    [0., 0., 0., 1.]])
  return R @ T


def make_workload(name: str, world: int):
  import splat_trainer_amd as sta
  from splat_trainer_amd import synthetic
  w = WORKLOADS[name]
  if w["scene"] == "A":
    g, cam = synthetic.scene_a(w["n"], w["w"], w["h"], sh_degree=w["sh"], seed=0)
    cams = [sta.CameraParams(yaw(cam.T_camera_world, 1.5 * k), cam.projection, cam.image_size, cam.near_plane,
                             cam.far_plane) for k in range(max(world, 1))]
  else:
    g, cams = synthetic.scene_b(w["n"], w["w"], w["h"], sh_degree=w["sh"], seed=1, num_cameras=max(world, 8),
                                radius=w.get("radius", 3.0))
  return g, cams, w


def measure_copy_bandwidth(dev, mib: int = 1024, reps: int = 5) -> float:
  """Device-to-device copy bandwidth (GB/s, read + write bytes) of this GPU: the practical HBM ceiling next to the
  8 TB/s datasheet figure."""
  n = mib * (1 << 20) // 4
  src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
  dst = torch.empty_like(src)
  dst.copy_(src)
  torch.cuda.synchronize()
  best = float("inf")
  for _ in range(reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    dst.copy_(src)
    e1.record()
    e1.synchronize()
    best = min(best, e0.elapsed_time(e1))
  del src, dst
  return 2.0 * n * 4 / (best * 1e-3) / 1e9


def newest_profile(pattern: str):
  import glob
  hits = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
  return hits[-1] if hits else None


def valu_roofline(kernel_key: str, isa_key: str, pmc: dict, ms: float, pairs: int):
  """VALU-issue roofline of one composite kernel: wave64 VALU instructions per launch (rocprofv3 SQ_INSTS_VALU, kept in
  profiles/) priced with the measured issue costs in the static packed / plain / transcendental proportions of the
  kernel's per-pair loop (tools/isa_stats.py), against the measured launch time; plus the fp32 rate those instructions
  amount to against the 157.3 TFLOP/s vector peak (2 flops per fma lane, packed ops count twice)."""
  try:
    insts = float(pmc["kernels"][kernel_key]["SQ_INSTS_VALU"])
    mix = pmc["isa"][isa_key]["static_loop_mix"]
  except Exception:   # noqa: BLE001
    return None
  tot = sum(mix.get(k, 0) for k in VALU_ISSUE_NS)
  if not tot or not ms or ms != ms:
    return None
  ns = sum(mix.get(k, 0) * c for k, c in VALU_ISSUE_NS.items()) / tot
  bound_ms = insts * ns * 1e-9 / 1024 * 1e3                      # 1024 SIMDs
  lane_flops = insts * 64 * 2 * (1.0 + mix.get("valu_packed", 0) / tot)
  return {"valu_insts_per_launch": insts, "valu_insts_per_pair": insts / max(pairs, 1), "mean_issue_ns_per_inst": ns,
          "issue_bound_ms": bound_ms, "frac_of_issue_bound": bound_ms / ms,
          "fp32_tflops_issued": lane_flops / (ms * 1e-3) / 1e12, "fp32_vector_peak_tflops": FP32_VECTOR_PEAK_TFLOPS,
          "frac_of_fp32_vector_peak": lane_flops / (ms * 1e-3) / 1e12 / FP32_VECTOR_PEAK_TFLOPS,
          "mix_source": "tools/isa_stats.py static loop mix + tools/valu_rate.hip issue costs (profiles/r02_valu_issue_costs.txt)"}


def cpu_baseline(g, cam, cfg, budget_s: float = 25.0, chunk_tiles: int = 1024):
  """The CPU PyTorch path (oracle, autograd) on the host cores: the SAME workload -- cull + projection + SH for all
  Gaussians and composite forward+backward over the image's tiles, MSE loss -- run tile chunk by tile chunk until
  the whole image is done or ``budget_s`` seconds of compositing have been spent; in the latter case the remaining
  tiles are extrapolated from the measured per-tile time (the sample is stated in the result)."""
  from oracle import torch_oracle as oracle        # checker / baseline only
  try:
    cores = len(os.sched_getaffinity(0))
  except AttributeError:
    cores = os.cpu_count() or 1
  cores = max(1, min(cores, 16))          # the GPU box grants a 16-CPU share per GPU; more threads only thrash
  torch.set_num_threads(cores)
  W, H = cam.image_size
  tiles_x, tiles_y = (W + 15) // 16, (H + 15) // 16
  n_tiles = tiles_x * tiles_y
  gen = torch.Generator().manual_seed(0)
  perm = torch.randperm(n_tiles, generator=gen)
  leaves = [t.clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]

  t0 = time.perf_counter()
  idx = oracle.frustum_cull(leaves[0], cam.T_camera_world, cam.projection, cam.image_size, cam.near_plane,
                            cam.far_plane, cfg.margin_tiles * cfg.tile_size)
  g2d, depth, _ = oracle.project(leaves[0], leaves[1], leaves[2], leaves[3], idx, cam.T_camera_world,
                                 cam.projection, cfg)
  feats = oracle.evaluate_sh_at(leaves[4], leaves[0], idx, cam.camera_position)
  g2d_d, feats_d = g2d.detach().requires_grad_(True), feats.detach().requires_grad_(True)
  lists = oracle._tile_lists(g2d_d, depth.detach(), cam.image_size, cfg)
  t1 = time.perf_counter()
  done, t_tiles = 0, 0.0
  while done < n_tiles and t_tiles < budget_s:
    tiles = perm[done:done + chunk_tiles]
    tc = time.perf_counter()
    out = oracle.rasterize(g2d_d, depth.detach(), feats_d, cam.image_size, cfg, tiles=tiles, lists=lists)
    loss = ((out.image.clamp(0, 1) - 0.5) ** 2).sum() / (W * H * feats.shape[1])      # this chunk's share of the MSE
    loss.backward()
    t_tiles += time.perf_counter() - tc
    done += tiles.numel()
  t2 = time.perf_counter()
  torch.autograd.backward([g2d, feats], [g2d_d.grad, feats_d.grad])
  t3 = time.perf_counter()
  t_geom = (t1 - t0) + (t3 - t2)
  est = t_geom + t_tiles * (n_tiles / done)
  n = g.position.shape[0]
  how = "full image" if done >= n_tiles else f"{done} of {n_tiles} tiles, remaining tiles extrapolated x{n_tiles / done:.2f}"
  return dict(value=n / est, unit="Gaussians/s", cores=cores, kind="port",
              sample=(f"oracle/torch_oracle.py (pure-PyTorch CPU, autograd): cull+project+SH+tile lists fwd+bwd on all "
                      f"{n} Gaussians ({t_geom:.2f}s) + composite fwd+bwd on {how} ({t_tiles:.2f}s); "
                      f"full pass {est:.1f}s"))


def measure_scale_workload(name: str, dev, steps: int = 10, warm: int = 12):
  """One-GPU, one-camera step of the workload the multi-GPU runs use (c3): the baseline a scaling curve over
  ``bench.py --gpus N`` needs (those runs render camera k of the same scene on GPU k; per-GPU work is this step)."""
  import splat_trainer_amd as sta
  from splat_trainer_amd.controller_math import PointState
  from splat_trainer_amd.distributed import CameraShardedStep
  g, cams, w = make_workload(name, 1)
  cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True, blur_cov=0.3, antialias=False)
  params = [t.to(dev).requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  scene = sta.Gaussians3D(position=params[0], rotation=params[2], log_scaling=params[1], alpha_logit=params[3], feature=params[4])
  cam = cams[0].to(dev)
  target = torch.full((w["h"], w["w"], 3), 0.5, device=dev)
  dp = CameraShardedStep(params, 1, 0)
  state = PointState.new_zeros(params[0].shape[0], dev)
  last = {}

  def render_backward(j, c, grad_out, collector):
    with torch.enable_grad():
      r = sta.render_gaussians(scene, c, cfg, use_sh=True, grad_out=grad_out, sh_collector=collector)
      sta.clamped_mse_loss(r.image, target).backward()
    last["r"] = r
    return r

  for _ in range(warm):
    dp.run([cam], render_backward, point_state=state)
  marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  marks[0].record()
  for i in range(steps):
    dp.run([cam], render_backward, point_state=state)
    marks[i + 1].record()
  torch.cuda.synchronize()
  elapsed = time.perf_counter() - t0
  per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
  N = params[0].shape[0]
  return {"workload": f"{name}: scene {w['scene']}, {N} Gaussians, {w['w']}x{w['h']}, SH deg {w['sh']}, camera 0 of the 8-camera "
                      f"orbit, MSE loss (what one rank of `bench.py --gpus N` does per step, without the exchange)",
          "steps": steps, "ms_per_step": 1e3 * elapsed / steps, "ms_per_step_median": per[len(per) // 2],
          "value": N * steps / elapsed, "unit": "Gaussians/s", "tile_overlaps": int(last["r"].num_overlaps),
          "visible": int(last["r"].points.idx.shape[0])}


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=20)
  ap.add_argument("--warmup", type=int, default=5)
  ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                  help="default: c2 on one GPU (BASELINE configs[1], the configuration the metric is quoted on); c3 at "
                       "--gpus > 1 (the 3M-Gaussian scene north_star quotes the scaling target on, camera k on GPU k)")
  ap.add_argument("--loss", default="mse", choices=["mse", "ref"],
                  help="mse (default): clamped MSE against a constant image (SURVEY.md section 8d); ref: the reference's loss "
                       "mix, L1 + MSE + 4-level fused_ssim on 2x average-pooled pyramids (trainer.py:448-488, loss.reference_loss)")
  ap.add_argument("--no-scale-workload", action="store_true",
                  help="skip the nested one-GPU measurement of the multi-GPU workload (c3) that the N = 1 line carries")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--form", default="one_call", choices=["one_call", "three_call"],
                  help="one_call (default): render_gaussians(use_sh=True), the fused node behind the native frame driver; "
                       "three_call: project_to_image -> evaluate_sh_at -> render_projected, the call shape of "
                       "MLPScene.render (mlp_scene.py:410-427) with the SH evaluation standing in for the colour MLP")
  ap.add_argument("--settle", type=int, default=30,
                  help="untimed steps before the timed region, INCLUDING --warmup (clocks and allocator settle; 0 = only "
                       "the --warmup steps; profiles/r03_settle_trace.txt shows the per-step times either way)")
  ap.add_argument("--collective", default="sh_factor", choices=["reduce_scatter", "all_reduce", "sh_factor"],
                  help="gradient exchange at WORLD_SIZE > 1 (nothing is communicated on one GPU).  sh_factor (default, "
                       "distributed.DEFAULT_COLLECTIVE): all_reduce only the geometry gradients and all_gather the "
                       "per-camera colour gradients (3 floats/splat), rebuilding the SH coefficient gradient (48 "
                       "floats/splat) on every rank; all_reduce: one fused all_reduce of the whole flat gradient buffer; "
                       "reduce_scatter: reduce_scatter + all_gather of the same buffer")
  ap.add_argument("--check-collective", action="store_true",
                  help="after the timed run, do one step with all_reduce and one with --collective and report the "
                       "largest relative difference of the summed gradients (rehearsal aid; not timed)")
  ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                  help="nccl = RCCL over xGMI (default); gloo only for rehearsing the multi-rank path on one GPU")
  args = ap.parse_args()

  if args.gpus < 1:
    raise SystemExit("--gpus must be >= 1")
  if args.workload is None:
    args.workload = "c2" if args.gpus == 1 else "c3"
  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # Started without a launcher: start the N ranks ourselves (one process per GPU over RCCL) as a CHILD process, before
    # anything in this process has touched the GPU (device_count() does not initialise it), and exit with its code.
    # A run asked for N GPUs never reports fewer.
    have = torch.cuda.device_count()
    if have < args.gpus:
      raise SystemExit(f"--gpus {args.gpus} but only {have} GPU(s) are visible: refusing to measure fewer than asked")
    import socket
    import subprocess
    with socket.socket() as s:
      s.bind(("127.0.0.1", 0))
      port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd))
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  if world != args.gpus:
    raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count must match --gpus")
  if not torch.cuda.is_available():
    raise SystemExit("bench.py needs a GPU: the rasterizer path has no CPU fallback")
  if os.environ.get("BENCH_SHARE_GPU") == "1":      # rehearsal only: every rank on device 0 (needs --backend gloo)
    local_rank = 0
  torch.cuda.set_device(local_rank)
  dev = torch.device("cuda", local_rank)
  world_seen = 1
  if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.backend == "nccl":
      dist.init_process_group("nccl", device_id=dev)
    else:
      dist.init_process_group("gloo")
    world_seen = dist.get_world_size()          # what the process group (RCCL) actually spans
    if world_seen != args.gpus:
      raise SystemExit(f"--gpus {args.gpus} but the process group has {world_seen} ranks")

  import splat_trainer_amd as sta
  from splat_trainer_amd import renderer
  from splat_trainer_amd.controller_math import PointState
  from splat_trainer_amd.distributed import CameraShardedStep

  g, cams, w = make_workload(args.workload, world)
  cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True, blur_cov=0.3, antialias=False)
  N = g.position.shape[0]
  params = [t.to(dev).requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  position, log_scaling, rotation, alpha_logit, feature = params
  factor_mode = args.collective == "sh_factor"
  # camera j of the batch -> rank j mod world; gradients accumulate straight into the flat collective buffer (the
  # reference accumulates into .grad over the cameras of a batch, trainer.py:500-514); + the per-point `visible`
  # accumulator (mlp_scene.py:244); the per-camera controller statistics of ALL cameras come back in camera order
  dp = CameraShardedStep(params, world, rank, mode=args.collective, position_term_local=args.form != "three_call")
  bucket = dp.bucket
  batch = [c.to(dev) for c in cams[:max(world, 1)]]                       # one camera per rank per step
  target_image = torch.full((w["h"], w["w"], 3), 0.5, device=dev)
  if args.loss == "ref":       # SSIM against a constant image is degenerate (zero variance): a smooth, seeded pattern instead
    yy, xx = torch.meshgrid(torch.linspace(0, 1, w["h"], device=dev), torch.linspace(0, 1, w["w"], device=dev), indexing="ij")
    target_image = (0.5 + 0.25 * torch.stack([torch.sin(9 * xx + 3 * yy), torch.cos(7 * yy - 2 * xx), torch.sin(5 * (xx + yy))],
                                             dim=-1)).contiguous()
  scene = sta.Gaussians3D(position=position, rotation=rotation, log_scaling=log_scaling, alpha_logit=alpha_logit,
                          feature=feature)
  point_state = PointState.new_zeros(N, dev)                               # the controller's state (point_state.py:22-32)
  last = {}

  def render_backward(j, cam, grad_out, collector):
    with torch.enable_grad():
      if args.form == "three_call":
        prefetch = {}
        g2d, depth, idx = sta.project_to_image(scene, cam, cfg, grad_out=grad_out, prefetch=prefetch)
        sh_out = collector if collector is not None else (grad_out.feature, grad_out.position, grad_out)
        feats = sta.evaluate_sh_at(feature, position, idx, cam.camera_position, grad_out=sh_out)
        r = sta.render_projected(idx, g2d, feats, depth, cam, cfg, _depth_order=prefetch.get("depth_order"))
      else:
        r = sta.render_gaussians(scene, cam, cfg, use_sh=True, grad_out=grad_out, sh_collector=collector)
      if args.loss == "ref":
        loss = sta.reference_loss(r.image, target_image)        # trainer.py:448-488: L1 + MSE + multi-scale SSIM
      else:
        loss = sta.clamped_mse_loss(r.image, target_image)      # = F.mse_loss(image.clamp(0, 1), target), trainer.py:472-475
      loss.backward()
    last["r"] = r
    return r

  def step():
    # controller.add_rendering for every camera of the batch (trainer.py:514), in camera order on every rank
    dp.run(batch, render_backward, point_state=point_state)

  def sync():
    torch.cuda.synchronize()
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  copy_gbs = measure_copy_bandwidth(dev) if rank == 0 else None
  import gc
  gc.collect()
  gc.disable()                       # no collector pauses inside the timed region (re-enabled right after it)
  # Untimed preparation: the device needs ~30 steps after an idle period before its clocks and the allocator reach a
  # steady state (per-step times fall monotonically from 1.30 to 1.12 ms over the first 25 steps on c2, BENCH_TRACE_STEPS=1),
  # so the W warm-up steps the caller asked for are preceded by enough extra untimed ones to make 30 in total.
  prewarm = max(0, args.settle - args.warmup)
  for _ in range(prewarm + args.warmup):
    step()
  timer = renderer.KernelTimer()
  timer.reserve(4 * args.steps * max(len(batch) // max(world, 1), 1) + 8)   # events exist before the timed region starts
  marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
  for m in marks:
    m.record()
  sync()
  if os.environ.get("BENCH_NO_KERNEL_TIMER") != "1":      # (diagnostic: the cost of the roofline leg's events on a host-bound step)
    renderer.KERNEL_TIMER = timer
  t0 = time.perf_counter()
  marks[0].record()
  for i in range(args.steps):
    step()
    marks[i + 1].record()
  sync()
  elapsed = time.perf_counter() - t0
  gc.enable()
  renderer.KERNEL_TIMER = None
  step_trace = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
  if os.environ.get("BENCH_TRACE_STEPS") == "1" and rank == 0:
    print("per-step ms:", " ".join(f"{t:.3f}" for t in step_trace), file=sys.stderr, flush=True)
  per_step_ms = sorted(step_trace)
  median_ms = per_step_ms[len(per_step_ms) // 2] if len(per_step_ms) % 2 else \
      0.5 * (per_step_ms[len(per_step_ms) // 2 - 1] + per_step_ms[len(per_step_ms) // 2])
  if world > 1:
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

  check = None
  if args.check_collective and factor_mode and world > 1:
    step()
    got = [v.clone() for v in dp.grads.values()]
    ref = CameraShardedStep(params, world, rank, mode="all_reduce", with_stats=False)
    ref.run(batch, render_backward)
    check = max(float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(got, ref.grads.values()))
    bucket.attach()
    # controller statistics: one exchanged batch against the reference's own loop over ALL cameras on this rank
    exchanged, sequential = PointState.new_zeros(N, dev), PointState.new_zeros(N, dev)
    dp.run(batch, render_backward, point_state=exchanged)
    solo = CameraShardedStep(params, 1, 0, with_stats=False)
    solo.run(batch, render_backward, point_state=sequential)
    bucket.attach()
    stats_check = {f: bool(torch.equal(getattr(exchanged, f), getattr(sequential, f)))
                   for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view")}
    stats_check["visibility_rel_err"] = float((exchanged.visibility - sequential.visibility).norm() /
                                              sequential.visibility.norm().clamp_min(1e-30))

  r = last["r"]
  M, O = int(r.points.idx.shape[0]), int(r.num_overlaps)
  P = w["w"] * w["h"]
  ksum = timer.summary()
  n_bwd, ms_bwd = ksum.get("composite_backward", (0, float("nan")))
  n_fwd, ms_fwd = ksum.get("composite_forward", (0, float("nan")))
  K = (w["sh"] + 1) ** 2
  B_par = 44 + 12 * K
  alg_bytes_bwd = 40 * O + 32 * P + 36 * M                 # (S+I) O + 32 P + S M
  alg_bytes_fwd = 40 * O + 20 * P                          # (S+I) O + 20 P
  alg_bytes_step = 12 * N + (3 * B_par + 104) * M + 104 * O + 52 * P
  # HBM traffic / VALU counters of the kernels cannot be collected from inside this process; the values measured with
  # rocprofv3 on the same command (separate --pmc passes) are kept under profiles/ and quoted when the workload matches
  pmc, pmc_path = None, newest_profile(f"r*_pmc_{args.workload}.json")
  try:
    pmc = json.load(open(pmc_path)) if pmc_path else None
  except Exception:   # noqa: BLE001
    pmc = None
  pmc_name = os.path.relpath(pmc_path, ROOT) if pmc else None

  def traffic_of(key):
    try:
      return pmc["kernels"][key]["hbm_bytes_per_launch"]
    except Exception:   # noqa: BLE001
      return None

  def line(alg_bytes, ms, traffic=None):
    gbs = alg_bytes / (ms * 1e-3) / 1e9 if ms == ms and ms > 0 else float("nan")
    return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "frac_of_measured_copy_bw": (gbs / copy_gbs) if copy_gbs else None, "traffic": traffic,
            "algorithmic_bytes": alg_bytes, "avg_ms": ms}

  # Useful arithmetic of the backward composite: contributing (pixel, splat) pairs x the fp32 operations one of them costs
  # (K7's per-pixel body: ~45, an fma counted as 2), against the vector peak -- the honest counterpart of the issue-based
  # figure below, which also counts the lanes masked to alpha = 0.  The pair count is the summed area of the ellipses
  # {q <= min(9, 2 ln(255 opacity))} of the frame's splats: an upper estimate (it ignores saturated pixels and the image
  # border).  Its ratio to the algorithmic bytes is the kernel's arithmetic intensity; above the chip's ridge
  # (peak flops / peak bytes) the HBM roofline is not the binding one even at perfect efficiency.
  with torch.no_grad():
    g2d_, _, _ = sta.project_to_image(scene, batch[0], cfg)
    qeff = torch.minimum(torch.full_like(g2d_[:, 5], cfg.gaussian_scale ** 2),
                         2.0 * torch.log((g2d_[:, 5] / cfg.alpha_threshold).clamp_min(1e-20))).clamp_min(0.0)
    px_pairs = float((math.pi * qeff / torch.sqrt((g2d_[:, 2] * g2d_[:, 4] - g2d_[:, 3] ** 2).clamp_min(1e-30))).sum())
  FLOP_PER_PX_PAIR_BWD = 45.0
  ridge = FP32_VECTOR_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)

  def useful(ms):
    tf = px_pairs * FLOP_PER_PX_PAIR_BWD / (ms * 1e-3) / 1e12 if ms == ms and ms > 0 else float("nan")
    ai = px_pairs * FLOP_PER_PX_PAIR_BWD / max(alg_bytes_bwd, 1)
    return {"contributing_pixel_pairs_estimate": px_pairs, "flop_per_pixel_pair": FLOP_PER_PX_PAIR_BWD,
            "useful_fp32_tflops": tf, "useful_fp32_frac": tf / FP32_VECTOR_PEAK_TFLOPS,
            "arithmetic_intensity_flop_per_byte": ai, "ridge_flop_per_byte": ridge,
            "binding_roofline_at_perfect_efficiency": "fp32 vector" if ai > ridge else "hbm"}

  cameras_per_step = world
  value = N * cameras_per_step * args.steps / elapsed
  step_ms = 1e3 * elapsed / args.steps

  if rank == 0:
    k7 = line(alg_bytes_bwd, ms_bwd, traffic_of("K7"))
    k7.update({"kernel": "composite_bwd_kernel<3> (K7 alpha-composite backward)", "launches_timed": n_bwd,
               "algorithmic_bytes_per_launch": alg_bytes_bwd, "avg_launch_ms": ms_bwd,
               "traffic_source": f"{pmc_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE doubled as the "
                                 f"guide prescribes for gfx950 -- an upper bound for this kernel's mix of scalar and 4-byte loads)"
               if traffic_of("K7") else None,
               "hbm_copy_gbs_measured": copy_gbs,
               # K7 / K6 are bound by VALU issue, not by HBM: the second roofline states that bound
               "valu": valu_roofline("K7", "K7_bwd_C3", pmc, ms_bwd, O) if pmc else None,
               "useful_fp32": useful(ms_bwd),
               "kernels": {
                   "K6 composite_fwd_kernel<3,vis> (alpha-composite forward, incl. heavy-tile passes)":
                       dict(line(alg_bytes_fwd, ms_fwd, traffic_of("K6")),
                            valu=valu_roofline("K6", "K6_fwd_C3_vis", pmc, ms_fwd, O) if pmc else None),
                   "whole step (cull -> project -> SH -> bin -> sort -> composite -> loss -> backward)":
                       line(alg_bytes_step, median_ms)}})
    out = {
        "metric": "fwd+bwd Gaussians/s", "value": value, "unit": "Gaussians/s", "n_gpus": world,
        "world_size": world_seen,
        "steps": args.steps, "warmup": args.warmup, "untimed_prewarm_steps": prewarm, "ms_per_step": step_ms,
        "ms_per_step_median": median_ms,
        "ms_per_step_min_max": [per_step_ms[0], per_step_ms[-1]],
        # the first timed step starts on an idle device right behind the barrier + synchronize: nothing of it overlaps a
        # previous step's tail, so it carries the host's whole enqueue time on top of the steady-state step
        "ms_first_step": step_trace[0],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: scene {w['scene']}, {N} Gaussians, {w['w']}x{w['h']}, SH deg {w['sh']}, "
                               f"1 camera per GPU per step, {'reference loss mix (L1 + MSE + 4-level SSIM)' if args.loss == 'ref' else 'MSE loss'}, "
                               f"compute_visibility+point_heuristic on"
                               + (", three-call form" if args.form == "three_call" else ""),
                   "gaussians": N, "visible": M, "tile_overlaps": O, "pixels": P, "cameras_per_step": cameras_per_step,
                   "parallelism": ("dp1 (one GPU: gradients accumulate in place, no collective)" if world == 1 else
                                   f"dp{world} (camera-sharded, sharded exchange, no host sync: SUM all_reduce of "
                                   f"{bucket.flat.numel() * 4 / 1e6:.0f} MB geometry grads + sums, MAX all_reduce of {N * 4 / 1e6:.0f} MB "
                                   f"screen scale, all_gather of {world} x {(3 * N + 3) * 4 / 1e6:.0f} MB colour-gradient factor blocks "
                                   f"(started inside the backward pass), all_to_all + all_gather of the controller scores by "
                                   f"point slices, {2 * 2 * ((N + world - 1) // world) * 4 / 1e6:.1f} MB per peer)" if factor_mode else
                                   f"dp{world} (camera-sharded, fused {args.collective} of {bucket.flat.numel() * 4 / 1e6:.0f} MB grads)"),
                   "parity": "parity unpinned by the reference (its rasterizer is an absent third-party package); "
                             "HIP vs this build's fp64 oracle is asserted by tests/ (-m gpu), observed errors in "
                             "profiles/r04_parity_observed.txt"},
        "roofline": k7,
    }
    if check is not None:
      out["config"]["collective_check_rel_err"] = check
      out["config"]["statistics_check_bit_identical"] = stats_check
    if world == 1 and args.workload == "c2" and not args.no_scale_workload:
      # the multi-GPU runs default to c3 (camera k on GPU k): its one-GPU, one-camera step, so a scaling curve has its baseline
      try:
        out["scale_workload"] = measure_scale_workload("c3", dev)
      except Exception as e:   # noqa: BLE001 -- never take the headline down with it
        out["scale_workload"] = {"error": f"{type(e).__name__}: {e}"}
    if world == 1 and not args.no_cpu_baseline:
      try:
        out["cpu_baseline"] = cpu_baseline(g, cams[0], cfg)
      except Exception as e:   # noqa: BLE001 -- the baseline must never take the GPU number down with it
        out["cpu_baseline"] = {"value": None, "unit": "Gaussians/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"failed: {type(e).__name__}: {e}"}
    print(json.dumps(out), flush=True)
  if world > 1:
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
  main()
