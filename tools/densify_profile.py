"""Phases of one densify / prune round of the c4 harness at 3M points (host wall time, synchronized per phase)."""
import hashlib
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import splat_trainer_amd as sta
from splat_trainer_amd import synthetic
from splat_trainer_amd.controller_math import PointState, find_split_prune_indexes
from splat_trainer_amd.harness import MiniTrainer, split_gaussians_uniform

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
w, h = 1920, 1080
g, cams = synthetic.scene_b(n, w, h, sh_degree=3, seed=1, num_cameras=8)
g = g.to("cuda")
cams = [c.to("cuda") for c in cams]
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
targets = [torch.full((h, w, 3), 0.5, device="cuda") for _ in cams]
tr = MiniTrainer(g, cams, targets, cfg, lr=1e-3, densify_every=25, target_points=int(1.1 * n), total_steps=100, seed=0)
for _ in range(2):
  tr.training_step()


def lap(name, t):
  torch.cuda.synchronize()
  print(f"  {name:28s} {(time.perf_counter() - t) * 1e3:8.1f} ms", flush=True)
  return time.perf_counter()


for rep, frac in enumerate((0.25, 0.5, 0.75)):
  print(f"round {rep}: N = {tr.num_points}, reserved {torch.cuda.memory_reserved() / 1e9:.1f} GB")
  with torch.no_grad():
    torch.cuda.synchronize()
    t = time.perf_counter()
    target = int(n * (1 + 0.1 * (rep + 1) / 3))
    split_mask, prune_mask = find_split_prune_indexes(tr.state, frac, target, tr.prune_rate, tr.min_views, tr.max_scale_px)
    t = lap("masks (radix select)", t)
    digest = hashlib.sha256(torch.cat([split_mask, prune_mask]).cpu().numpy().tobytes()).hexdigest()
    t = lap("digest (cat, .cpu, sha256)", t)
    keep_mask = ~(split_mask | prune_mask)
    split_idx = split_mask.nonzero().squeeze(1)
    t = lap("keep mask, split rows", t)
    splits = split_gaussians_uniform(tr.points[split_idx].detach(), k=2, random_axis=True, generator=tr.gen)
    t = lap("split_gaussians_uniform", t)
    tr.points = tr.points.keep_and_append(keep_mask, splits)
    t = lap("keep_and_append", t)
    tr.state = PointState.new_zeros(tr.num_points, tr.device)
    t = lap("new PointState", t)
  t = time.perf_counter()
  tr.training_step()
  t = lap("first training_step after", t)
  tr.training_step()
  t = lap("second training_step after", t)
