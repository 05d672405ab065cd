"""Where an 8-camera training iteration of the c4 harness spends its time (3M Gaussians, 1080p, SH3): host wall time per
phase with a synchronize after each (so phases do not overlap; the sum exceeds the pipelined iteration).
    python tools/c4_profile.py [n_points]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import splat_trainer_amd as sta
from splat_trainer_amd import synthetic
from splat_trainer_amd.harness import MiniTrainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
w, h = 1920, 1080
g, cams = synthetic.scene_b(n, w, h, sh_degree=3, seed=1, num_cameras=8)
g = g.to("cuda")
cams = [c.to("cuda") for c in cams]
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
targets = [torch.full((h, w, 3), 0.5, device="cuda") for _ in cams]
tr = MiniTrainer(g, cams, targets, cfg, lr=1e-3, densify_every=25, target_points=int(1.1 * n), total_steps=100, seed=0)
for _ in range(3):
  tr.training_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
  tr.training_step()
torch.cuda.synchronize()
print(f"pipelined: {(time.perf_counter() - t0) / 5 * 1e3:.1f} ms per 8-camera iteration")

# one densify / prune round at this size (masks, digest, split, compaction, new controller state)
tr.step_idx = 25
for rep in range(2):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  tr.densify_and_prune()
  torch.cuda.synchronize()
  print(f"densify_and_prune round {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms, N = {tr.num_points}")
  for _ in range(2):
    tr.training_step()
