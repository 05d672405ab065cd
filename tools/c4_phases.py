"""Phase table of the c4 harness (3M Gaussians, 1080p, SH3, 8 cameras per iteration): every phase of an iteration timed
with a synchronize on both sides (so nothing overlaps; the sum exceeds the pipelined iteration), next to the pipelined
iteration itself.  Run under `rocprofv3 --kernel-trace --stats` for the kernel table of the same loop.
    python tools/c4_phases.py [n_points] [cameras]"""
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import splat_trainer_amd as sta
from splat_trainer_amd import synthetic, harness
from splat_trainer_amd.harness import MiniTrainer
from splat_trainer_amd.loss import clamped_mse_loss
from splat_trainer_amd.optim import point_basis_rows
from splat_trainer_amd.renderer import render_gaussians

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
ncam = int(sys.argv[2]) if len(sys.argv) > 2 else 8
w, h = 1920, 1080
g, cams = synthetic.scene_b(n, w, h, sh_degree=3, seed=1, num_cameras=8)
g = g.to("cuda")
cams = [c.to("cuda") for c in cams[:ncam]]
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
targets = [torch.full((h, w, 3), 0.5, device="cuda") for _ in cams]
tr = MiniTrainer(g, cams, targets, cfg, lr=1e-3, densify_every=25, target_points=int(1.1 * n), total_steps=100, seed=0)
for _ in range(3):
  tr.training_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
  tr.training_step()
torch.cuda.synchronize()
print(f"pipelined: {(time.perf_counter() - t0) / K * 1e3:.2f} ms per {ncam}-camera iteration", flush=True)


class Phase:
  acc = {}

  def __init__(self, name):
    self.name = name

  def __enter__(self):
    torch.cuda.synchronize()
    self.t = time.perf_counter()

  def __exit__(self, *a):
    torch.cuda.synchronize()
    Phase.acc[self.name] = Phase.acc.get(self.name, 0.0) + time.perf_counter() - self.t


R = 5
for _ in range(R):
  with Phase("grad_target"):
    go = tr._grad_target()
  for cam, target in zip(tr.cameras, tr.targets):
    with torch.enable_grad():
      with Phase("render forward"):
        r = render_gaussians(tr.scene(), cam, tr.config, use_sh=True, grad_out=go)
      with Phase("loss forward"):
        loss = clamped_mse_loss(r.image, target)
      with Phase("backward"):
        loss.backward()
    with torch.no_grad():
      with Phase("add_rendering"):
        tr.state.add_rendering(r, visible_sum=tr.points.visible)
    with Phase("loss.item"):
      float(loss.item())
    del r, loss
  pts = tr.points
  with torch.no_grad():
    with Phase("opt: nonzero"):
      vis_idx = pts.visible.nonzero().squeeze(1)
    with Phase("opt: basis"):
      basis = point_basis_rows(pts.log_scaling, pts.rotation, vis_idx)
    with Phase("opt: step"):
      pts.step(visibility=pts.visible[vis_idx], indexes=vis_idx, basis=basis)
    with Phase("opt: normalize+clamp"):
      pts.rotation.data = F.normalize(pts.rotation.data, dim=1)
      pts.log_scaling.data.clamp_(min=-8, max=8)
    with Phase("opt: zero"):
      pts.visible.zero_()
      pts.zero_grad()
total = 0.0
for k, v in Phase.acc.items():
  print(f"  {k:24s} {v / R * 1e3:8.3f} ms per iteration")
  total += v / R
print(f"  {'sum (serialised)':24s} {total * 1e3:8.3f} ms")

tr.step_idx = 25
for rep in range(2):
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  tr.densify_and_prune()
  torch.cuda.synchronize()
  print(f"densify_and_prune round {rep}: {(time.perf_counter() - t0) * 1e3:.1f} ms, N = {tr.num_points}")
  for _ in range(2):
    tr.training_step()
