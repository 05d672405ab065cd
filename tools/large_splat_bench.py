"""What splats that have grown large cost the binning and reduction kernels: the c2 frame (500k splats, 1080p, SH3) with
G of its splats scaled up to fill most of the frame (x60; faint), kernel times from rocprofv3-free HIP events of the whole
step plus K6 / K7.  Run under rocprofv3 --kernel-trace --stats for the per-kernel table.
    python tools/large_splat_bench.py [G ...]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import splat_trainer_amd as sta
from splat_trainer_amd import renderer, synthetic

counts = [int(a) for a in sys.argv[1:]] or [0, 8, 64, 512]
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
for G in counts:
  g, cam = synthetic.scene_a(500_000, 1920, 1080, sh_degree=3, seed=0)
  gen = torch.Generator().manual_seed(7)
  pick = torch.randperm(500_000, generator=gen)[:G]
  ls, al = g.log_scaling.clone(), g.alpha_logit.clone()
  ls[pick] += 4.1
  al[pick] = -3.0
  g = sta.Gaussians3D(g.position, g.rotation, ls, al, g.feature).to("cuda")
  cam = cam.to("cuda")
  params = [t.requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])

  def step():
    for p in params:
      p.grad = None
    with torch.enable_grad():
      r = sta.render_gaussians(scene, cam, cfg, use_sh=True)
      ((r.image.clamp(0, 1) - 0.5) ** 2).mean().backward()
    return r

  for _ in range(5):
    r = step()
  torch.cuda.synchronize()
  timer = renderer.KernelTimer()
  renderer.KERNEL_TIMER = timer
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(10):
    r = step()
  e1.record()
  torch.cuda.synchronize()
  renderer.KERNEL_TIMER = None
  ks = timer.summary()
  print(f"{G:5d} large splats: O {r.num_overlaps:9d}  step {e0.elapsed_time(e1) / 10 * 1e3:7.0f} us  K6 {ks['composite_forward'][1] * 1e3:6.1f}  "
        f"K7 {ks['composite_backward'][1] * 1e3:6.1f}  rest {e0.elapsed_time(e1) / 10 * 1e3 - (ks['composite_forward'][1] + ks['composite_backward'][1]) * 1e3:7.0f} us", flush=True)
