"""Does the number of tiles relative to resident waves (tail effect) matter for K6/K7?  Same splat density, different
image sizes -> time per (tile, splat) pair."""
import sys, torch
sys.path.insert(0, ".")
import splat_trainer_amd as sta
from splat_trainer_amd import synthetic, renderer
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
for (W, H) in [(1920, 1088), (1536, 1024), (1024, 768), (2048, 1024), (2048, 1536), (3072, 2048)]:
  tiles = (W // 16) * (H // 16)
  n = int(500_000 * tiles / 8160)
  g, cam = synthetic.scene_a(n, W, H, sh_degree=0, seed=0)
  g = g.to("cuda"); cam = cam.to("cuda")
  params = [t.requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])
  def step():
    with torch.enable_grad():
      r = sta.render_gaussians(scene, cam, cfg, use_sh=True)
      ((r.image - 0.5) ** 2).mean().backward()
    return r
  for _ in range(3): r = step()
  torch.cuda.synchronize()
  timer = renderer.KernelTimer(); renderer.KERNEL_TIMER = timer
  for _ in range(10): r = step()
  torch.cuda.synchronize(); renderer.KERNEL_TIMER = None
  ks = timer.summary(); O = r.num_overlaps
  print(f"{W}x{H}: tiles {tiles} ({tiles / 1024:.2f}/SIMD)  N {n}  O {O}  K7 {ks['composite_backward'][1]*1e3:.0f} us = {ks['composite_backward'][1]*1e6/O*1e3:.3f} ps/pair... "
        f"K7 ns/pair {ks['composite_backward'][1]*1e6/O:.4f}  K6 ns/pair {ks['composite_forward'][1]*1e6/O:.4f}")
