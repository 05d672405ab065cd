"""Tile-load imbalance: K6/K7 time on a clustered scene (a fraction of the splats packed into a small image region)
against the uniform scene of the same size, with and without heavy-tile list segmentation.  Without it one wave walks
one tile's list serially, so the heaviest tile bounds both kernels from below."""
import sys, math, torch
sys.path.insert(0, ".")
import splat_trainer_amd as sta
from splat_trainer_amd import synthetic, renderer
W, H, n = 1920, 1080, 500_000
import itertools
CFGS = {"segmented (automatic thresholds)": sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True),
        "one wave per tile": sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True, segment_pairs=0)}
for (frac, region), (cfg_name, cfg) in itertools.product([(0.0, 1.0), (0.3, 0.2), (0.5, 0.1), (0.5, 0.05)], CFGS.items()):
  g, cam = synthetic.scene_a(n, W, H, sh_degree=0, seed=0)
  k = int(frac * n)
  if k:
    gen = torch.Generator().manual_seed(1)
    fx = W / (2.0 * math.tan(math.radians(30.0)))
    z = g.position[:k, 2]
    u = (0.5 + region * (torch.rand(k, generator=gen) - 0.5)) * W
    v = (0.5 + region * (torch.rand(k, generator=gen) - 0.5)) * H
    g.position[:k, 0] = (u - W / 2) * z / fx
    g.position[:k, 1] = (v - H / 2) * z / fx
  g = g.to("cuda"); cam_d = cam.to("cuda")
  params = [t.requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
  scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])
  def step():
    with torch.enable_grad():
      r = sta.render_gaussians(scene, cam_d, cfg, use_sh=True)
      ((r.image - 0.5) ** 2).mean().backward()
    return r
  for _ in range(3): r = step()
  torch.cuda.synchronize()
  timer = renderer.KernelTimer(); renderer.KERNEL_TIMER = timer
  for _ in range(10): r = step()
  torch.cuda.synchronize(); renderer.KERNEL_TIMER = None
  ks = timer.summary()
  print(f"{frac:.0%} of the splats in the central {region:.0%} x {region:.0%} of the image, {cfg_name}: O {r.num_overlaps}  "
        f"K6 {ks['composite_forward'][1] * 1e3:.0f} us  K7 {ks['composite_backward'][1] * 1e3:.0f} us", flush=True)
