"""BASELINE config c4 at full size: 3M Gaussians, 1080p, 100-iteration training loop with controller densify/prune
(TargetController arithmetic, every 25 iterations), 8 cameras per iteration, sparse
visibility-aware LaProp step.  Prints the loop's trajectory -- ms per iteration, pairs per frame and points per window of
10 iterations (the scene changes while it trains, and with it the render's cost) -- and one JSON summary line.
    python tools/c4_train_loop.py [n_points] [cameras] [mse|ref]
Not the headline metric: a measured line for DESIGN.md."""
import json, sys, time, torch
sys.path.insert(0, ".")
import splat_trainer_amd as sta
from splat_trainer_amd import synthetic
from splat_trainer_amd.harness import MiniTrainer
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
ncam = int(sys.argv[2]) if len(sys.argv) > 2 else 8
loss_kind = sys.argv[3] if len(sys.argv) > 3 else "mse"
g, cams = synthetic.scene_b(n, 1920, 1080, sh_degree=3, seed=1, num_cameras=8)
dev = "cuda"
g = g.to(dev); cams = [c.to(dev) for c in cams[:ncam]]
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
targets = [torch.full((1080, 1920, 3), 0.5, device=dev) for _ in cams]
if loss_kind == "ref":      # SSIM against a constant image is degenerate: a smooth pattern per camera
  yy, xx = torch.meshgrid(torch.linspace(0, 1, 1080, device=dev), torch.linspace(0, 1, 1920, device=dev), indexing="ij")
  targets = [(0.5 + 0.25 * torch.stack([torch.sin(9 * xx + 3 * yy + k), torch.cos(7 * yy - 2 * xx + k), torch.sin(5 * (xx + yy) + k)], dim=-1)).contiguous() for k in range(len(cams))]
warm = MiniTrainer(g, cams, targets, cfg, lr=1e-3, densify_every=2, target_points=int(1.1 * n), prune_rate=0.025,
                   min_views=1, total_steps=100, seed=0, loss=loss_kind)
warm.train(3); torch.cuda.synchronize(); del warm          # first-call costs (library handles, allocator growth) stay outside
tr = MiniTrainer(g, cams, targets, cfg, lr=1e-3, densify_every=25, target_points=int(1.1 * n), prune_rate=0.025,
                 min_views=5, total_steps=100, seed=0, loss=loss_kind)
torch.cuda.synchronize()
t_start = time.perf_counter()
windows = []
for w0 in range(0, 100, 10):
  t0 = time.perf_counter()
  tr.train(10)
  torch.cuda.synchronize()
  dt = time.perf_counter() - t0
  with torch.no_grad():
    O = sta.render_gaussians(tr.scene(), cams[0], cfg, use_sh=True).num_overlaps
  windows.append(dict(iterations=[w0, w0 + 10], ms_per_iteration=round(1e3 * dt / 10, 2), pairs_camera0=O, points=tr.num_points))
  print(windows[-1], flush=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t_start
log = tr.log
renders = sum(log.num_points) * len(cams)
print(json.dumps(dict(workload=f"c4: {n} Gaussians, 1080p, SH3, 100 iterations x {ncam} cameras, densify/prune every 25, LaProp step, loss " + loss_kind,
                      iterations_per_s=100 / dt, ms_per_iteration=1e3 * dt / 100, gaussian_renders_per_s=renders / dt,
                      points=[log.num_points[0], log.num_points[-1]], loss=[log.losses[0], log.losses[-1]],
                      mask_digests=[d[:12] for d in log.mask_digests], peak_mem_GB=torch.cuda.max_memory_allocated() / 1e9,
                      windows=windows)))
