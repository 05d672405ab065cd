"""Where is a step host-bound?  Host enqueue time against GPU time of the three phases of one c2/c3 step
(project+SH | bin+sort+composite | loss+backward), unprofiled.   python tools/host_vs_gpu.py [c2|c3]"""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
import splat_trainer_amd as sta
from splat_trainer_amd import renderer

name = sys.argv[1] if len(sys.argv) > 1 else "c2"
g, cams, w = bench.make_workload(name, 1)
dev = torch.device("cuda:0")
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True, blur_cov=0.3, antialias=False)
params = [t.to(dev).requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
position, log_scaling, rotation, alpha_logit, feature = params
for p in params:
  p.grad = torch.zeros_like(p)
go = renderer.GradOut(position.grad, log_scaling.grad, rotation.grad, alpha_logit.grad, feature.grad)
scene = sta.Gaussians3D(position=position, rotation=rotation, log_scaling=log_scaling, alpha_logit=alpha_logit, feature=feature)
cam = cams[0].to(dev)
target = torch.full((w["h"], w["w"], 3), 0.5, device=dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
rows = []
for it in range(60):
  t = [time.perf_counter()]
  ev[0].record()
  with torch.enable_grad():
    prefetch = {}
    sh_out = (go._check("feature", feature), go._check("position", position), go)
    prefetch["sh"] = (feature, position, cam.camera_position, True)
    g2d, depth, idx = sta.project_to_image(scene, cam, cfg, grad_out=go, prefetch=prefetch)
    feats = sta.evaluate_sh_at(feature, position, idx, cam.camera_position, grad_out=sh_out, _precomputed=prefetch.pop("sh_out", None))
    t.append(time.perf_counter()); ev[1].record()
    r = sta.render_projected(idx, g2d, feats, depth, cam, cfg, _depth_order=prefetch.get("depth_order"))
    t.append(time.perf_counter()); ev[2].record()
    loss = sta.clamped_mse_loss(r.image, target)
    loss.backward()
  t.append(time.perf_counter()); ev[3].record()
  torch.cuda.synchronize()
  t.append(time.perf_counter())
  if it >= 30:
    rows.append([1e6 * (t[i + 1] - t[i]) for i in range(4)] + [1e3 * ev[i].elapsed_time(ev[i + 1]) for i in range(3)])
m = [sorted(c)[len(c) // 2] for c in zip(*rows)]
print(f"{name}: host us  project+sh {m[0]:.0f}  raster fwd {m[1]:.0f}  loss+bwd {m[2]:.0f}  final wait {m[3]:.0f}")
print(f"{name}: gpu  us  project+sh {m[4]:.0f}  raster fwd {m[5]:.0f}  loss+bwd {m[6]:.0f}   sum {m[4]+m[5]+m[6]:.0f}")
