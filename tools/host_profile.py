"""cProfile of the host side of a step (which Python calls the enqueue time goes to).  python tools/host_profile.py [c2]"""
import cProfile
import pstats
import sys

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


def step():
  with torch.enable_grad():
    r = sta.render_gaussians(scene, cam, cfg, use_sh=True, grad_out=go)
    loss = sta.clamped_mse_loss(r.image, target)
    loss.backward()


for _ in range(30):
  step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
  step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(45)
