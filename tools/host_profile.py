"""Host-side (Python) cost of one step: cProfile over N steps of the bench's own step function on a small workload, where
the GPU finishes each stage faster than the host enqueues the next (c1: 10 k splats).  Run on a GPU box:
    python tools/host_profile.py [c1] [steps]
"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench
import splat_trainer_amd as sta
from splat_trainer_amd.controller_math import PointState
from splat_trainer_amd.distributed import CameraShardedStep

which = sys.argv[1] if len(sys.argv) > 1 else "c1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dev = torch.device("cuda", 0)
g, cams, w = bench.make_workload(which, 1)
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True, blur_cov=0.3, antialias=False)
params = [t.to(dev).requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
scene = sta.Gaussians3D(position=params[0], rotation=params[2], log_scaling=params[1], alpha_logit=params[3], feature=params[4])
dp = CameraShardedStep(params, 1, 0)
batch = [cams[0].to(dev)]
target = torch.full((w["h"], w["w"], 3), 0.5, device=dev)
state = PointState.new_zeros(params[0].shape[0], dev)


def render_backward(j, cam, grad_out, collector):
  with torch.enable_grad():
    r = sta.render_gaussians(scene, cam, cfg, use_sh=True, grad_out=grad_out, sh_collector=collector)
    sta.clamped_mse_loss(r.image, target).backward()
  return r


def step():
  dp.run(batch, render_backward, point_state=state)


for _ in range(50):
  step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
  step()
torch.cuda.synchronize()
print(f"{which}: {(time.perf_counter() - t0) / steps * 1e6:.0f} us/step unprofiled")
# the backward functions run on autograd's device thread, which cProfile does not see: time them by hand, then profile
# one of them in isolation on that thread
from splat_trainer_amd import loss as _loss, renderer as _renderer
acc = {}
profs = {}


def timed(cls, name):
  inner = cls.backward

  def wrapper(ctx, *grads):
    pr_ = profs.get(name)
    if pr_ is not None:
      pr_.enable()
    t = time.perf_counter()
    out = inner(ctx, *grads)
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    if pr_ is not None:
      pr_.disable()
    return out
  cls.backward = staticmethod(wrapper)


timed(_renderer._FrameFn, "frame_backward")
timed(_loss._PixelLossFn, "loss_backward")
for _ in range(steps):
  step()
torch.cuda.synchronize()
print({k: f"{v / steps * 1e6:.1f} us/step" for k, v in acc.items()})
profs["frame_backward"] = cProfile.Profile()
for _ in range(steps):
  step()
torch.cuda.synchronize()
pstats.Stats(profs["frame_backward"]).sort_stats("tottime").print_stats(25)
profs.clear()
pr = cProfile.Profile()
pr.enable()
for _ in range(steps):
  step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumulative").print_stats(45)
