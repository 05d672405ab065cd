"""K6 / K7 timing of one build of the library on the c2 (or c3) workload, with a digest of the gradients so that variants
can be checked for bit-identity.  Used with GSPLAT_HIP_LIB=<variant .so> to compare kernel experiments:

    python splat-trainer_amd/build.py      # product build
    python -c "import importlib.util,sys; ..."   # or: tools/build_variant.py NAME -DFLAG
    GSPLAT_HIP_LIB=/path/libvariant.so python tools/k67_bench.py [c2|c3] [steps]
"""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import splat_trainer_amd as sta
from splat_trainer_amd import renderer, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
if which == "c2":
  g, cam = synthetic.scene_a(500_000, 1920, 1080, sh_degree=3, seed=0)
else:
  g, cams = synthetic.scene_b(3_000_000, 1920, 1080, sh_degree=3, seed=1, num_cameras=8)
  cam = cams[0]
if os.environ.get("MORTON") == "1":            # experiment: scene rows in Morton order of their positions (30-bit codes)
  def spread(v):
    v = v & 0x3FF
    v = (v | (v << 16)) & 0x030000FF
    v = (v | (v << 8)) & 0x0300F00F
    v = (v | (v << 4)) & 0x030C30C3
    v = (v | (v << 2)) & 0x09249249
    return v
  p = g.position
  q = ((p - p.min(0).values) / (p.max(0).values - p.min(0).values).clamp_min(1e-9) * 1023.0).long()
  code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
  perm = torch.argsort(code)
  g = sta.Gaussians3D(position=g.position[perm].contiguous(), log_scaling=g.log_scaling[perm].contiguous(),
                      rotation=g.rotation[perm].contiguous(), alpha_logit=g.alpha_logit[perm].contiguous(),
                      feature=g.feature[perm].contiguous())
g, cam = g.to("cuda"), cam.to("cuda")
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True,
                       segment_pairs=int(os.environ.get("SEG_PAIRS", "-1")), segment_min_pairs=int(os.environ.get("SEG_MIN", "0")))
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
for _ in range(steps):
  r = step()
e1.record()
torch.cuda.synchronize()
renderer.KERNEL_TIMER = None
ks = timer.summary()
h = hashlib.sha256()
for t in (r.image, r.points.prune_cost, r.points.split_score, r.points.visibility) + tuple(p.grad for p in params):
  h.update(t.detach().cpu().numpy().tobytes())
print(f"{os.environ.get('GSPLAT_HIP_LIB', 'product')} seg {cfg.segment_pairs}/{cfg.segment_min_pairs}: {which} O {r.num_overlaps}  K6 {ks['composite_forward'][1] * 1e3:.1f} us  "
      f"K7 {ks['composite_backward'][1] * 1e3:.1f} us  step {e0.elapsed_time(e1) / steps * 1e3:.0f} us  digest {h.hexdigest()[:16]}",
      flush=True)
