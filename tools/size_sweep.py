"""Whole-step timing over splat counts, against tuning cliffs: every switch of the path that was set on two scenes (c2 at
500 k, c3 at 3 M) -- the K6 instantiation with row prefetch (GSR_PREFETCH_MIN_ROWS), the radix sort's block size
(GSR_RS_BIG_MIN), the automatic segment lengths -- is crossed somewhere between 0.25 M and 3 M splats.
    python tools/size_sweep.py [A|B|AB] [n_millions ...]
One line per (scene, N): median step time (HIP events per step), K6 / K7 (events around the launches), pairs.
GSPLAT_HIP_LIB=<variant .so> selects a build with a switch forced one way (tools/build_variant.py)."""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import splat_trainer_amd as sta
from splat_trainer_amd import renderer, synthetic

scenes = sys.argv[1] if len(sys.argv) > 1 else "AB"
sizes = [float(a) for a in sys.argv[2:]] or [0.25, 0.5, 0.8, 1.0, 1.2, 1.5, 2.0, 3.0]
tag = os.path.basename(os.environ.get("GSPLAT_HIP_LIB", "product")).replace("libgsplat_hip_", "").replace(".so", "")
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
for sc in scenes:
  for nm in sizes:
    n = int(nm * 1e6)
    if sc == "A":
      g, cam = synthetic.scene_a(n, 1920, 1080, sh_degree=3, seed=0)
    else:
      g, cams = synthetic.scene_b(n, 1920, 1080, sh_degree=3, seed=1, num_cameras=8)
      cam = cams[0]
    g, cam = g.to("cuda"), cam.to("cuda")
    params = [t.requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
    scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])

    def step():
      for p in params:
        p.grad = None
      with torch.enable_grad():
        r = sta.render_gaussians(scene, cam, cfg, use_sh=True)
        ((r.image.clamp(0, 1) - 0.5) ** 2).mean().backward()
      return r

    for _ in range(8):
      r = step()
    torch.cuda.synchronize()
    timer = renderer.KernelTimer()
    renderer.KERNEL_TIMER = timer
    K = 16
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev[0].record()
    for i in range(K):
      r = step()
      ev[i + 1].record()
    torch.cuda.synchronize()
    renderer.KERNEL_TIMER = None
    ks = timer.summary()
    per = [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(K)]
    print(f"{tag:10s} scene {sc} N {nm:4.2f} M  O {r.num_overlaps:9d}  step median {statistics.median(per):7.0f} us  "
          f"K6 {ks['composite_forward'][1] * 1e3:6.1f}  K7 {ks['composite_backward'][1] * 1e3:6.1f}", flush=True)
    del g, params, scene, r
    torch.cuda.empty_cache()
