"""Per-wave timeline of the forward composite (K6): needs the diagnostic build
    python tools/build_variant.py trace GSR_K6_TRACE=1
    GSPLAT_HIP_LIB=$PWD/variants/libgsplat_hip_trace.so python tools/k6_trace.py [c2|c3]
Every forward wave records start / end (s_memrealtime, 10 ns ticks), its hardware id and its tile's list length; the
script prints how the launch's duration splits into per-SIMD load, occupancy over time and pace per pair."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import splat_trainer_amd as sta
from splat_trainer_amd import _lib, synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
if which == "c2":
  g, cam = synthetic.scene_a(500_000, 1920, 1080, sh_degree=3, seed=0)
else:
  g, cams = synthetic.scene_b(3_000_000, 1920, 1080, sh_degree=3, seed=1, num_cameras=8)
  cam = cams[0]
g, cam = g.to("cuda"), cam.to("cuda")
cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)
params = [t.requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit, g.feature)]
scene = sta.Gaussians3D(position=params[0], log_scaling=params[1], rotation=params[2], alpha_logit=params[3], feature=params[4])
lib = _lib.load()
num_tiles = ((1920 + 15) // 16) * ((1080 + 15) // 16)
trace = torch.zeros(num_tiles, 4, dtype=torch.int64, device="cuda")
lib.gsr_debug_set_k6_trace.argtypes = [C.c_void_p]
assert lib.gsr_debug_set_k6_trace(trace.data_ptr()) == 0


def step():
  with torch.enable_grad():
    r = sta.render_gaussians(scene, cam, cfg, use_sh=True)
    ((r.image.clamp(0, 1) - 0.5) ** 2).mean().backward()


for _ in range(8):
  step()
torch.cuda.synchronize()
t = trace.cpu().numpy().astype(np.uint64)
ran = t[:, 1] > 0
t0, t1 = t[ran, 0].astype(np.int64), t[ran, 1].astype(np.int64)
hw, xcc = (t[ran, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64), (t[ran, 2] >> np.uint64(32)).astype(np.int64) & 0xF
length = (t[ran, 3] & np.uint64(0xFFFFFFFF)).astype(np.int64)
block = (t[ran, 3] >> np.uint64(32)).astype(np.int64)
base = t0.min()
t0, t1 = (t0 - base) * 0.01, (t1 - base) * 0.01               # microseconds
dur = t1 - t0
simd = (hw >> 4) & 3
cu = (hw >> 8) & 0xF
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
unit = ((xcc * 8 + se) * 2 + sh) * 64 + cu * 4 + simd
print(f"{which}: {ran.sum()} waves, launch span {t1.max():.0f} us, starts within {t0.max():.0f} us; list length mean {length.mean():.0f} "
      f"max {length.max()}; wave duration mean {dur.mean():.0f} max {dur.max():.0f} us")
units = np.unique(unit)
print(f"{len(units)} distinct SIMDs seen; waves per SIMD: min {min((unit == u).sum() for u in units)} max {max((unit == u).sum() for u in units)}")
# per-SIMD: sum of pairs, last end
pairs = np.array([length[unit == u].sum() for u in units])
ends = np.array([t1[unit == u].max() for u in units])
print(f"pairs per SIMD: mean {pairs.mean():.0f} min {pairs.min()} max {pairs.max()}  (max/mean {pairs.max() / pairs.mean():.2f});  "
      f"last wave of a SIMD ends at: mean {ends.mean():.0f} min {ends.min():.0f} max {ends.max():.0f} us")
print(f"correlation(pairs on the SIMD, its end time) = {np.corrcoef(pairs, ends)[0, 1]:.2f}")
# occupancy over time
grid = np.linspace(0, t1.max(), 21)
occ = [(np.logical_and(t0 <= x, t1 > x)).sum() / len(units) for x in grid]
print("waves per SIMD over time:", " ".join(f"{o:.1f}" for o in occ))
# pace: ns per pair of the longest waves vs concurrency at their end
order = np.argsort(-length)[:10]
for i in order:
  alone_from = np.sort(t1[unit == unit[i]])[-2] if (unit == unit[i]).sum() > 1 else 0.0
  print(f"  tile list {length[i]:5d} pairs: {dur[i]:6.0f} us = {dur[i] * 1e3 / max(length[i], 1):5.0f} ns/pair; ran {t0[i]:.0f}..{t1[i]:.0f}; "
        f"alone on its SIMD from {alone_from:.0f} us; block {block[i]}")
q = np.argsort(length)
for lo, hi in ((0.45, 0.55), (0.9, 0.95)):
  sel = q[int(lo * len(q)):int(hi * len(q))]
  print(f"  lists at the {int(lo * 100)}-{int(hi * 100)} % quantile ({length[sel].mean():.0f} pairs): {dur[sel].mean():.0f} us = "
        f"{(dur[sel] * 1e3 / np.maximum(length[sel], 1)).mean():.0f} ns/pair, end at {t1[sel].mean():.0f} us")
