"""Randomised parity sweep: many small scenes with extreme parameters (tiny / huge / very anisotropic splats,
opacities near 0 and 1, points behind the camera and on the frustum edge, antialias on/off, odd image sizes)
rendered by the HIP path and by the fp64 oracle.  Catches corner cases of the cull, the exact tile test, the
tile-half masks, the depth sort and the saturation logic that hand-picked scenes miss."""
import math

import pytest
import torch
import torch.nn.functional as F

import splat_trainer_amd as sta
from helpers import hip_render_and_grads, observe, oracle_render_and_grads

pytestmark = pytest.mark.gpu


def random_case(seed):
  gen = torch.Generator().manual_seed(seed)
  r = lambda *s: torch.rand(*s, generator=gen)
  n = int(20 + r(1).item() * 600)
  w, h = int(8 + r(1).item() * 150), int(8 + r(1).item() * 110)
  fov = math.radians(40 + 50 * r(1).item())
  fx = fy = w / (2 * math.tan(fov / 2))
  z = 0.05 + 12 * r(n) ** 2                                   # some in front of the near plane
  x = (r(n) * 1.6 - 0.3) * w                                   # beyond the image on both sides
  y = (r(n) * 1.6 - 0.3) * h
  pos = torch.stack([(x - w / 2) * z / fx, (y - h / 2) * z / fy, z], 1)
  if seed % 3 == 0:
    pos[: n // 10, 2] *= -1                                    # behind the camera
  base = torch.log(z * (0.3 + 8 * r(n) ** 3) / fx)             # sigma from 0.3 px to ~8 px, a few much larger
  ls = base[:, None] + 0.7 * torch.randn(n, 3, generator=gen)  # anisotropic (axis ratios up to ~10)
  rot = F.normalize(torch.randn(n, 4, generator=gen), dim=1)
  al = 4.0 * torch.randn(n, 1, generator=gen)                  # opacities from ~0 to ~1
  deg = seed % 4
  feat = 0.5 * torch.randn(n, 3, (deg + 1) ** 2, generator=gen)
  # random rigid pose
  ang = 0.3 * torch.randn(3, generator=gen)
  Rx = torch.tensor([[1, 0, 0], [0, math.cos(ang[0]), -math.sin(ang[0])], [0, math.sin(ang[0]), math.cos(ang[0])]])
  Ry = torch.tensor([[math.cos(ang[1]), 0, math.sin(ang[1])], [0, 1, 0], [-math.sin(ang[1]), 0, math.cos(ang[1])]])
  R = (Rx @ Ry).float()
  t = 0.2 * torch.randn(3, generator=gen)
  T = torch.eye(4); T[:3, :3] = R; T[:3, 3] = t
  pos_w = (pos - t) @ R                                        # world point such that R p + t = pos
  cam = sta.CameraParams(T, torch.tensor([fx, fy, w / 2 + 3 * (r(1).item() - .5), h / 2]), (w, h), 0.1, 50.0)
  aa = seed % 2 == 1
  cfg = sta.RasterConfig(antialias=aa, blur_cov=0.0 if aa else 0.3, compute_visibility=True, compute_point_heuristic=True)
  g = sta.Gaussians3D(pos_w.float(), rot.float(), ls.float(), al.float(), feat.float())
  return g, cam, cfg


@pytest.mark.parametrize("seed", list(range(32)))
def test_random_scene_matches_oracle(seed):
  g, cam, cfg = random_case(seed)
  hip = hip_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  orc = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  assert len(set(hip["idx"].tolist()) ^ set(orc["idx"].tolist())) == 0
  # Extreme anisotropy makes the conic ill-conditioned, so this sweep (a net for gross errors: missed tiles, wrong half
  # masks, wrong order) runs at 2e-4 instead of 1e-4 -- with NO outliers allowed.  Observed over the 32 seeds
  # (profiles/r02_parity_observed.txt): largest error 7.7e-5, nothing above 1e-4.
  tol = 2e-4
  for k in ("image", "final_T"):
    worst, frac = observe(f"fuzz seed {seed}", k, hip[k], orc[k], tol)
    assert frac == 0 and worst < tol, (seed, k, frac, worst)
  for k in ("visibility", "prune_cost", "split_score", "screen_scale", "depth", "d_position", "d_log_scaling",
            "d_rotation", "d_alpha_logit", "d_feature"):
    if orc[k].abs().max() == 0:
      assert hip[k].abs().max() == 0, (seed, k)
      continue
    worst, frac = observe(f"fuzz seed {seed}", k, hip[k], orc[k], tol)
    assert frac == 0 and worst < tol, (seed, k, frac, worst)


FLIP_SEEDS = (46, 73, 88, 95, 96, 161, 174, 206, 222)


@pytest.mark.parametrize("seed", FLIP_SEEDS)
def test_random_scenes_with_a_boundary_flip_stay_isolated(seed):
  """A one-off sweep over 200 further seeds (32..231) found these nine (4.5 %) with entries above 2e-4: a pixel within
  fp32 rounding of a discrete contribute / skip boundary (q = 9, alpha = 1/255, T = 1e-4) takes the other branch than the
  fp64 oracle and moves that pixel -- and the sums of the one or two splats involved -- by one minimal contribution.
  The round-1 build shows the same nine seeds with the same figures, so this is arithmetic, not a defect of a later
  change.  Kept as a regression net with the observed sizes x 2: at most 0.4 % of a tensor's entries (or 3 of them, for
  the small per-point tensors) may exceed 2e-4, by at most 1e-2 of the tensor's largest magnitude, and the visible set
  must be identical."""
  g, cam, cfg = random_case(seed)
  hip = hip_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  orc = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  assert len(set(hip["idx"].tolist()) ^ set(orc["idx"].tolist())) == 0
  flipped = False
  for k in ("image", "final_T", "visibility", "prune_cost", "split_score", "screen_scale", "depth", "d_position",
            "d_log_scaling", "d_rotation", "d_alpha_logit", "d_feature"):
    if orc[k].abs().max() == 0:
      assert hip[k].abs().max() == 0, (seed, k)
      continue
    worst, frac = observe(f"fuzz flip seed {seed}", k, hip[k], orc[k], 2e-4)
    assert frac * hip[k].numel() <= max(3.0, 4e-3 * hip[k].numel()) + 0.5 and worst < 1e-2, (seed, k, frac, worst)
    flipped |= frac > 0
  assert flipped, "this seed no longer shows a flip: move it to the clean sweep"
