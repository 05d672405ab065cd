"""Randomised parity sweep: many small scenes with extreme parameters (tiny / huge / very anisotropic splats,
opacities near 0 and 1, points behind the camera and on the frustum edge, antialias on/off, odd image sizes)
rendered by the HIP path and by the fp64 oracle.  Catches corner cases of the cull, the exact tile test, the
tile-half masks, the depth sort and the saturation logic that hand-picked scenes miss."""
import math

import pytest
import torch
import torch.nn.functional as F

import splat_trainer_amd as sta
from helpers import clean_or_isolated_flip, compare_explained, hip_render_and_grads, oracle_render_and_grads

pytestmark = pytest.mark.gpu


def random_case(seed, mid=False):
  """``mid``: thousands to tens of thousands of splats on images of a few hundred pixels (lists of hundreds of pairs per
  tile, several radix-sort blocks, segmented tiles under the product's own thresholds) instead of hundreds on a thumbnail."""
  gen = torch.Generator().manual_seed(seed)
  r = lambda *s: torch.rand(*s, generator=gen)
  if mid:
    n = int(3000 + r(1).item() ** 2 * 37000)
    w, h = int(90 + r(1).item() * 500), int(70 + r(1).item() * 400)
  else:
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


# Seeds that showed an isolated flip under one build or another (a one-off sweep over seeds 32..231 in round 2 found
# 46, 73, 88, 95, 96, 161, 174, 206, 222; pinning the forward projection's fma pattern in round 3 moved the flips to
# other seeds): kept in the sweep as the scenes known to sit close to a boundary.
SENSITIVE = (46, 73, 88, 95, 96, 161, 174, 206, 222)
SWEEP = list(range(48)) + list(SENSITIVE)
FLIPPED = {}
KEYS = ("image", "final_T", "visibility", "prune_cost", "split_score", "screen_scale", "depth", "d_position",
        "d_log_scaling", "d_rotation", "d_alpha_logit", "d_feature")


@pytest.mark.parametrize("seed", SWEEP)
def test_random_scene_matches_oracle_up_to_isolated_flips(seed):
  """Extreme anisotropy makes the conic ill-conditioned, so this sweep (a net for gross errors: missed tiles, wrong half
  masks, wrong order, wrong saturation) runs at 2e-4 instead of 1e-4.  Every tensor of every scene is either clean --
  no entry above 2e-4 of the tensor's largest magnitude -- or shows an isolated boundary flip (helpers.py:
  clean_or_isolated_flip); the visible set is always identical.  The share of scenes with a flip is bounded below."""
  g, cam, cfg = random_case(seed)
  hip = hip_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  orc = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  assert len(set(hip["idx"].tolist()) ^ set(orc["idx"].tolist())) == 0
  flipped = False
  live = []
  for k in KEYS:
    if orc[k].abs().max() == 0:
      assert hip[k].abs().max() == 0, (seed, k)
      continue
    live.append(k)
    flipped |= clean_or_isolated_flip(f"fuzz seed {seed}", k, hip[k], orc[k], 2e-4)
  # the share / size rule above is the net for gross errors; what passes it must also be EXPLAINED: every entry above the
  # tolerance on a pixel / a splat within FLIP_ULPS of a decision boundary of the oracle's own walk (median depth aside:
  # its decision, T crossing one half, is not one of the walk's reported boundaries)
  compare_explained(f"fuzz seed {seed} (explained)", hip, orc, 2e-4, keys=[k for k in live if k != "median"])
  FLIPPED[seed] = flipped


def test_flips_are_rare():
  """Round 2 observed 9 flipped scenes among 200 random ones (4.5 %).  Of the 48 unselected seeds of this sweep at most
  one in eight may show one (the SENSITIVE seeds are selected for it and do not count)."""
  seen = [s for s in range(48) if s in FLIPPED]
  if len(seen) < 40:
    pytest.skip("the sweep did not run in this session")
  flips = [s for s in seen if FLIPPED[s]]
  print("fuzz seeds with an isolated flip:", flips, "sensitive:", [s for s in SENSITIVE if FLIPPED.get(s)])
  assert len(flips) <= len(seen) // 8, flips


@pytest.mark.parametrize("seed", [300, 301, 302, 303, 304, 305])
def test_random_scene_with_many_large_splats(seed):
  """The sweep's scenes with a tenth of their splats blown up 8-40x (supports of tens to hundreds of tiles, hanging over
  every image border, on images whose sides are no multiple of 16): the wave-cooperative count / emit of large extents and
  the group-wise reduction of large slot ranges (csrc/binning.hip) against the oracle, under the same rules as above."""
  g, cam, cfg = random_case(seed)
  gen = torch.Generator().manual_seed(seed)
  n = g.position.shape[0]
  pick = torch.randperm(n, generator=gen)[: max(2, n // 10)]
  ls, al = g.log_scaling.clone(), g.alpha_logit.clone()
  ls[pick] += 2.1 + 1.6 * torch.rand(pick.numel(), 1, generator=gen)
  al[pick] = -2.5 + 1.5 * torch.rand(pick.numel(), 1, generator=gen)      # faint: what lies behind them still counts
  g = sta.Gaussians3D(g.position, g.rotation, ls, al, g.feature)
  hip = hip_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  orc = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
  assert len(set(hip["idx"].tolist()) ^ set(orc["idx"].tolist())) == 0
  live = []
  for k in KEYS:
    if orc[k].abs().max() == 0:
      assert hip[k].abs().max() == 0, (seed, k)
      continue
    live.append(k)
    clean_or_isolated_flip(f"fuzz large {seed}", k, hip[k], orc[k], 2e-4)
  compare_explained(f"fuzz large {seed} (explained)", hip, orc, 2e-4, keys=live)


def blown_up(g, seed):
  """A tenth of the scene's splats scaled 8-40x and made faint (the large-splat variant above)."""
  gen = torch.Generator().manual_seed(seed)
  n = g.position.shape[0]
  pick = torch.randperm(n, generator=gen)[: max(2, n // 10)]
  ls, al = g.log_scaling.clone(), g.alpha_logit.clone()
  ls[pick] += 2.1 + 1.6 * torch.rand(pick.numel(), 1, generator=gen)
  al[pick] = -2.5 + 1.5 * torch.rand(pick.numel(), 1, generator=gen)
  return sta.Gaussians3D(g.position, g.rotation, ls, al, g.feature)


def test_extended_sweep_on_request():
  """One-off wide net, off by default: ``GSPLAT_FUZZ_EXTRA=n`` runs seeds 1000 .. 1000+n-1 (every fourth one with the
  large-splat variant; the frame's path -- default, forced segments, checkpointed backward, three-call form -- varies with
  the seed) under the same rules as the sweep above and reports every seed that breaks one instead of stopping
  at the first.  The result of the round's run is kept in profiles/ (r04_fuzz_extended.txt)."""
  import os
  extra = int(os.environ.get("GSPLAT_FUZZ_EXTRA", "0"))
  mid = os.environ.get("GSPLAT_FUZZ_MID", "0") == "1"       # the same sweep over mid-size scenes (seconds of oracle per scene)
  if extra <= 0:
    pytest.skip("set GSPLAT_FUZZ_EXTRA=n to run n more random scenes")
  first = int(os.environ.get("GSPLAT_FUZZ_START", "1000"))   # (a range the margin model was not developed on: START=3000)
  broken, outside_share_size, flips, above = [], [], 0, 0
  for seed in range(first, first + extra):
    g, cam, cfg = random_case(seed, mid=mid)
    if seed % 4 == 3:
      g = blown_up(g, seed)
    # the paths a frame can take, by seed: 0 / 4 the product's own choices; 1 lists cut into tiny segments forward AND backward
    # (the heavy-tile passes); 2 the checkpointed backward walk only; 3 the reference's three-call sequence
    form = (seed // 4) % 5
    import dataclasses
    if form == 1:
      n = 1 + seed % 13
      cfg = dataclasses.replace(cfg, segment_pairs=n, segment_min_pairs=n)
    elif form == 2:
      cfg = dataclasses.replace(cfg, segment_pairs=4 * (1 + seed % 7), segment_min_pairs=10 ** 9)
    hip = hip_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0, three_call=form == 3)
    orc = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
    try:
      assert len(set(hip["idx"].tolist()) ^ set(orc["idx"].tolist())) == 0, "visible sets differ"
      live = []
      flipped = False
      for k in KEYS:
        if orc[k].abs().max() == 0:
          assert hip[k].abs().max() == 0, (seed, k)
          continue
        live.append(k)
        try:                                # the share / size rule of the committed sweep: counted here, not fatal --
          flipped |= clean_or_isolated_flip(f"fuzz x{seed}", k, hip[k], orc[k], 2e-4)
        except AssertionError as e:         # an explained flip has no principled size bound (a splat seen through a dozen
          flipped = True                    # rim pixels loses a tenth of its gradient row with one of them)
          outside_share_size.append((seed, k, str(e)[-60:]))
      above += compare_explained(f"fuzz x{seed} (explained)", hip, orc, 2e-4, keys=live, size=1.0)
      flips += int(flipped)
    except AssertionError as e:
      broken.append((seed, str(e)[:300]))
    if (seed - first + 1) % (2 if mid else 20) == 0:
      print(f"[extended sweep] {seed - first + 1} scenes, {flips} with a flip, {above} entries above tolerance, "
            f"{len(broken)} broken", flush=True)
  print(f"[extended sweep] seeds {first}..{first + extra - 1}: {flips} scenes with an isolated flip, {above} entries above "
        f"2e-4, unexplained / broken: {broken}; explained but outside the committed sweep's share / size allowance: "
        f"{outside_share_size}", flush=True)
  assert not broken, broken


def test_diagnose_seeds_on_request():
  """``GSPLAT_FUZZ_DIAG=1240,1789`` prints, for every gradient / per-point row above 2e-4 in those scenes of the extended
  sweep, what the explained-flip rule looks at: the splat's margin, its own margin, its conic's condition, the row's error."""
  import os
  from helpers import conic_condition
  seeds = [int(s) for s in os.environ.get("GSPLAT_FUZZ_DIAG", "").split(",") if s.strip()]
  if not seeds:
    pytest.skip("set GSPLAT_FUZZ_DIAG=seed,seed,...")
  mid = os.environ.get("GSPLAT_FUZZ_MID", "0") == "1"
  for seed in seeds:
    g, cam, cfg = random_case(seed, mid=mid)
    if seed >= 1000 and seed % 4 == 3:
      g = blown_up(g, seed)
    form = (seed // 4) % 5 if seed >= 1000 else 0              # (the extended sweep's choice of path)
    import dataclasses
    if form == 1:
      cfg = dataclasses.replace(cfg, segment_pairs=1 + seed % 13, segment_min_pairs=1 + seed % 13)
    elif form == 2:
      cfg = dataclasses.replace(cfg, segment_pairs=4 * (1 + seed % 7), segment_min_pairs=10 ** 9)
    print(f"--- seed {seed} form {form}")
    hip = hip_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0, three_call=form == 3)
    orc = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0)
    o32 = oracle_render_and_grads(g, cam, cfg, use_sh=True, want_median=True, loss_scale=100.0, dtype=torch.float32)
    idx = orc["idx"].cpu()
    sm, om = orc["splat_margin"].double().cpu(), orc["splat_own_margin"].double().cpu()
    cond = conic_condition(orc["g2d"].detach().cpu())
    g2d = orc["g2d"].detach().cpu()
    print(f"--- seed {seed}: {idx.numel()} visible, image {tuple(orc['image'].shape)}, aa={cfg.antialias}")
    for k in KEYS:
      a, b = hip[k].detach().double().cpu(), orc[k].detach().double().cpu()
      scale = max(b.abs().max().item(), 1e-30)
      if scale == 1e-30:
        continue
      bad = ((a - b).abs() / scale) > 2e-4
      if not bad.any():
        continue
      if k in ("image", "final_T"):
        pm = orc["pixel_margin"].double().cpu()
        ys, xs = torch.nonzero(bad.reshape(pm.shape[0], pm.shape[1], -1).any(dim=2), as_tuple=True)
        for y, x in zip(ys.tolist(), xs.tolist()):
          print(f"  {k:14s} pixel ({x},{y}) margin {pm[y, x]:.3g} err {((a - b).abs().reshape(pm.shape[0], pm.shape[1], -1)[y, x].max() / scale):.3g}"
                f"  hip {a.reshape(pm.shape[0], pm.shape[1], -1)[y, x].tolist()} oracle {b.reshape(pm.shape[0], pm.shape[1], -1)[y, x].tolist()} tensor max {scale:.3g}")
        continue
      rows = bad.reshape(bad.shape[0], -1).any(dim=1)
      per_point = rows.shape[0] == sm.shape[0] and not k.startswith("d_")
      vis_rows = torch.nonzero(rows if per_point else rows[idx]).flatten().tolist()
      for r in vis_rows:
        full = r if per_point else int(idx[r])
        ar, br = a.reshape(a.shape[0], -1)[full], b.reshape(b.shape[0], -1)[full]
        cr = o32[k].detach().double().cpu().reshape(a.shape[0], -1)[full]
        print(f"  {k:14s} oracle-fp32 row err/rowmax {((cr - br).abs().max() / br.abs().max().clamp_min(1e-30)):.3g}")
        print(f"  {k:14s} visible row {r:5d} margin {sm[r]:9.3g} own {om[r]:9.3g} cond {cond[r]:9.3g} "
              f"row_err/rowmax {((ar - br).abs().max() / br.abs().max().clamp_min(1e-30)):.3g} err/tensormax {((ar - br).abs().max() / scale):.3g} "
              f"opacity {g2d[r, 5]:.3g} uv ({g2d[r, 0]:.1f},{g2d[r, 1]:.1f}) conic ({g2d[r, 2]:.3g},{g2d[r, 3]:.3g},{g2d[r, 4]:.3g})")
