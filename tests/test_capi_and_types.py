"""The C-ABI library loads and exports every symbol include/gsplat_hip.h declares (no compute calls: there
is no GPU here), the ctypes binding covers exactly that set, the product path refuses CPU tensors and a
missing library (no fallback), and the boundary types behave as the reference's consumers expect."""
import ctypes as C
import os
import re

import pytest
import torch

import splat_trainer_amd as sta
from splat_trainer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
  text = open(os.path.join(ROOT, "include", "gsplat_hip.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(gsr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(built_libs):
  lib = C.CDLL(built_libs[0])
  syms = _header_symbols()
  assert len(syms) >= 20
  for s in syms:
    assert hasattr(lib, s), f"{s} declared in gsplat_hip.h but not exported"
  assert sorted(_lib.PROTOTYPES) == syms, "ctypes binding and header disagree"


def test_host_only_entry_points(built_libs):
  lib = _lib.load()
  assert lib.gsr_abi_version() == _lib.ABI_VERSION
  assert lib.gsr_error_string(-2).decode() == "workspace too small"
  assert lib.gsr_sort_workspace_bytes(0) > 0
  assert lib.gsr_sort_workspace_bytes(10_000_000) >= (10_000_000 // 4096) * 256 * 4
  assert lib.gsr_scan_workspace_bytes(5_000_000) >= 256
  assert lib.gsr_cull_workspace_bytes(3_000_000) >= (3_000_000 // 1024) * 4
  p = _lib.raster_params(sta.RasterConfig(antialias=True, blur_cov=0.0))
  assert abs(p.blur - 0.3) < 1e-7 and p.antialias == 1 and p.tile_size == 16 and abs(p.q_max - 9.0) < 1e-6
  assert C.sizeof(_lib.GsrRasterParamsC) == 32
  # the ctypes mirrors of the ABI's structs have the layout the library was compiled with
  for which, mirror in enumerate((_lib.GsrRasterParamsC, _lib.GsrSegmentsC, _lib.GsrFrameC, _lib.GsrFramePlanC,
                                  _lib.GsrFrameResultC, _lib.GsrFrameBackwardC)):
    assert lib.gsr_struct_bytes(which) == C.sizeof(mirror), mirror.__name__
  assert lib.gsr_struct_bytes(6) == -1


def test_frame_plan_lays_out_disjoint_aligned_buffers(built_libs):
  """gsr_frame_plan (host only): every buffer of a frame gets its own 256-byte aligned range inside the arena it belongs
  to, the zero-filled head of the output arena covers exactly the buffers that must start at zero, optional buffers
  are absent (-1) when not asked for, sizes follow N / the pair capacity / the image, in both modes of the driver."""
  lib = _lib.load()
  cfg = sta.RasterConfig(compute_visibility=True, compute_point_heuristic=True)

  def plan(N, W, H, cap, K=16, jac=1, median=0, vis=1, grad=1, projected=False, C_=3):
    f = _lib.GsrFrameC(None if projected else 1, 1, 1, 1, 1, N, K, W, H, 1, 1, 1, 0.1, 100.0, _lib.raster_params(cfg), jac,
                       median, vis, grad, -1, 0, cap, 1 if projected else None, 1 if projected else None,
                       1 if projected else None, C_, None, None, None, None)
    p = _lib.GsrFramePlanC()
    rc = lib.gsr_frame_plan(C.byref(f), C.byref(p))
    return rc, p

  out_fields = ("prune_cost", "split_score", "counts", "tile_range", "vis_partial", "indexes", "rows", "screen_scale",
                "jacobian", "visibility", "image", "final_T", "last", "median", "count", "offsets", "vals_a", "vals_b",
                "tvals_a", "tvals_b", "trank_a", "trank_b", "pair_vis", "seg_tables", "seg_pix", "seg_last")
  work_fields = ("cull_ws", "sort_ws", "scan_ws", "tsort_ws", "keys_a", "keys_b", "tile_hits", "tkeys_a", "tkeys_b")
  for kw in (dict(N=500_000, W=1920, H=1080, cap=2_000_000), dict(N=10_000, W=256, H=256, cap=65_536, K=1, median=1),
             dict(N=3_000_000, W=1920, H=1080, cap=8_000_000, grad=0), dict(N=777, W=33, H=17, cap=4096, projected=True, C_=1),
             dict(N=1, W=1, H=1, cap=0)):
    rc, p = plan(**kw)
    assert rc == 0, kw
    N, cap, P = kw["N"], kw["cap"], kw["W"] * kw["H"]
    for fields, total in ((out_fields, p.out_bytes), (work_fields, p.work_bytes)):
      offs = sorted(getattr(p, f) for f in fields if getattr(p, f) >= 0)
      assert all(o % 256 == 0 for o in offs) and len(set(offs)) == len(offs) and (not offs or offs[-1] <= total), kw
    assert p.zero_begin == 0 and p.prune_cost == 0 and p.zero_bytes % 256 == 0
    zeroed = {f for f in out_fields if 0 <= getattr(p, f) < p.zero_bytes}
    want_zero = {"prune_cost", "split_score", "counts", "tile_range"} | ({"vis_partial"} if (kw.get("vis", 1) or kw.get("grad", 1)) else set())
    assert zeroed == want_zero, (kw, zeroed)
    assert p.rows - p.indexes >= 8 * N if p.indexes >= 0 else kw.get("projected")
    assert (p.median >= 0) == bool(kw.get("median", 0))
    assert (p.jacobian >= 0) == (not kw.get("projected") and kw.get("K", 16) > 1)
    assert p.final_T - p.image >= 4 * kw.get("C_", 3) * P
    assert p.out_bytes >= 64 * N + 20 * P + 24 * cap and p.work_bytes >= 24 * N + 8 * cap
    assert (p.seg_capacity > 0) == (cap > 0) and p.seg_heavy_capacity <= p.seg_capacity
  assert plan(N=10, W=8, H=8, cap=16, K=5)[0] < 0                       # unsupported SH size
  assert plan(N=10, W=8, H=8, cap=16, projected=True, C_=4)[0] < 0      # more than 3 feature channels
  assert plan(N=-1, W=8, H=8, cap=16)[0] < 0


def test_segment_rule_and_capacity_bounds(built_libs):
  """Host side of the list segmentation (no GPU needed): the per-frame thresholds, and the buffer bounds -- the bound
  for "at most O pairs" (what the renderer sizes its tables with before the pair count is known) must cover the exact
  bound of every smaller frame, for the segment table and for the heavy-tile list alike."""
  lib = _lib.load()

  def thresholds(seg, heavy, O, tiles, grads):
    a, b = C.c_int32(0), C.c_int32(0)
    assert lib.gsr_segment_thresholds(seg, heavy, O, tiles, grads, C.byref(a), C.byref(b)) == 0
    return a.value, b.value

  assert thresholds(-1, 0, 1_465_883, 8160, 1) == (64, 625)           # c2: ~180 pairs per tile -> 64-pair segments
  assert thresholds(-1, 0, 6_593_876, 8160, 1) == (136, 2393)         # c3: mean list / 6, heavy = 120 + O / 2900
  assert thresholds(-1, 0, 6_593_876, 8160, 0) == (1196, 2393)        # evaluation: only heavy tiles are cut
  assert thresholds(-1, 0, 100, 12, 1) == (64, 512)
  assert thresholds(8, 20, 10 ** 6, 100, 1) == (8, 20)                # explicit values win
  assert thresholds(40, 0, 0, 100, 1) == (40, 512)
  import random
  rnd = random.Random(1)
  for _ in range(3000):
    tiles = rnd.choice([1, 12, 300, 8160, 32400])
    bound = rnd.randint(1, 40_000_000)
    grads = rnd.randint(0, 1)
    seg_cfg, heavy_cfg = rnd.choice([(-1, 0), (-1, 0), (0 + 16, 64), (4, 0), (-1, 900)])
    cap_bound = lib.gsr_segment_capacity(bound, 1, seg_cfg, heavy_cfg, tiles, grads)
    hcap_bound = lib.gsr_segment_heavy_capacity(bound, 1, seg_cfg, heavy_cfg, tiles, grads)
    for O in (bound, bound // 2, bound // 7 + 1, rnd.randint(1, bound)):
      seg, heavy = thresholds(seg_cfg, heavy_cfg, O, tiles, grads)
      # the most segments a frame of O pairs on `tiles` tiles can produce: every cut tile yields <= len / seg + 1
      worst = O // seg + min(tiles, O // (seg + 1))
      assert cap_bound >= min(worst, lib.gsr_segment_capacity(O, 0, seg_cfg, heavy_cfg, tiles, grads)), (tiles, bound, O)
      piece = max(seg, min(256, (heavy // 2) & ~3), 1)
      worst_heavy = O // piece + min(tiles, O // (heavy + 1))
      assert hcap_bound >= min(worst_heavy, lib.gsr_segment_heavy_capacity(O, 0, seg_cfg, heavy_cfg, tiles, grads)), (tiles, bound, O)


def test_no_cpu_fallback():
  g = sta.Gaussians3D(torch.randn(4, 3), torch.randn(4, 4), torch.randn(4, 3), torch.randn(4, 1), torch.randn(4, 3))
  cam = sta.CameraParams(torch.eye(4), torch.tensor([50., 50., 16., 16.]), (32, 32))
  with pytest.raises(sta.GsplatHipError):
    sta.project_to_image(g, cam, sta.RasterConfig())
  with pytest.raises(sta.GsplatHipError):
    sta.evaluate_sh_at(torch.randn(4, 3, 1), torch.randn(4, 3), torch.arange(4), torch.zeros(3))
  with pytest.raises(sta.GsplatHipError):
    sta.render_projected(torch.arange(4), torch.randn(4, 6), torch.randn(4, 3), torch.rand(4, 1), cam, sta.RasterConfig())


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
  monkeypatch.setattr(_lib, "_lib", None)
  monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
  with pytest.raises(sta.GsplatHipError, match="no CPU fallback"):
    _lib.load()


def test_product_never_imports_oracle():
  pkg = os.path.join(ROOT, "splat-trainer_amd")
  for dirpath, _, files in os.walk(pkg):
    for f in files:
      if f.endswith(".py"):
        src = open(os.path.join(dirpath, f)).read()
        assert not re.search(r"^\s*(import|from)\s+\.*oracle", src, flags=re.M), f
        assert "torch_oracle" not in src, f


def test_pop_raster_config_contract():
  opts = dict(antialias=True, blur_cov=0.0, compute_visibility=True, compute_point_heuristic=True,
              render_median_depth=True, specular_weight=0.5)
  cfg = sta.pop_raster_config(opts)
  assert cfg.antialias and cfg.compute_visibility and cfg.compute_point_heuristic and cfg.blur_cov == 0.0
  assert opts == dict(render_median_depth=True, specular_weight=0.5)


def test_camera_params_and_rendered_points():
  T = torch.eye(4); T[:3, 3] = torch.tensor([1., 2., 3.])
  cam = sta.CameraParams(T, torch.tensor([10., 11., 4., 5.]), (8, 6), 0.1, 50.0).to("cpu", torch.float32)
  assert torch.allclose(cam.camera_position, torch.tensor([-1., -2., -3.]))
  M = 5
  vis = torch.tensor([0., 1., 0., 2., 3.])
  pts = sta.RenderedPoints(idx=torch.arange(M) * 2, depths=torch.rand(M, 1), opacity=torch.rand(M),
                           screen_scale=torch.rand(M, 2), visibility=vis, prune_cost=torch.zeros(M),
                           split_score=torch.zeros(M))
  assert pts.num_visible == 3
  v = pts.visible
  assert v.idx.tolist() == [2, 6, 8] and v.depths.shape == (3, 1) and v.batch_size == (3,)
  pts2 = pts.replace(attributes=dict(a=1))
  assert pts2.attributes == dict(a=1) and pts.attributes is None
  r = sta.Rendering(image=torch.rand(6, 8, 3, requires_grad=True), camera=cam, points=pts,
                    median_depth_image=torch.rand(6, 8) + 0.2)
  d = r.detach()
  assert not d.image.requires_grad and d.image_size == (8, 6)
  assert d.median_ndc_image.shape == (6, 8)


def test_point_state_consumes_rendered_points():
  from splat_trainer_amd.controller_math import PointState, find_split_prune_indexes
  torch.manual_seed(0)
  N, M = 50, 20
  st = PointState.new_zeros(N, "cpu")
  for _ in range(6):
    idx = torch.randperm(N)[:M].sort().values
    pts = sta.RenderedPoints(idx=idx, depths=torch.rand(M, 1), opacity=torch.rand(M), screen_scale=torch.rand(M, 2) * 10,
                             visibility=torch.rand(M) * (torch.rand(M) > 0.3), prune_cost=torch.rand(M),
                             split_score=torch.rand(M))
    st.add_rendering(sta.Rendering(image=torch.zeros(2, 2, 3), camera=None, points=pts))
  split, prune = find_split_prune_indexes(st, t=0.2, target_points=60, min_views=1)
  assert split.dtype == torch.bool and not (split & prune).any() and split.sum() > 0


def test_compat_names_of_the_import_swap():
  """TaichiQueue / count_nonfinite / random test data exist with the call shapes the reference uses
  (mlp_scene.py:417, trainer.py:582, scripts/test_split.py:21-25)."""
  import math
  import torch
  import splat_trainer_amd as sta
  sta.TaichiQueue.init(arch=None, debug=True, threaded=True)
  assert sta.TaichiQueue.run_sync(lambda a, b: a + b, 2, b=3) == 5
  state = dict(points=dict(position=torch.tensor([[1.0, float("nan")]]), idx=torch.tensor([1, 2])),
               opt=[torch.tensor([float("inf"), 1.0, -float("inf")])], ok=torch.ones(3))
  assert sta.count_nonfinite(state, "scene") == {"scene.points.position": 1, "scene.opt[0]": 2}
  assert sta.count_nonfinite(dict(a=torch.ones(2)), "x") == {}
  try:
    sta.check_finite(state, "scene")
    raise AssertionError("check_finite did not raise")
  except ValueError:
    pass
  gen = torch.Generator().manual_seed(0)
  cam = sta.random_camera(image_size=(640, 480), generator=gen)
  g = sta.random_3d_gaussians(50, cam, alpha_range=(0.5, 1.0), scale_factor=0.2, generator=gen)
  assert cam.image_size == (640, 480) and g.position.shape == (50, 3) and g.feature.shape == (50, 3)
  op = torch.sigmoid(g.alpha_logit)
  assert float(op.min()) >= 0.5 - 1e-6 and float(op.max()) <= 1.0
  # every generated point projects inside the image, in front of the camera
  p = (cam.T_camera_world @ torch.cat([g.position, torch.ones(50, 1)], dim=1).T).T
  u = p[:, 0] / p[:, 2] * cam.projection[0] + cam.projection[2]
  v = p[:, 1] / p[:, 2] * cam.projection[1] + cam.projection[3]
  assert bool((p[:, 2] > 0).all()) and bool(((u >= -1e-3) & (u <= 640 + 1e-3) & (v >= -1e-3) & (v <= 480 + 1e-3)).all())
  R = cam.T_camera_world[:3, :3]
  assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5) and math.isclose(float(torch.linalg.det(R)), 1.0, abs_tol=1e-5)
  d = g.to_tensordict()
  assert set(d) == {"position", "rotation", "log_scaling", "alpha_logit", "feature"}
  assert sta.Gaussians3D.from_tensordict(d).position is g.position
