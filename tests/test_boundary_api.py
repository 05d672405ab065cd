"""The boundary types answer the calls the reference makes on them (CPU only; no kernel is launched).

Each test replays the call SHAPE of one reference site with this build's types:
  scene/io.py:106-123 ............ Gaussians3D(..., batch_size=(n,)), gaussians.apply(torch.detach)
  scene/mlp_scene.py:296 ......... points.tensors.to_dict()
  scene/mlp_scene.py:301-310 ..... split_gaussians_uniform(points[split_idx].detach(), k=2, random_axis=True);
                                   points[keep_mask].append_tensors(splits)
  scene/mlp_scene.py:394-398 ..... points.tensors.select(...).replace(feature=...); Gaussians3D.from_dict(td, batch_dims=1)
  gaussians/split.py:44-52,87-113  points['log_scaling'], points.update(dict(...)) in place, points.apply(fn, batch_size=[n])
"""
import math
from functools import partial

import pytest
import torch
import torch.nn.functional as F

import splat_trainer_amd as sta
from splat_trainer_amd.harness import point_basis, split_gaussians_uniform
from splat_trainer_amd.optim import ParameterClass, SparseAdam
from splat_trainer_amd.tensor_rows import TensorRows


def _points(n=12, k=4):
  torch.manual_seed(0)
  tensors = dict(position=torch.randn(n, 3), log_scaling=0.3 * torch.randn(n, 3), rotation=torch.randn(n, 4),
                 alpha_logit=torch.randn(n, 1), feature=torch.randn(n, 3, k), visible=torch.zeros(n))
  groups = {name: dict(lr=1e-3) for name in ("position", "log_scaling", "rotation", "alpha_logit", "feature")}
  return ParameterClass(tensors, groups, optimizer=SparseAdam)


def test_gaussians3d_constructor_and_apply_as_in_scene_io():
  n = 7
  g = sta.Gaussians3D(position=torch.randn(n, 3), rotation=torch.randn(n, 4), alpha_logit=torch.randn(n, 1),
                      log_scaling=torch.randn(n, 3), feature=torch.randn(n, 3, 4), batch_size=(n,))   # io.py:106-115
  assert g.batch_size == (n,)
  with pytest.raises(ValueError):
    sta.Gaussians3D(position=torch.randn(n, 3), rotation=torch.randn(n, 4), alpha_logit=torch.randn(n, 1),
                    log_scaling=torch.randn(n, 3), feature=torch.randn(n, 3), batch_size=(n + 1,))
  g.position.requires_grad_(True)
  d = g.apply(torch.detach)                                                                           # io.py:122
  assert isinstance(d, sta.Gaussians3D) and not d.position.requires_grad and d.position.data_ptr() == g.position.data_ptr()
  assert torch.equal(g.to(dtype=torch.float64).feature, g.feature.double())


def test_tensors_select_replace_from_dict_as_in_to_sh_gaussians():
  pts = _points()
  assert isinstance(pts.tensors.to_dict(), dict) and set(pts.tensors.to_dict()) == set(pts.keys())    # mlp_scene.py:296
  sel = pts.tensors.select("position", "rotation", "log_scaling", "alpha_logit")                      # mlp_scene.py:395
  assert set(sel) == {"position", "rotation", "log_scaling", "alpha_logit"} and sel.batch_size == (12,)
  sh = torch.randn(12, 3, 9)
  td = sel.replace(feature=sh)                                                                        # mlp_scene.py:396
  assert "feature" not in sel and td["feature"] is sh
  g = sta.Gaussians3D.from_dict(td, batch_dims=1)                                                     # mlp_scene.py:398
  assert g.feature is sh and g.position is pts.position
  with pytest.raises(ValueError):
    sta.Gaussians3D.from_dict(td, batch_dims=2)


def test_tensor_rows_update_is_in_place_and_apply_rebatches():
  td = TensorRows(a=torch.arange(6.).view(3, 2), b=torch.arange(3.))
  same = td.update(dict(b=torch.ones(3)))                                       # split.py:103-104: returns self
  assert same is td and torch.equal(td["b"], torch.ones(3))
  rep = td.apply(partial(torch.repeat_interleave, repeats=2, dim=0), batch_size=[6])    # split.py:47-49
  assert rep.batch_size == (6,) and rep["a"].shape == (6, 2)
  with pytest.raises(ValueError):
    td.apply(partial(torch.repeat_interleave, repeats=2, dim=0), batch_size=[5])
  with pytest.raises(ValueError):
    td["c"] = torch.zeros(4)
  rows = td[torch.tensor([True, False, True])]
  assert isinstance(rows, TensorRows) and rows.batch_size == (2,) and torch.equal(rows["a"], td["a"][[0, 2]])


def test_split_and_prune_call_shapes_as_in_mlp_scene():
  pts = _points()
  split_idx = torch.tensor([1, 4, 5])
  keep_mask = torch.ones(12, dtype=torch.bool)
  keep_mask[[1, 4, 5, 9]] = False
  rows = pts[split_idx].detach()                                                # mlp_scene.py:303
  assert isinstance(rows, TensorRows) and rows.batch_size == (3,) and not rows["position"].requires_grad
  gen = torch.Generator().manual_seed(3)
  splits = split_gaussians_uniform(rows, k=2, random_axis=True, generator=gen)
  assert splits.batch_size == (6,) and set(splits) == set(pts.keys())           # extras ('visible') ride along
  out = pts[keep_mask].append_tensors(splits)                                   # mlp_scene.py:306-310
  assert out.num_points == 8 + 6 and out.tensor_state["position"]["exp_avg"].shape == (14, 3)
  with pytest.raises(sta.GsplatHipError):           # the fused device form has no CPU fallback (tests/test_gpu_densify.py)
    pts.keep_and_append(keep_mask, splits)


def test_split_offsets_follow_the_in_place_update_of_the_reference():
  """gaussians/split.py:103-108: ``points.update(...)`` shrinks log_scaling IN PLACE before ``multi_sample_gaussians``
  builds the basis, so the two children of a parent sit at -/+ sep * sigma / sqrt(2) along the chosen axis
  (hand-computed below for an axis-aligned splat), and the chosen axis shrinks by 1/sqrt(2)."""
  rows = TensorRows(position=torch.tensor([[1.0, 2.0, 3.0]]), log_scaling=torch.log(torch.tensor([[0.5, 2.0, 1.0]])),
                    rotation=torch.tensor([[0.0, 0.0, 0.0, 1.0]]), alpha_logit=torch.zeros(1, 1), feature=torch.zeros(1, 3))
  out = split_gaussians_uniform(rows, k=2, random_axis=False)                  # largest axis: y, sigma = 2
  shift = 0.7 * 2.0 / math.sqrt(2.0)
  want = torch.tensor([[1.0, 2.0 - shift, 3.0], [1.0, 2.0 + shift, 3.0]])
  assert torch.allclose(out["position"], want, atol=1e-6)
  assert torch.allclose(out["log_scaling"].exp(), torch.tensor([[0.5, 2.0 / math.sqrt(2.0), 1.0]]).expand(2, 3), atol=1e-6)
  assert torch.equal(rows["log_scaling"], torch.log(torch.tensor([[0.5, 2.0, 1.0]])))   # caller's rows left alone

  # general rotation: children are symmetric about the parent along the basis column of the chosen axis
  torch.manual_seed(0)
  pts = TensorRows(position=torch.randn(50, 3), log_scaling=torch.randn(50, 3) * 0.3, rotation=torch.randn(50, 4),
                   alpha_logit=torch.randn(50, 1), feature=torch.randn(50, 3, 4))
  out = split_gaussians_uniform(pts, k=2, random_axis=False)
  axis = torch.argmax(pts["log_scaling"], dim=1)
  sigma = pts["log_scaling"].exp().gather(1, axis[:, None]).squeeze(1)
  d = out["position"][1::2] - out["position"][0::2]
  assert torch.allclose(d.norm(dim=1), 1.4 * sigma / math.sqrt(2.0), rtol=1e-4)
  col = point_basis(pts["log_scaling"], pts["rotation"])[torch.arange(50), :, axis]
  assert torch.allclose(F.normalize(d, dim=1), F.normalize(col, dim=1), atol=1e-5)
  assert torch.allclose(0.5 * (out["position"][0::2] + out["position"][1::2]), pts["position"], atol=1e-5)
  assert out["feature"].shape == (100, 3, 4)
