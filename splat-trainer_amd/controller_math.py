"""Harness-side restatement of the reference maths that CONSUMES the rasterizer's per-point outputs.

The reference's controller is out of scope ("untouched"); these few functions
exist so the benchmark/tests can drive the same consumer arithmetic over HIP and oracle outputs and
compare the resulting densification masks bit for bit:

  exp_lerp .................... splat_trainer/util/misc.py:57-59      (pinned by tests/golden/misc_vectors.json)
  PointState.add_rendering ..... splat_trainer/controller/point_state.py:34-50
  PointState.masked_heuristics . splat_trainer/controller/point_state.py:52-57
  take_n ....................... splat_trainer/controller/target_controller.py:150-160
  find_split_prune_indexes ..... splat_trainer/controller/target_controller.py:73-96
"""
from __future__ import annotations

from dataclasses import dataclass

import torch


def exp_lerp(t: float, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
  """log-sum-exp blend: max + log(lerp(exp(a-max), exp(b-max), t))."""
  m = torch.maximum(a, b)
  return m + torch.log(torch.lerp(torch.exp(a - m), torch.exp(b - m), t))


@dataclass
class PointState:
  prune_cost: torch.Tensor
  split_score: torch.Tensor
  max_scale_px: torch.Tensor
  points_in_view: torch.Tensor     # int16
  visibility: torch.Tensor

  @staticmethod
  def new_zeros(num_points: int, device) -> "PointState":
    z = lambda dt=torch.float32: torch.zeros(num_points, dtype=dt, device=device)
    return PointState(z(), z(), z(), z(torch.int16), z())

  def add_rendering(self, rendering, split_alpha: float = 0.01, prune_alpha: float = 0.1, visible_sum=None):
    """point_state.py:34-50.  On the device the five updates run as one fused launch (densify.point_state_add); on the
    CPU (the oracle side of the mask-parity tests) as the reference's torch ops below.  ``visible_sum``: the scene's
    ``visible`` accumulator (mlp_scene.py:244, ``+= visibility`` on the same rows), folded into the same launch."""
    points = rendering.points
    if self.prune_cost.is_cuda:
      from .densify import point_state_add
      point_state_add(self, points, split_alpha, prune_alpha, visible_sum=visible_sum)
      return
    if visible_sum is not None:
      visible_sum.index_add_(0, points.idx, points.visibility)
    image_scale_px = points.screen_scale.max(1).values
    self.max_scale_px[points.idx] = torch.maximum(self.max_scale_px[points.idx], image_scale_px)
    self.points_in_view[points.visible.idx] += 1
    self.visibility[points.idx] += points.visibility
    self.split_score[points.idx] = exp_lerp(split_alpha, self.split_score[points.idx], points.split_score)
    self.prune_cost[points.idx] = exp_lerp(prune_alpha, self.prune_cost[points.idx], points.prune_cost)

  def add_scores(self, idx, split_score: torch.Tensor, prune_cost: torch.Tensor, split_alpha: float = 0.01,
                 prune_alpha: float = 0.1):
    """The order-dependent part of add_rendering alone: the two exp_lerp EMAs over the rows ``idx`` (None: all points).
    Used by the data-parallel exchange, which replays only these per camera and reduces the rest."""
    if self.prune_cost.is_cuda:
      from .densify import point_state_update
      point_state_update(self, idx, split_score=split_score, prune_cost=prune_cost, split_alpha=split_alpha,
                         prune_alpha=prune_alpha)
      return
    rows = slice(None) if idx is None else idx
    self.split_score[rows] = exp_lerp(split_alpha, self.split_score[rows], split_score)
    self.prune_cost[rows] = exp_lerp(prune_alpha, self.prune_cost[rows], prune_cost)

  def masked_heuristics(self, min_views: int):
    seen = self.points_in_view >= min_views
    prune_cost = torch.where(seen, self.prune_cost, torch.full_like(self.prune_cost, torch.inf))
    split_score = torch.where(seen, self.split_score, torch.zeros_like(self.split_score))
    return prune_cost, split_score


def take_n(t: torch.Tensor, n: int, descending: bool = False) -> torch.Tensor:
  """Bool mask of the n smallest (largest) entries (target_controller.py:150-160).  The reference's non-stable argsort
  leaves the choice among equal values (zeros and infs by the thousand) to the sort implementation; here ties go to the
  lowest indexes.  On the device this is the radix-select kernel (densify.select_n); on the CPU -- the oracle side of
  the mask-parity tests -- a stable argsort, which makes the same choice."""
  assert n >= 0, f"n must be >= 0, got {n}"
  if t.is_cuda:
    from .densify import select_n
    return select_n(t, n, descending=descending)
  idx = torch.argsort(t, descending=descending, stable=True)[:n]
  mask = torch.zeros_like(t, dtype=torch.bool)
  mask[idx] = True
  return mask


def find_split_prune_indexes(state: PointState, t: float, target_points: int, prune_rate: float = 0.025,
                             min_views: int = 5, max_scale_px: float = 200.0, min_split_px: float = 0.0):
  """-> (split_mask, prune_mask), as target_controller.py:73-96 (defaults: config/controller/target.yaml)."""
  import math
  n = state.prune_cost.shape[0]
  exceeds_scale = state.max_scale_px > max_scale_px
  prune_schedule = int(math.ceil(prune_rate * n * (1 - t)))
  prune_cost, _ = state.masked_heuristics(min_views)
  prune_mask = take_n(prune_cost, prune_schedule, descending=False) | exceeds_scale
  target_split = max((target_points - n) + int(prune_mask.sum().item()), 0)
  split_score = state.split_score.clone()
  split_score[prune_mask] = 0.
  if min_split_px > 0:
    split_score[state.max_scale_px < min_split_px] = 0.
  split_mask = take_n(split_score, min(target_split, n), descending=True)
  both = split_mask & prune_mask
  return split_mask ^ both, prune_mask ^ both
