"""Minimal harness around the rasterizer path for BASELINE config c4: a short training loop with the
reference's densify/prune arithmetic, so that the dynamic-splat-count use of the boundary is exercised
end to end.  It restates, for the harness only, the consumers the reference keeps untouched:

  evaluate_backward_with ........ splat_trainer/trainer/trainer.py:500-514  (per-camera render + loss + backward,
                                   gradients accumulate across the cameras of a batch)
  PointState / TargetController .. splat_trainer/controller/point_state.py:34-110,
                                   splat_trainer/controller/target_controller.py:73-126   (controller_math.py)
  split_gaussians_uniform ........ splat_trainer/gaussians/split.py:87-113  (k = 2, +/-0.7 sigma along a sampled
                                   axis, that axis scaled by 1/sqrt(2))
  scene.split_and_prune .......... splat_trainer/scene/mlp_scene.py:301-310 (keep_mask rows + appended splits)
  scene.step ..................... mlp_scene.py:214-239 (visible rows -> basis -> points.step(visibility, indexes,
                                   basis); renormalise quaternions, clamp log_scaling to [-8, 8]; zero_grad)
  scene.add_rendering ............ mlp_scene.py:241-244 (visible[idx] += visibility)

The optimizer is optim.ParameterClass with VisibilityAwareLaProp and the reference's group types
(config/scene/mlp.yaml:8-14).  Nothing here is on the measured hot path.
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from .controller_math import PointState, find_split_prune_indexes
from .data_types import CameraParams, Gaussians3D, RasterConfig
from .optim import ParameterClass, VisibilityAwareLaProp, point_basis_rows
from .tensor_rows import TensorRows
from .loss import clamped_mse_loss, reference_loss
from .renderer import GradOut, render_gaussians

PARAM_NAMES = ("position", "log_scaling", "rotation", "alpha_logit", "feature")


def quat_to_rotmat_xyzw(q: torch.Tensor) -> torch.Tensor:
  q = F.normalize(q, dim=1)
  x, y, z, w = q.unbind(-1)
  return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                      2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                      2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=-1).reshape(-1, 3, 3)


def point_basis(log_scaling: torch.Tensor, rotation: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
  """split.py:16-20: R(q) * clamp_min(exp(log_s), eps) (columns scaled)."""
  return quat_to_rotmat_xyzw(rotation) * torch.clamp_min(torch.exp(log_scaling), eps).unsqueeze(-2)


def split_gaussians_uniform(points: TensorRows, k: int = 2, scaling: Optional[float] = None, sep: float = 0.7,
                            random_axis: bool = False, eps: float = 1e-4,
                            generator: Optional[torch.Generator] = None) -> TensorRows:
  """gaussians/split.py:87-113 over a TensorRows: every parent becomes ``k`` children spread over [-sep, sep] sigma
  along one axis (sampled in proportion to the scales, or the largest), and that axis shrinks by ``scaling`` (default
  1/sqrt(k)).  The reference shrinks first -- ``points.update(...)`` is in place, so the ``point_basis`` that places the
  children already sees the shrunk axis -- hence the children sit at +/- sep * scaling * sigma, not +/- sep * sigma.
  All columns of ``points`` (extras such as ``visible`` included) are repeated k times, as split_with_offsets does."""
  points = points if isinstance(points, TensorRows) else TensorRows(points)
  ls = points["log_scaling"]
  if random_axis:
    weights = F.normalize(torch.clamp_min(ls.exp(), eps), dim=1)
    pick = torch.multinomial(weights, num_samples=1, generator=generator).squeeze(1)
  else:
    pick = torch.argmax(ls, dim=1)
  axis = F.one_hot(pick, num_classes=3).to(ls.dtype)                              # (n, 3)
  scaling = 1.0 / math.sqrt(k) if scaling is None else scaling
  shrunk = ls + math.log(scaling) * axis
  steps = torch.linspace(-sep, sep, k, device=ls.device, dtype=ls.dtype)          # (k,)
  local = steps.view(1, k, 1) * axis.view(-1, 1, 3)                               # (n, k, 3) in splat space
  offsets = torch.einsum("nij,nkj->nki", point_basis(shrunk, points["rotation"], eps), local)
  children = points.replace(log_scaling=shrunk).apply(lambda t: t.repeat_interleave(k, dim=0))
  return children.update(position=children["position"] + offsets.reshape(-1, 3))


@dataclass
class TrainLog:
  losses: List[float] = field(default_factory=list)
  num_points: List[int] = field(default_factory=list)
  mask_digests: List[str] = field(default_factory=list)


class MiniTrainer:
  """Replicated-parameter training loop: batch of cameras -> grads -> sparse visibility-aware LaProp step -> (every
  ``densify_every`` its) TargetController-style split/prune with buffer re-sizing (optimizer state rides along)."""

  def __init__(self, gaussians: Gaussians3D, cameras: Sequence[CameraParams], targets: Sequence[torch.Tensor],
               config: Optional[RasterConfig] = None, lr: float = 1e-3, densify_every: int = 25,
               target_points: Optional[int] = None, prune_rate: float = 0.025, min_views: int = 5,
               max_scale_px: float = 200.0, total_steps: int = 100, seed: int = 0, optimizer=VisibilityAwareLaProp,
               loss: str = "mse"):
    self.config = config or RasterConfig(compute_visibility=True, compute_point_heuristic=True)
    self.cameras, self.targets = list(cameras), list(targets)
    self.device = gaussians.position.device
    tensors = {n: getattr(gaussians, n).detach().clone() for n in PARAM_NAMES}
    tensors["visible"] = torch.zeros(gaussians.position.shape[0], device=self.device)       # mlp_scene.py:75
    # LaProp steps are ~lr per iteration whatever the gradient scale; ratios between groups as in mlp.yaml:8-14
    groups = dict(position=dict(lr=30 * lr, type="local_vector"), log_scaling=dict(lr=8 * lr),
                  rotation=dict(lr=1 * lr, type="vector"), alpha_logit=dict(lr=10 * lr), feature=dict(lr=4 * lr))
    self.points = ParameterClass(tensors, groups, optimizer=optimizer, betas=(0.8, 0.95), vis_beta=0.999,
                                 vis_smooth=0.01, bias_correction=True, grad_clip=2.0)      # mlp.yaml:25-31
    self.lr, self.densify_every, self.total_steps = lr, densify_every, total_steps
    self.target_points = target_points or int(1.1 * gaussians.position.shape[0])
    self.prune_rate, self.min_views, self.max_scale_px = prune_rate, min_views, max_scale_px
    self.gen = torch.Generator(device=self.device).manual_seed(seed)
    self.state = PointState.new_zeros(self.num_points, self.device)
    self.step_idx = 0
    self.log = TrainLog()
    if loss not in ("mse", "ref"):
      raise ValueError("loss must be 'mse' (clamped MSE, SURVEY.md section 8d) or 'ref' (the reference's L1 + MSE + SSIM mix)")
    self.loss_kind = loss

  @property
  def num_points(self) -> int:
    return self.points.num_points

  @property
  def params(self) -> dict:
    return {n: self.points.tensors[n] for n in PARAM_NAMES}

  def scene(self) -> Gaussians3D:
    return Gaussians3D(**self.params)

  @torch.no_grad()
  def optimizer_step(self):
    """mlp_scene.py:214-239."""
    pts = self.points
    vis_idx = pts.visible.nonzero().squeeze(1)          # (the iteration's one host wait: the row count sizes the step)
    if vis_idx.shape[0] == 0:
      raise RuntimeError("No visible points")           # trainer.py:507-509, for the batch as a whole
    if pts.log_scaling.is_cuda:          # one launch instead of the expression's dozen (0.1 instead of 1 ms at 3 M rows)
      basis = point_basis_rows(pts.log_scaling, pts.rotation, vis_idx)
    else:
      basis = point_basis(pts.log_scaling.detach()[vis_idx], pts.rotation.detach()[vis_idx]).contiguous()
    if pts.optimizer.visibility_aware:
      pts.step(visibility=pts.visible[vis_idx], indexes=vis_idx, basis=basis)
    else:
      pts.step(indexes=vis_idx, basis=basis)
    pts.rotation.data = F.normalize(pts.rotation.data, dim=1)
    pts.log_scaling.data.clamp_(min=-8, max=8)
    pts.visible.zero_()
    pts.zero_grad()

  def _grad_target(self) -> GradOut:
    """The parameters' ``.grad`` tensors as the accumulation target of the backward kernels (renderer.GradOut): the
    reference sums the cameras of a batch into ``.grad`` through autograd (trainer.py:500-514, zero_grad only in
    scene.step); here the kernels add their rows straight into the same tensors, which spares every camera the dense
    N-sized temporaries and autograd's accumulate pass over them.  Nothing is zero-filled: the first camera's backward
    pass writes every row (zeros where it saw nothing), the later ones add."""
    grads = {}
    for n in PARAM_NAMES:
      p = self.points.tensors[n]
      if p.grad is None or p.grad.shape != p.shape:
        p.grad = torch.empty_like(p)
      grads[n] = p.grad
    return GradOut(feature_uninitialized=True, geometry_uninitialized=True, **grads)

  def training_step(self) -> float:
    """trainer.py:531-545: evaluate_backward_with over the batch, then the optimizer step.  The host waits for the
    device ONCE per iteration (the optimizer's ``visible.nonzero()``): the per-camera losses are added up on the device
    and read afterwards, and the reference's "no visible points" guard (trainer.py:507-509, ``points.num_visible`` -- a
    reduction and a read-back per camera, in front of the backward pass) is split into its host-known half here (no
    point in view: M = 0) and the batch-wide half in ``optimizer_step`` (no point of the batch has visibility > 0)."""
    fused = self.device.type == "cuda"
    grad_out = self._grad_target() if fused else None
    total = torch.zeros((), dtype=torch.float32, device=self.device)
    for cam, target in zip(self.cameras, self.targets):
      with torch.enable_grad():
        r = render_gaussians(self.scene(), cam, self.config, use_sh=True, grad_out=grad_out)
        if r.points.idx.shape[0] == 0:
          raise RuntimeError("No visible points")                     # trainer.py:507-509
        if self.loss_kind == "ref":
          loss = reference_loss(r.image, target)                      # trainer.py:448-488: L1 + MSE + multi-scale SSIM
        else:
          loss = clamped_mse_loss(r.image, target) if fused else F.mse_loss(r.image.clamp(0, 1), target)
        loss.backward()
      with torch.no_grad():
        # point_state.py:34-50 (camera order) and mlp_scene.py:244 (visible[idx] += visibility) in one launch
        self.state.add_rendering(r, visible_sum=self.points.visible)
        total += loss.detach()
    if grad_out is not None:
      grad_out.finish_batch()               # (a buffer no backward pass reached would be zero-filled here; none with >= 1 camera)
    self.optimizer_step()
    self.step_idx += 1
    self.log.losses.append(float(total.item()) / len(self.cameras))
    self.log.num_points.append(self.num_points)
    if self.densify_every and self.step_idx % self.densify_every == 0 and self.step_idx < self.total_steps:
      self.densify_and_prune()
    return self.log.losses[-1]

  @torch.no_grad()
  def densify_and_prune(self):
    t = self.step_idx / self.total_steps
    split_mask, prune_mask = find_split_prune_indexes(self.state, t, self.target_points, self.prune_rate,
                                                      self.min_views, self.max_scale_px)
    digest = hashlib.sha256(torch.cat([split_mask, prune_mask]).cpu().numpy().tobytes()).hexdigest()
    self.log.mask_digests.append(digest)
    keep_mask = ~(split_mask | prune_mask)
    split_idx = split_mask.nonzero().squeeze(1)
    # mlp_scene.py:301-310: children from the split rows, kept rows (with their optimizer state), children appended
    splits = split_gaussians_uniform(self.points[split_idx].detach(), k=2, random_axis=True, generator=self.gen)
    self.points = self.points.keep_and_append(keep_mask, splits)
    self.state = PointState.new_zeros(self.num_points, self.device)   # target_controller.py:120-122

  def train(self, steps: int) -> TrainLog:
    for _ in range(steps):
      self.training_step()
    return self.log
