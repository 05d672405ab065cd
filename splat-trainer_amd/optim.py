"""Per-point parameter container + sparse optimizers, the consumer right after the rasterizer path.

Host-side mirror of the ``taichi_splatting.optim`` names the reference uses -- ``ParameterClass``,
``VisibilityOptimizer``, ``VisibilityAwareLaProp``, ``VisibilityAwareAdam``, ``SparseAdam``, ``SparseLaProp``
(imports: splat_trainer/scene/mlp_scene.py:21, scene/util.py:4; use: mlp_scene.py:58-60, 79, 85, 146, 157-159, 183,
214-230, 306-310, 317; controller/mcmc_controller.py:71) -- over the fused HIP step in ``csrc/optim.hip``.
The arithmetic is specified in ``oracle/optim_oracle.py`` (published Adam / LaProp; the visibility weighting is this
build's own definition: the reference's lives in a package that is not in its tree -- parity unpinned).

tensordict is not available here: ``tensors`` is a ``TensorRows`` (tensor_rows.py: a dict of (N, ...) tensors with the
``select`` / ``replace`` / ``to_dict`` / ``apply`` calls the reference makes on it, mlp_scene.py:296,395-396).  Entries named in
``parameter_groups`` are optimised; the others (e.g. ``visible``, mlp_scene.py:75) only ride along through indexing
and appending.  There is no CPU fallback: ``step`` needs the HIP library and CUDA tensors.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch

from . import _lib
from .tensor_rows import TensorRows

SCALAR, VECTOR, LOCAL_VECTOR = "scalar", "vector", "local_vector"
_TYPE_ID = {SCALAR: 0, VECTOR: 1, LOCAL_VECTOR: 2}


class SparseAdam:
  """Adam (Kingma & Ba) on the rows given to ``step``; per-point step counts drive the bias correction."""
  algo = "adam"
  visibility_aware = False


class SparseLaProp(SparseAdam):
  """LaProp (Ziyin et al. 2020): normalise by the second moment first, momentum afterwards."""
  algo = "laprop"


class VisibilityOptimizer:
  """Marker base (mlp_scene.py:226,275): ``step`` takes the per-point visibility of the batch as a weight."""
  visibility_aware = True


class VisibilityAwareAdam(VisibilityOptimizer, SparseAdam):
  algo = "adam"
  visibility_aware = True


class VisibilityAwareLaProp(VisibilityOptimizer, SparseLaProp):
  algo = "laprop"
  visibility_aware = True


def _ptr(t: Optional[torch.Tensor]):
  return None if t is None else t.data_ptr()     # plain int: the prototypes declare c_void_p


def point_basis_rows(log_scaling: torch.Tensor, rotation: torch.Tensor, indexes: Optional[torch.Tensor] = None,
                     eps: float = 1e-4) -> torch.Tensor:
  """``point_basis(log_scaling[indexes], rotation[indexes])`` (splat_trainer/gaussians/split.py:16-20) as one launch:
  (M,3,3) = R(normalised xyzw quaternion) with columns scaled by ``clamp_min(exp(log_scaling), eps)`` -- the ``basis``
  argument of ``ParameterClass.step`` (mlp_scene.py:219-230).  ``indexes`` None: all rows."""
  ls = log_scaling.detach().to(torch.float32).contiguous()
  rot = rotation.detach().to(torch.float32).contiguous()
  if not (ls.is_cuda and rot.is_cuda):
    raise _lib.GsplatHipError("point_basis_rows runs only on a HIP device; there is no CPU fallback")
  M = int(indexes.shape[0]) if indexes is not None else int(ls.shape[0])
  out = torch.empty(M, 3, 3, dtype=torch.float32, device=ls.device)
  _lib.check(_lib.load().gsr_point_basis(_ptr(ls), _ptr(rot), _ptr(indexes.contiguous()) if indexes is not None else None,
                                         M, float(eps), _ptr(out), _lib.current_stream_ptr()), "gsr_point_basis")
  return out


class ParameterClass:
  """Named per-point tensors with one parameter group each, optimizer state stored as same-length columns so that
  masking (``pc[keep_mask]``) and ``append_tensors`` carry it along (mlp_scene.py:301-310)."""

  def __init__(self, tensors: Dict[str, torch.Tensor], parameter_groups: Dict[str, dict], optimizer=SparseAdam,
               betas=(0.9, 0.999), eps: float = 1e-16, vis_beta: float = 0.9, vis_smooth: float = 0.01,
               bias_correction: bool = True, grad_clip: Optional[float] = None, state: Optional[dict] = None):
    sizes = {t.shape[0] for t in tensors.values()}
    if len(sizes) != 1:
      raise ValueError(f"all tensors need the same number of rows, got {sorted(sizes)}")
    unknown = set(parameter_groups) - set(tensors)
    if unknown:
      raise KeyError(f"parameter groups without a tensor: {sorted(unknown)}")
    self.optimizer = optimizer() if isinstance(optimizer, type) else optimizer
    self.options = dict(betas=tuple(betas), eps=eps, vis_beta=vis_beta, vis_smooth=vis_smooth,
                        bias_correction=bias_correction, grad_clip=grad_clip)
    self.parameter_groups = {k: dict(lr=float(v["lr"]), type=v.get("type", SCALAR)) for k, v in parameter_groups.items()}
    for k, grp in self.parameter_groups.items():
      if grp["type"] not in _TYPE_ID:
        raise ValueError(f"group {k}: unknown type {grp['type']!r}")
    cols = {}
    for k, t in tensors.items():
      t = t.detach().contiguous()
      cols[k] = t.requires_grad_(True) if k in self.parameter_groups else t
    self.tensors = TensorRows(cols)
    self._state = state if state is not None else self._new_state(self.num_points)

  # ---------------------------------------------------------------------------------------------------- basics
  @property
  def num_points(self) -> int:
    return next(iter(self.tensors.values())).shape[0]

  @property
  def batch_size(self):
    return (self.num_points,)

  @property
  def device(self):
    return next(iter(self.tensors.values())).device

  def keys(self):
    return self.tensors.keys()

  def detach(self) -> TensorRows:
    """The rows as a plain container of detached tensors, optimizer state left behind
    (``self.points[split_idx].detach()``, mlp_scene.py:303)."""
    return self.tensors.detach()

  def __getattr__(self, name):
    tensors = self.__dict__.get("tensors")
    if tensors is not None and name in tensors:
      return tensors[name]
    raise AttributeError(name)

  def _new_state(self, n: int) -> dict:
    dev = self.device
    groups = {}
    for k, grp in self.parameter_groups.items():
      d = int(math.prod(self.tensors[k].shape[1:]))
      sq = torch.zeros(n, d, device=dev) if grp["type"] == SCALAR else torch.zeros(n, device=dev)
      groups[k] = dict(exp_avg=torch.zeros(n, d, device=dev), exp_avg_sq=sq)
    return dict(step=torch.zeros(n, device=dev), vis_avg=torch.zeros(n, device=dev), groups=groups)

  @property
  def tensor_state(self) -> Dict[str, Dict[str, torch.Tensor]]:
    """Per-parameter optimizer state (mlp_scene.py:183-187 logs histograms of it)."""
    return {k: dict(v) for k, v in self._state["groups"].items()}

  def update_groups(self, **groups) -> Dict[str, float]:
    """mlp_scene.py:146: new learning rates (``name=lr`` or ``name=dict(lr=...)``); returns {name: lr}."""
    for k, v in groups.items():
      if k not in self.parameter_groups:
        raise KeyError(k)
      self.parameter_groups[k]["lr"] = float(v["lr"] if isinstance(v, dict) else v)
    return {k: g["lr"] for k, g in self.parameter_groups.items()}

  def zero_grad(self):
    for k in self.parameter_groups:
      self.tensors[k].grad = None

  # ------------------------------------------------------------------------------------------------------ step
  @torch.no_grad()
  def step(self, indexes: torch.Tensor, visibility: Optional[torch.Tensor] = None, basis: Optional[torch.Tensor] = None):
    """One optimizer step on the rows ``indexes`` (unique, int64).  ``visibility`` (M,) weights the rows for
    visibility-aware optimizers; ``basis`` (M,3,3) is needed by ``local_vector`` groups (mlp_scene.py:219-230)."""
    lib = _lib.load()
    opt, o = self.optimizer, self.options
    if indexes.dtype != torch.int64 or not indexes.is_cuda:
      raise ValueError("indexes must be a CUDA int64 tensor")
    if opt.visibility_aware and visibility is None:
      raise ValueError(f"{type(opt).__name__}.step needs visibility=")
    M = indexes.shape[0]
    if M == 0:
      return
    indexes = indexes.contiguous()
    vis = visibility.to(torch.float32).contiguous() if (opt.visibility_aware and visibility is not None) else None
    if vis is not None and vis.shape[0] != M:
      raise ValueError("visibility and indexes differ in length")
    stream = _lib.current_stream_ptr()
    row_scale = torch.empty(M, 4, dtype=torch.float32, device=indexes.device)
    st = self._state
    _lib.check(lib.gsr_opt_point_weights(_ptr(indexes), _ptr(vis), M, _ptr(st["step"]), _ptr(st["vis_avg"]),
                                         o["betas"][0], o["betas"][1], o["vis_beta"], o["vis_smooth"],
                                         int(o["bias_correction"]), _ptr(row_scale), stream), "gsr_opt_point_weights")
    algo = 1 if opt.algo == "laprop" else 0
    clip = float(o["grad_clip"]) if o["grad_clip"] else 0.0
    for k, grp in self.parameter_groups.items():
      p = self.tensors[k]
      if p.grad is None:
        continue
      if p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
        raise ValueError(f"{k}: parameters must be contiguous float32 CUDA tensors")
      g = p.grad.to(torch.float32).contiguous()
      b = None
      if grp["type"] == LOCAL_VECTOR:
        if basis is None or tuple(basis.shape) != (M, 3, 3):
          raise ValueError(f"{k}: local_vector groups need basis of shape ({M}, 3, 3)")
        b = basis.to(torch.float32).contiguous()
      gs = st["groups"][k]
      D = int(math.prod(p.shape[1:]))
      _lib.check(lib.gsr_opt_step(_ptr(p), _ptr(g), _ptr(gs["exp_avg"]), _ptr(gs["exp_avg_sq"]), _ptr(indexes),
                                  _ptr(row_scale), _ptr(b), M, D, _TYPE_ID[grp["type"]], algo, grp["lr"],
                                  o["betas"][0], o["betas"][1], o["eps"], clip, stream), f"gsr_opt_step({k})")

  # ----------------------------------------------------------------------------------- densify / prune support
  def _like(self, tensors, state) -> "ParameterClass":
    return ParameterClass(tensors, self.parameter_groups, optimizer=self.optimizer, state=state, **self.options)

  @torch.no_grad()
  def __getitem__(self, rows) -> "ParameterClass":
    """Rows by boolean mask or index tensor, optimizer state included (mlp_scene.py:306-308)."""
    st = self._state
    groups = {k: {n: v[rows] for n, v in g.items()} for k, g in st["groups"].items()}
    state = dict(step=st["step"][rows], vis_avg=st["vis_avg"][rows], groups=groups)
    return self._like({k: t.detach()[rows] for k, t in self.tensors.items()}, state)

  @torch.no_grad()
  def append_tensors(self, tensors: Dict[str, torch.Tensor]) -> "ParameterClass":
    """mlp_scene.py:310: new rows start with zero optimizer state."""
    missing = set(self.tensors) - set(tensors)
    if missing:
      raise KeyError(f"append_tensors: missing {sorted(missing)}")
    n_new = next(iter(tensors.values())).shape[0]
    st = self._state
    pad = lambda v: torch.cat([v, v.new_zeros((n_new,) + tuple(v.shape[1:]))])
    groups = {k: {n: pad(v) for n, v in g.items()} for k, g in st["groups"].items()}
    state = dict(step=pad(st["step"]), vis_avg=pad(st["vis_avg"]), groups=groups)
    return self._like({k: torch.cat([t.detach(), tensors[k].to(t.dtype)]) for k, t in self.tensors.items()}, state)

  @torch.no_grad()
  def keep_and_append(self, keep_mask: torch.Tensor, tensors: Dict[str, torch.Tensor]) -> "ParameterClass":
    """``self[keep_mask].append_tensors(tensors)`` (mlp_scene.py:306-310) fused on the device (densify.compact_rows,
    csrc/densify.hip): one ballot/prefix pass turns the mask into destination rows, then ONE gather launch moves every
    column -- parameters, extras and optimizer state -- to its final buffer, copies the appended rows behind the kept
    ones and zero-fills the appended rows of the optimizer state.  One host sync (the kept count sizes the buffers)."""
    from .densify import compact_rows
    missing = set(self.tensors) - set(tensors)
    if missing:
      raise KeyError(f"keep_and_append: missing {sorted(missing)}")
    n_new = next(iter(tensors.values())).shape[0]
    st = self._state
    names = list(self.tensors)
    columns = [(self.tensors[k], tensors[k]) for k in names]
    state_cols = [("step", None, st["step"]), ("vis_avg", None, st["vis_avg"])]
    for gname, g in st["groups"].items():
      state_cols += [(gname, n, v) for n, v in g.items()]
    columns += [(v, None) for _, _, v in state_cols]
    outs = compact_rows(keep_mask, columns, n_tail=n_new)
    new_tensors = dict(zip(names, outs[:len(names)]))
    groups: Dict[str, Dict[str, torch.Tensor]] = {k: {} for k in st["groups"]}
    state = dict(groups=groups)
    for (gname, n, _), out in zip(state_cols, outs[len(names):]):
      if n is None:
        state[gname] = out
      else:
        groups[gname][n] = out
    return self._like(new_tensors, state)

  def state_dict(self) -> dict:
    return dict(tensors={k: t.detach() for k, t in self.tensors.items()}, optimizer_state=self._state,
                parameter_groups=self.parameter_groups)

  @staticmethod
  def from_state_dict(state: dict, optimizer=SparseAdam, **options) -> "ParameterClass":
    """mlp_scene.py:85."""
    return ParameterClass(state["tensors"], state["parameter_groups"], optimizer=optimizer,
                          state=state["optimizer_state"], **options)
