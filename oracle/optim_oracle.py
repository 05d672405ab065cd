"""CPU oracle for the sparse optimizer step next to the rasterizer path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED for the visibility-aware part.  The reference steps its per-point parameters with
``taichi_splatting.optim.ParameterClass`` / ``VisibilityAwareLaProp`` (call site splat_trainer/scene/mlp_scene.py:214-230,
options :58-60, groups and types config/scene/mlp.yaml:8-14); that package (``taichi-splatting >= 0.31.0``,
/root/reference/pyproject.toml:16) is neither vendored nor installed, and the reference holds no optimizer fixtures.
What is pinned: with ``visibility=None`` the Adam form below is the published algorithm (Kingma & Ba 2015) and is
checked against ``torch.optim.Adam`` on the visited rows (tests/test_optim_oracle.py); the LaProp form is the
published one (Ziyin, Wang & Ueda 2020: normalise the gradient first, then take the momentum).

Specification (one step over the rows ``indexes`` (unique), per-point weight ``w = visibility`` or 1):

  shared per point      t        = step[idx] + 1                       (rows seen so far, the bias-correction clock)
                        avg      = vis_beta * vis_avg[idx] + (1 - vis_beta) * w
                        inv_w    = 1 / (w + vis_smooth)                (the reference logs grad * inv_w as its
                                                                        "norm_grad", mlp_scene.py:177)
                        rho      = w / (avg / (1 - vis_beta^t) + vis_smooth)   (visibility relative to the point's norm)
                        (visibility None: inv_w = rho = 1 and vis_avg is untouched)
  per group             g        = grad[idx] * inv_w                   (per-unit-visibility gradient)
    local_vector        g        = B^T g,  B = basis (R diag(scale)), mlp_scene.py:219
    second moment       v        = beta2 v + (1 - beta2) s,  s = g^2 (scalar) or mean_j g_j^2 (vector, local_vector)
    LaProp              u        = clamp(g / (sqrt(v / (1 - beta2^t)) + eps), +-grad_clip);  m = beta1 m + (1 - beta1) u
                        dec      = lr * rho * m / (1 - beta1^t)
    Adam                m        = beta1 m + (1 - beta1) g
                        dec      = lr * rho * (m / (1 - beta1^t)) / (sqrt(v / (1 - beta2^t)) + eps)
    local_vector        dec      = B dec
                        param[idx] -= dec
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

SCALAR, VECTOR, LOCAL_VECTOR = "scalar", "vector", "local_vector"


def new_state(tensors: Dict[str, torch.Tensor], types: Dict[str, str]) -> dict:
  n = next(iter(tensors.values())).shape[0]
  ref = next(iter(tensors.values()))
  state = dict(step=torch.zeros(n, dtype=ref.dtype, device=ref.device),
               vis_avg=torch.zeros(n, dtype=ref.dtype, device=ref.device), groups={})
  for name, p in tensors.items():
    flat = p.reshape(n, -1)
    sq = torch.zeros_like(flat) if types.get(name, SCALAR) == SCALAR else torch.zeros(n, dtype=p.dtype, device=p.device)
    state["groups"][name] = dict(exp_avg=torch.zeros_like(flat), exp_avg_sq=sq)
  return state


@torch.no_grad()
def step(tensors: Dict[str, torch.Tensor], grads: Dict[str, Optional[torch.Tensor]], state: dict, lrs: Dict[str, float],
         types: Dict[str, str], indexes: torch.Tensor, visibility: Optional[torch.Tensor] = None,
         basis: Optional[torch.Tensor] = None, algo: str = "laprop", betas=(0.9, 0.999), eps: float = 1e-16,
         vis_beta: float = 0.9, vis_smooth: float = 0.01, bias_correction: bool = True,
         grad_clip: Optional[float] = None) -> None:
  """In-place step of ``tensors`` (name -> (N, ...)) on rows ``indexes``."""
  beta1, beta2 = betas
  idx = indexes
  t = state["step"][idx] + 1
  state["step"][idx] = t
  if visibility is not None:
    w = visibility.to(t.dtype)
    avg = vis_beta * state["vis_avg"][idx] + (1 - vis_beta) * w
    state["vis_avg"][idx] = avg
    avg_hat = avg / (1 - vis_beta ** t) if bias_correction else avg
    inv_w = 1 / (w + vis_smooth)
    rho = w / (avg_hat + vis_smooth)
  else:
    inv_w = torch.ones_like(t)
    rho = torch.ones_like(t)
  bc1 = 1 - beta1 ** t if bias_correction else torch.ones_like(t)
  bc2 = 1 - beta2 ** t if bias_correction else torch.ones_like(t)
  for name, p in tensors.items():
    grad = grads.get(name)
    if grad is None:
      continue
    n = p.shape[0]
    flat = p.reshape(n, -1)
    kind = types.get(name, SCALAR)
    g = grad.reshape(n, -1)[idx] * inv_w[:, None]
    if kind == LOCAL_VECTOR:
      g = torch.einsum("mrk,mr->mk", basis.to(g.dtype), g)
    st = state["groups"][name]
    if kind == SCALAR:
      v = beta2 * st["exp_avg_sq"][idx] + (1 - beta2) * g * g
      st["exp_avg_sq"][idx] = v
      second = v
    else:
      v = beta2 * st["exp_avg_sq"][idx] + (1 - beta2) * (g * g).mean(dim=1)
      st["exp_avg_sq"][idx] = v
      second = v[:, None]
    denom = torch.sqrt(second / bc2[:, None]) + eps
    m = st["exp_avg"][idx]
    if algo == "laprop":
      u = g / denom
      if grad_clip is not None and grad_clip > 0:
        u = u.clamp(-grad_clip, grad_clip)
      m = beta1 * m + (1 - beta1) * u
      dec = lrs[name] * (rho / bc1)[:, None] * m
    else:
      m = beta1 * m + (1 - beta1) * g
      dec = lrs[name] * (rho / bc1)[:, None] * m / denom
    st["exp_avg"][idx] = m
    if kind == LOCAL_VECTOR:
      dec = torch.einsum("mrk,mk->mr", basis.to(g.dtype), dec)
    flat[idx] = flat[idx] - dec
