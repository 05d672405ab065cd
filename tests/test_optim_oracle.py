"""CPU checks of oracle/optim_oracle.py: the sparse Adam form against torch.optim.Adam (the published algorithm as torch
implements it), LaProp against its definition written out by hand, and the group types."""
import math

import torch

from helpers import oracle_optim as oo


def _problem(n=40, seed=0):
  g = torch.Generator().manual_seed(seed)
  tensors = dict(position=torch.randn(n, 3, generator=g, dtype=torch.float64),
                 alpha_logit=torch.randn(n, 1, generator=g, dtype=torch.float64),
                 feature=torch.randn(n, 3, 4, generator=g, dtype=torch.float64))
  grads = lambda s: {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(100 + s), dtype=torch.float64)
                     for k, v in tensors.items()}
  return tensors, grads


def test_sparse_adam_equals_torch_adam_on_all_rows():
  tensors, grads = _problem()
  types = {k: oo.SCALAR for k in tensors}
  lrs = dict(position=0.01, alpha_logit=0.05, feature=0.002)
  ref = {k: v.clone().requires_grad_(True) for k, v in tensors.items()}
  opt = torch.optim.Adam([dict(params=[ref[k]], lr=lrs[k]) for k in ref], betas=(0.8, 0.95), eps=1e-12)
  state = oo.new_state(tensors, types)
  idx = torch.arange(40)
  for s in range(5):
    gr = grads(s)
    for k in ref:
      ref[k].grad = gr[k].clone()
    opt.step()
    oo.step(tensors, gr, state, lrs, types, idx, algo="adam", betas=(0.8, 0.95), eps=1e-12)
  for k in tensors:
    assert torch.allclose(tensors[k], ref[k].detach(), rtol=1e-12, atol=1e-14), k


def test_sparse_rows_keep_their_own_clock():
  """A row visited in steps 0 and 2 must match a dense Adam that only ever saw those two gradients."""
  tensors, grads = _problem()
  types = {k: oo.SCALAR for k in tensors}
  lrs = {k: 0.01 for k in tensors}
  state = oo.new_state(tensors, types)
  before = {k: v.clone() for k, v in tensors.items()}
  rows = [torch.tensor([1, 5, 7]), torch.tensor([2, 5]), torch.tensor([7, 1])]
  for s, idx in enumerate(rows):
    oo.step(tensors, grads(s), state, lrs, types, idx, algo="adam")
  ref = before["position"][1:2].clone().requires_grad_(True)
  opt = torch.optim.Adam([ref], lr=0.01, eps=1e-16)
  for s in (0, 2):
    ref.grad = grads(s)["position"][1:2].clone()
    opt.step()
  assert torch.allclose(tensors["position"][1], ref.detach()[0], rtol=1e-12)
  assert torch.equal(tensors["position"][0], before["position"][0])          # never visited: untouched
  assert state["step"].tolist()[:8] == [0, 2, 1, 0, 0, 2, 0, 2]


def test_laprop_by_hand_with_clip_and_visibility():
  p0, g, w = 0.7, 3.0, 0.25
  tensors = dict(x=torch.tensor([[p0]], dtype=torch.float64))
  types, lrs = dict(x=oo.SCALAR), dict(x=0.1)
  state = oo.new_state(tensors, types)
  b1, b2, vb, sm, eps, clip = 0.8, 0.95, 0.9, 0.01, 1e-16, 2.0
  oo.step(tensors, dict(x=torch.tensor([[g]], dtype=torch.float64)), state, lrs, types, torch.tensor([0]),
          visibility=torch.tensor([w], dtype=torch.float64), algo="laprop", betas=(b1, b2), eps=eps, vis_beta=vb,
          vis_smooth=sm, grad_clip=clip)
  gn = g / (w + sm)
  v = (1 - b2) * gn * gn
  u = min(max(gn / (math.sqrt(v / (1 - b2)) + eps), -clip), clip)        # = +1 -> not clipped at t = 1
  m = (1 - b1) * u
  avg = (1 - vb) * w
  rho = w / (avg / (1 - vb) + sm)
  want = p0 - 0.1 * rho * m / (1 - b1)
  assert abs(float(tensors["x"][0, 0]) - want) < 1e-14
  assert abs(u - 1.0) < 1e-12 and abs(rho - w / (w + sm)) < 1e-15


def test_vector_and_local_vector_groups():
  g = torch.Generator().manual_seed(3)
  n = 6
  tensors = dict(position=torch.randn(n, 3, generator=g, dtype=torch.float64),
                 rotation=torch.randn(n, 4, generator=g, dtype=torch.float64))
  grads = dict(position=torch.randn(n, 3, generator=g, dtype=torch.float64),
               rotation=torch.randn(n, 4, generator=g, dtype=torch.float64))
  types = dict(position=oo.LOCAL_VECTOR, rotation=oo.VECTOR)
  lrs = dict(position=0.3, rotation=0.01)
  state = oo.new_state(tensors, types)
  before = {k: v.clone() for k, v in tensors.items()}
  idx = torch.arange(n)
  # an isotropic basis s*I: the local_vector update is the vector update scaled by s (g -> s g leaves g/sqrt(v)
  # unchanged, the step is then multiplied by s)
  s = 0.05
  basis = (s * torch.eye(3, dtype=torch.float64)).expand(n, 3, 3).contiguous()
  oo.step(tensors, grads, state, lrs, types, idx, basis=basis, algo="laprop", betas=(0.8, 0.95))
  # first LaProp step: u = g / sqrt(mean g^2), m = (1-b1) u, dec = lr * u
  for k, scale in (("position", s), ("rotation", 1.0)):
    u = grads[k] / grads[k].pow(2).mean(dim=1, keepdim=True).sqrt()
    assert torch.allclose(before[k] - tensors[k], lrs[k] * scale * u, rtol=1e-9), k
  assert state["groups"]["rotation"]["exp_avg_sq"].shape == (n,)
