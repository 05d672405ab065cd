"""Sparse optimizer step: csrc/optim.hip through ParameterClass vs oracle/optim_oracle.py.

The call pattern is the reference's scene.step (splat_trainer/scene/mlp_scene.py:214-237): visible rows from the
``visible`` accumulator, basis from log_scaling / rotation of those rows, ``points.step(visibility=, indexes=, basis=)``.
Tolerance: fp32 kernels vs the fp64 oracle, 2e-5 relative to the largest magnitude of each tensor after 4 steps."""
import pytest
import torch

import splat_trainer_amd as sta
from helpers import oracle_optim as oo, rel_err
from splat_trainer_amd import optim
from splat_trainer_amd.harness import point_basis

pytestmark = pytest.mark.gpu

GROUPS = dict(position=dict(lr=0.3, type="local_vector"), log_scaling=dict(lr=0.08),
              rotation=dict(lr=0.01, type="vector"), alpha_logit=dict(lr=0.1),
              feature=dict(lr=5.0, type="vector"))                     # config/scene/mlp.yaml:8-14


def _tensors(n, feature_shape, seed):
  g = torch.Generator().manual_seed(seed)
  return dict(position=torch.randn(n, 3, generator=g), log_scaling=torch.randn(n, 3, generator=g) * 0.5 - 3,
              rotation=torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=1),
              alpha_logit=torch.randn(n, 1, generator=g), feature=torch.randn(n, *feature_shape, generator=g),
              visible=torch.zeros(n))


def _run(opt_cls, algo, feature_shape, groups, steps=4, n=3001, grad_clip=2.0, seed=0):
  cpu = _tensors(n, feature_shape, seed)
  options = dict(betas=(0.8, 0.95), vis_beta=0.999, vis_smooth=0.01, bias_correction=True, grad_clip=grad_clip)
  pc = optim.ParameterClass({k: v.cuda() for k, v in cpu.items()}, groups, optimizer=opt_cls, **options)
  ref = {k: cpu[k].double().clone() for k in groups}
  types = {k: groups[k].get("type", "scalar") for k in groups}
  lrs = {k: groups[k]["lr"] for k in groups}
  state = oo.new_state(ref, types)
  g = torch.Generator().manual_seed(seed + 1)
  vis_aware = opt_cls.visibility_aware
  for s in range(steps):
    visible = torch.rand(n, generator=g) * (torch.rand(n, generator=g) < 0.6)        # ~40 % of the rows unseen
    grads = {k: torch.randn(cpu[k].shape, generator=g) * visible.view(-1, *[1] * (cpu[k].dim() - 1)) for k in groups}
    pc.visible.copy_(visible)
    for k in groups:
      pc.tensors[k].grad = grads[k].cuda()
    vis_idx = pc.visible.nonzero().squeeze(1)                                       # mlp_scene.py:217
    basis = point_basis(pc.log_scaling[vis_idx].detach(), pc.rotation[vis_idx].detach()).contiguous()
    if vis_aware:
      pc.step(visibility=pc.visible[vis_idx], indexes=vis_idx, basis=basis)
    else:
      pc.step(indexes=vis_idx, basis=basis)
    idx = visible.nonzero().squeeze(1)
    obasis = point_basis(ref["log_scaling"][idx], ref["rotation"][idx])
    oo.step(ref, {k: v.double() for k, v in grads.items()}, state, lrs, types, idx,
            visibility=visible[idx].double() if vis_aware else None, basis=obasis, algo=algo,
            grad_clip=grad_clip, **{k: options[k] for k in ("betas", "vis_beta", "vis_smooth", "bias_correction")})
    pc.zero_grad()
  for k in groups:
    assert rel_err(pc.tensors[k].detach().cpu().double(), ref[k]) < 2e-5, k
    st = pc.tensor_state[k]
    assert rel_err(st["exp_avg"].cpu().double(), state["groups"][k]["exp_avg"]) < 2e-5, k
    assert rel_err(st["exp_avg_sq"].cpu().double(), state["groups"][k]["exp_avg_sq"]) < 2e-5, k
  assert torch.equal(pc._state["step"].cpu().double(), state["step"])
  return pc


@pytest.mark.parametrize("opt_cls,algo", [(optim.VisibilityAwareLaProp, "laprop"), (optim.VisibilityAwareAdam, "adam"),
                                          (optim.SparseAdam, "adam"), (optim.SparseLaProp, "laprop")])
def test_step_matches_oracle(opt_cls, algo):
  _run(opt_cls, algo, (16,), GROUPS)


def test_wide_scalar_rows_and_sh_shaped_features():
  groups = dict(GROUPS, feature=dict(lr=0.05, type="scalar"))
  _run(optim.VisibilityAwareLaProp, "laprop", (3, 16), groups)                       # D = 48, scalar second moments
  _run(optim.SparseAdam, "adam", (3, 9), dict(GROUPS, feature=dict(lr=0.05, type="vector")), grad_clip=None)


def test_unseen_rows_are_untouched_and_mask_append_carry_state():
  pc = _run(optim.VisibilityAwareLaProp, "laprop", (16,), GROUPS, steps=2, n=500)
  never = (pc._state["step"] == 0).nonzero().squeeze(1)
  assert never.numel() > 0
  fresh = _tensors(500, (16,), 0)
  assert torch.equal(pc.position.detach()[never].cpu(), fresh["position"][never.cpu()])
  assert float(pc.tensor_state["feature"]["exp_avg"][never].abs().max()) == 0.0
  keep = torch.rand(500, device="cuda") < 0.7                                        # mlp_scene.py:306-310
  kept = pc[keep]
  assert kept.num_points == int(keep.sum()) and torch.equal(kept._state["step"], pc._state["step"][keep])
  assert torch.equal(kept.tensor_state["rotation"]["exp_avg_sq"], pc.tensor_state["rotation"]["exp_avg_sq"][keep])
  extra = {k: v[:10].detach().clone() for k, v in pc.tensors.items()}
  grown = kept.append_tensors(extra)
  assert grown.num_points == kept.num_points + 10 and grown.position.requires_grad and not grown.visible.requires_grad
  assert float(grown._state["step"][-10:].abs().max()) == 0.0
  assert float(grown.tensor_state["position"]["exp_avg"][-10:].abs().max()) == 0.0
  fused = pc.keep_and_append(keep, extra)                                              # same result in one pass
  for k in grown.tensors:
    assert torch.equal(fused.tensors[k], grown.tensors[k]), k
  for k in GROUPS:
    for n in ("exp_avg", "exp_avg_sq"):
      assert torch.equal(fused.tensor_state[k][n], grown.tensor_state[k][n]), (k, n)
  assert torch.equal(fused._state["step"], grown._state["step"]) and fused.position.requires_grad
  assert grown.update_groups(position=0.1, feature=dict(lr=1.0))["position"] == 0.1
  restored = optim.ParameterClass.from_state_dict(grown.state_dict(), optimizer=optim.VisibilityAwareLaProp)
  assert torch.equal(restored.position, grown.position) and restored.num_points == grown.num_points


def test_step_argument_errors():
  pc = optim.ParameterClass({k: v.cuda() for k, v in _tensors(8, (4,), 1).items()}, GROUPS,
                            optimizer=optim.VisibilityAwareLaProp)
  for k in GROUPS:
    pc.tensors[k].grad = torch.ones_like(pc.tensors[k])
  idx = torch.arange(8, device="cuda")
  with pytest.raises(ValueError):
    pc.step(indexes=idx)                                                              # visibility missing
  with pytest.raises(ValueError):
    pc.step(indexes=idx, visibility=torch.ones(8, device="cuda"))                     # basis missing
  with pytest.raises(ValueError):
    pc.step(indexes=idx.int(), visibility=torch.ones(8, device="cuda"))


def test_point_basis_rows_matches_the_torch_expression():
  """optim.point_basis_rows (one launch) against harness.point_basis, the restated split.py:16-20 expression."""
  from splat_trainer_amd.harness import point_basis
  from splat_trainer_amd.optim import point_basis_rows
  gen = torch.Generator().manual_seed(4)
  n = 5000
  ls = (torch.randn(n, 3, generator=gen) * 2 - 3).cuda()
  ls[:5] = -20.0                                             # exp below eps: the clamp decides
  rot = torch.randn(n, 4, generator=gen).cuda()              # deliberately not normalised
  idx = torch.randperm(n, generator=gen)[:1234].sort().values.cuda()
  for rows in (None, idx):
    got = point_basis_rows(ls, rot, rows)
    want = point_basis(ls if rows is None else ls[rows], rot if rows is None else rot[rows])
    assert got.shape == want.shape
    # entries such as 1 - 2 (y^2 + z^2) cancel: absolute tolerance relative to the row's own scale
    err = (got - want).abs() / want.abs().amax(dim=(1, 2), keepdim=True).clamp_min(1e-30)
    assert err.max() < 2e-6, float(err.max())
