"""Densify / prune kernels (csrc/densify.hip) against torch on the same device: the top-n mask must equal a stable
argsort's choice bit for bit (massive ties, inf, NaN, -0, n = 0 / N), the fused compaction must equal
``cat([column[mask], tail])`` bit for bit for every column, and the densify round built from them must leave the
same ParameterClass as the reference's two-step form (mlp_scene.py:306-310)."""
import time

import pytest
import torch

from splat_trainer_amd import densify
from splat_trainer_amd.controller_math import PointState, find_split_prune_indexes, take_n
from splat_trainer_amd.optim import ParameterClass, VisibilityAwareLaProp

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _reference_mask(t, n, descending):
  idx = torch.argsort(t, descending=descending, stable=True)[:n]
  mask = torch.zeros_like(t, dtype=torch.bool)
  mask[idx] = True
  return mask


@pytest.mark.parametrize("n_points", [1, 7, 255, 256, 257, 10_000, 300_001])
@pytest.mark.parametrize("descending", [False, True])
def test_select_n_matches_stable_argsort(n_points, descending):
  gen = torch.Generator(device=DEV).manual_seed(n_points)
  t = torch.randn(n_points, device=DEV, generator=gen).exp()
  for n in sorted({0, 1, n_points // 3, n_points - 1, n_points, n_points + 5}):
    got = densify.select_n(t, n, descending)
    assert got.dtype == torch.bool and int(got.sum()) == min(n, n_points)
    assert torch.equal(got, _reference_mask(t, min(n, n_points), descending)), (n_points, n, descending)


def test_select_n_massive_ties_inf_nan_and_signed_zero():
  gen = torch.Generator(device=DEV).manual_seed(0)
  N = 200_000
  t = torch.rand(N, device=DEV, generator=gen)
  t[torch.rand(N, device=DEV, generator=gen) < 0.4] = 0.0                # unseen points: prune_cost / split_score = 0
  t[torch.rand(N, device=DEV, generator=gen) < 0.3] = float("inf")       # masked_heuristics: not seen often enough
  t[5] = -0.0
  t[17] = float("nan")
  t[123] = -1.5
  for descending in (False, True):
    for n in (1, 10, 50_000, 90_000, 150_000, N - 3):
      got = densify.select_n(t, n, descending)
      assert torch.equal(got, _reference_mask(t, n, descending)), (descending, n)
  # all equal: the first n indexes
  z = torch.zeros(1000, device=DEV)
  assert torch.equal(densify.select_n(z, 300).nonzero().squeeze(1), torch.arange(300, device=DEV))


def test_compact_rows_matches_torch_for_every_column():
  gen = torch.Generator(device=DEV).manual_seed(1)
  for N, n_tail in ((1000, 64), (70_001, 3000), (257, 0), (5, 2)):
    keep = torch.rand(N, device=DEV, generator=gen) < 0.7
    shapes = [(3,), (4,), (1,), (3, 16), (), (48,)]
    cols = [torch.randn((N,) + s, device=DEV, generator=gen) for s in shapes]
    cols.append(torch.randint(-5, 5, (N,), device=DEV, dtype=torch.int32))
    tails = [torch.randn((n_tail,) + s, device=DEV, generator=gen) if i % 2 == 0 else None for i, s in enumerate(shapes)]
    tails.append(None)
    outs = densify.compact_rows(keep, list(zip(cols, tails)), n_tail=n_tail)
    for c, t, o in zip(cols, tails, outs):
      want = torch.cat([c[keep], t if t is not None else c.new_zeros((n_tail,) + tuple(c.shape[1:]))])
      assert o.dtype == c.dtype and torch.equal(o, want), (N, tuple(c.shape))
  none = densify.compact_rows(torch.zeros(300, dtype=torch.bool, device=DEV), [(torch.randn(300, 3, device=DEV), None)])
  assert none[0].shape == (0, 3)


def _points(n, k=16):
  gen = torch.Generator(device=DEV).manual_seed(3)
  tensors = dict(position=torch.randn(n, 3, device=DEV, generator=gen), log_scaling=torch.randn(n, 3, device=DEV, generator=gen),
                 rotation=torch.randn(n, 4, device=DEV, generator=gen), alpha_logit=torch.randn(n, 1, device=DEV, generator=gen),
                 feature=torch.randn(n, 3, k, device=DEV, generator=gen), visible=torch.rand(n, device=DEV, generator=gen))
  groups = dict(position=dict(lr=1e-3, type="local_vector"), log_scaling=dict(lr=1e-3), rotation=dict(lr=1e-3, type="vector"),
                alpha_logit=dict(lr=1e-3), feature=dict(lr=1e-3))
  pc = ParameterClass(tensors, groups, optimizer=VisibilityAwareLaProp, betas=(0.8, 0.95))
  for name, g in pc._state["groups"].items():                      # non-trivial optimizer state
    for v in g.values():
      v.copy_(torch.randn(v.shape, device=DEV, generator=gen))
  pc._state["step"].copy_(torch.randint(0, 50, (n,), device=DEV, generator=gen).float())
  pc._state["vis_avg"].copy_(torch.rand(n, device=DEV, generator=gen))
  return pc


def test_keep_and_append_equals_mask_then_append_and_times_3m():
  for n, label in ((20_000, "20k"), (3_000_000, "3M")):
    pc = _points(n)
    gen = torch.Generator(device=DEV).manual_seed(4)
    state = PointState.new_zeros(n, DEV)
    state.prune_cost.copy_(torch.rand(n, device=DEV, generator=gen))
    state.split_score.copy_(torch.rand(n, device=DEV, generator=gen))
    state.points_in_view.copy_(torch.randint(0, 12, (n,), device=DEV, generator=gen).to(torch.int16))
    state.max_scale_px.copy_(300 * torch.rand(n, device=DEV, generator=gen) ** 8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    split_mask, prune_mask = find_split_prune_indexes(state, 0.25, int(1.1 * n))     # device radix select inside
    torch.cuda.synchronize()
    t_select = time.perf_counter() - t0
    cpu_state = PointState(*(getattr(state, f).cpu() for f in ("prune_cost", "split_score", "max_scale_px", "points_in_view", "visibility")))
    s_cpu, p_cpu = find_split_prune_indexes(cpu_state, 0.25, int(1.1 * n))             # stable argsort on the host
    assert torch.equal(split_mask.cpu(), s_cpu) and torch.equal(prune_mask.cpu(), p_cpu)
    keep = ~(split_mask | prune_mask)
    n_new = 2 * int(split_mask.sum())
    children = {k: torch.randn((n_new,) + tuple(v.shape[1:]), device=DEV, generator=gen) for k, v in pc.tensors.items()}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fused = pc.keep_and_append(keep, children)
    torch.cuda.synchronize()
    t_fused = time.perf_counter() - t0
    t0 = time.perf_counter()
    two_step = pc[keep].append_tensors(children)                                      # the reference's form
    torch.cuda.synchronize()
    t_torch = time.perf_counter() - t0
    for k in pc.keys():
      assert torch.equal(fused.tensors[k], two_step.tensors[k]), k
    for g in fused._state["groups"]:
      for name in fused._state["groups"][g]:
        assert torch.equal(fused._state["groups"][g][name], two_step._state["groups"][g][name]), (g, name)
    assert torch.equal(fused._state["step"], two_step._state["step"])
    assert torch.equal(fused._state["vis_avg"], two_step._state["vis_avg"])
    def timed(fn, reps=5):
      fn()
      torch.cuda.synchronize()
      t = time.perf_counter()
      for _ in range(reps):
        fn()
      torch.cuda.synchronize()
      return (time.perf_counter() - t) / reps
    pc_, _ = state.masked_heuristics(5)
    t_sel = timed(lambda: densify.select_n(pc_, n // 40))
    t_argsort = timed(lambda: _reference_mask(pc_, n // 40, False))
    print(f"densify round at {label}: find_split_prune_indexes {t_select * 1e3:.2f} ms (first call); one take_n: radix "
          f"select {t_sel * 1e3:.3f} ms vs torch stable argsort + mask {t_argsort * 1e3:.3f} ms; keep+append of "
          f"{len(pc.keys()) + 2 + 2 * len(pc._state['groups'])} columns fused {t_fused * 1e3:.2f} ms vs mask-then-append "
          f"{t_torch * 1e3:.2f} ms; {n} -> {fused.num_points} points")


def test_point_state_add_matches_the_reference_arithmetic():
  """gsr_point_state_add vs PointState.add_rendering's torch ops (point_state.py:34-50, run on the CPU) over a few
  cameras: max / count / sum exact, the two exp_lerp EMAs to 2e-6 (expf / logf of the device vs the host's)."""
  import splat_trainer_amd as sta
  gen = torch.Generator().manual_seed(9)
  n = 50_000
  dev_state, cpu_state = PointState.new_zeros(n, DEV), PointState.new_zeros(n, "cpu")
  for cam in range(4):
    m = 30_000 + 1000 * cam
    idx = torch.randperm(n, generator=gen)[:m].sort().values
    vis = torch.rand(m, generator=gen)
    vis[torch.rand(m, generator=gen) < 0.3] = 0.0
    pts = sta.RenderedPoints(idx=idx, depths=torch.zeros(m, 1), opacity=torch.zeros(m),
                             screen_scale=50 * torch.rand(m, 2, generator=gen), visibility=vis,
                             prune_cost=torch.rand(m, generator=gen) * 10 ** (4 * torch.rand(m, generator=gen) - 2) * (vis > 0),
                             split_score=torch.rand(m, generator=gen) * (vis > 0))
    cpu_state.add_rendering(sta.Rendering(image=None, camera=None, points=pts))
    dev_pts = sta.RenderedPoints(**{k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in pts.to_dict().items()})
    dev_state.add_rendering(sta.Rendering(image=None, camera=None, points=dev_pts))
  assert torch.equal(dev_state.max_scale_px.cpu(), cpu_state.max_scale_px)
  assert torch.equal(dev_state.points_in_view.cpu(), cpu_state.points_in_view)
  assert torch.allclose(dev_state.visibility.cpu(), cpu_state.visibility, rtol=1e-6, atol=0)
  for f in ("split_score", "prune_cost"):
    a, b = getattr(dev_state, f).cpu(), getattr(cpu_state, f)
    assert torch.allclose(a, b, rtol=2e-6, atol=2e-6), (f, (a - b).abs().max().item())    # observed 4e-7 absolute
