"""K1 cull, K2 projection fwd/bwd and K3 SH fwd/bwd on the GPU against the CPU oracle.
Tolerance: 1e-4 relative to the tensor's max magnitude (BASELINE.json north_star), fp32 compute."""
import pytest
import torch

import splat_trainer_amd as sta
from helpers import oracle, rel_err, small_scene
from splat_trainer_amd import synthetic

pytestmark = pytest.mark.gpu
TOL = 1e-4


def test_frustum_cull_matches_oracle_and_is_ascending():
  g, cams = synthetic.scene_b(200_000, 640, 360, sh_degree=0, seed=2, radius=1.2)   # ~half culled
  cam = cams[3]
  cfg = sta.RasterConfig()
  idx = sta.frustum_cull(g.position.cuda(), cam.to("cuda"), cfg).cpu()
  want = oracle.frustum_cull(g.position, cam.T_camera_world, cam.projection, cam.image_size, cam.near_plane,
                             cam.far_plane, cfg.margin_tiles * cfg.tile_size)
  assert idx.dtype == torch.int64
  assert 0.1 * 200_000 < want.numel() < 0.9 * 200_000
  assert (idx[1:] > idx[:-1]).all()
  a, b = set(idx.tolist()), set(want.tolist())
  assert len(a ^ b) <= 4                      # fp32 ties exactly on a frustum plane
  # sizes around the wave span / block edges, and the empty input
  for n in (0, 1, 63, 1023, 1024, 1025, 4097):
    p = g.position[:n]
    i = sta.frustum_cull(p.cuda(), cam.to("cuda"), cfg).cpu()
    w = oracle.frustum_cull(p, cam.T_camera_world, cam.projection, cam.image_size, cam.near_plane, cam.far_plane, 48)
    assert len(set(i.tolist()) ^ set(w.tolist())) <= 1, n


@pytest.mark.parametrize("antialias", [False, True])
def test_project_forward_backward_match_oracle(antialias):
  g, cams = synthetic.scene_b(20_000, 320, 240, sh_degree=0, seed=4)
  cam = cams[1]
  cfg = sta.RasterConfig(antialias=antialias, blur_cov=0.0 if antialias else 0.3)
  gd = sta.Gaussians3D(*(t.clone().cuda().requires_grad_(True) for t in
                         (g.position, g.rotation, g.log_scaling, g.alpha_logit, g.feature)))
  g2d, depth, idx = sta.project_to_image(gd, cam.to("cuda"), cfg)
  args = [t.clone().double().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit)]
  og, od, _ = oracle.project(*args, idx.cpu(), cam.T_camera_world.double(), cam.projection.double(), cfg)
  assert g2d.shape == (idx.numel(), 6) and depth.shape == (idx.numel(), 1)
  for c, name in enumerate(["u", "v", "A", "B", "C", "opacity"]):
    assert rel_err(g2d[:, c], og[:, c]) < TOL, name
  assert rel_err(depth, od) < 1e-6
  torch.manual_seed(1)
  dg = torch.randn(idx.numel(), 6, dtype=torch.float64)
  dd = torch.randn(idx.numel(), 1, dtype=torch.float64)
  ((og * dg).sum() + (od * dd).sum()).backward()
  ((g2d * dg.float().cuda()).sum() + (depth * dd.float().cuda()).sum()).backward()
  assert rel_err(gd.position.grad, args[0].grad) < TOL
  assert rel_err(gd.log_scaling.grad, args[1].grad) < TOL
  assert rel_err(gd.rotation.grad, args[2].grad) < TOL
  assert rel_err(gd.alpha_logit.grad, args[3].grad) < TOL
  # rows outside the cull receive exactly zero
  mask = torch.ones(g.position.shape[0], dtype=torch.bool); mask[idx.cpu()] = False
  assert gd.position.grad.cpu()[mask].abs().max().item() == 0 if mask.any() else True


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_evaluate_sh_at_matches_oracle(deg):
  torch.manual_seed(deg)
  N, K = 30_000, (deg + 1) ** 2
  sh = torch.randn(N, 3, K)
  pos = torch.randn(N, 3) * 2
  cam_pos = torch.tensor([0.3, -0.2, 4.0])
  idx = torch.randperm(N)[: N // 2].sort().values
  shd = sh.clone().cuda().requires_grad_(True)
  posd = pos.clone().cuda().requires_grad_(True)
  got = sta.evaluate_sh_at(shd, posd, idx.cuda(), cam_pos.cuda())
  sho = sh.clone().double().requires_grad_(True)
  poso = pos.clone().double().requires_grad_(True)
  want = oracle.evaluate_sh_at(sho, poso, idx, cam_pos.double())
  assert got.shape == (idx.numel(), 3)
  assert rel_err(got, want) < TOL
  w = torch.randn(idx.numel(), 3)
  (got * w.cuda()).sum().backward()
  (want * w.double()).sum().backward()
  assert rel_err(shd.grad, sho.grad) < TOL
  assert shd.grad.shape == (N, 3, K)
  if deg > 0:
    assert rel_err(posd.grad, poso.grad) < TOL      # gradient through the view direction
  else:
    assert posd.grad.abs().max().item() == 0


def test_empty_inputs():
  cfg = sta.RasterConfig()
  g, cam = small_scene(16, 32, 32)
  empty = sta.Gaussians3D(*(t[:0].cuda() for t in (g.position, g.rotation, g.log_scaling, g.alpha_logit, g.feature)))
  g2d, depth, idx = sta.project_to_image(empty, cam.to("cuda"), cfg)
  assert g2d.shape == (0, 6) and depth.shape == (0, 1) and idx.shape == (0,)
  out = sta.evaluate_sh_at(torch.zeros(5, 3, 4).cuda(), torch.zeros(5, 3).cuda(),
                           torch.zeros(0, dtype=torch.int64).cuda(), torch.zeros(3).cuda())
  assert out.shape == (0, 3)


@pytest.mark.parametrize("K,frac", [(16, 0.6), (9, 0.3), (4, 1.0), (1, 0.6), (16, 0.05)])
def test_sh_backward_overwrite_paths_match_accumulate(K, frac):
  """GradOut.feature_uninitialized / plain autograd: d_sh is overwritten row for row (gsr_inverse_map +
  gsr_sh_backward_dense; zero-fill + accumulate when fewer than 1/8 of the rows are visible) -- same values as adding
  into a zero-filled buffer, zeros in the rows the camera did not see, stale buffer contents gone."""
  torch.manual_seed(K)
  n = 4099
  sh = torch.randn(n, 3, K, device="cuda")
  pos = torch.randn(n, 3, device="cuda") * 2
  cam = torch.tensor([0.3, -0.2, 5.0], device="cuda")
  idx = (torch.rand(n, device="cuda") < frac).nonzero().squeeze(1) if frac < 1.0 else torch.arange(n, device="cuda")
  g = torch.randn(idx.shape[0], 3, device="cuda")
  # reference: accumulate into zero-filled caller buffers
  d_sh_a, d_pos_a = torch.zeros_like(sh), torch.zeros_like(pos)
  out = sta.evaluate_sh_at(sh.requires_grad_(True), pos.requires_grad_(True), idx, cam, grad_out=(d_sh_a, d_pos_a))
  out.backward(g)
  # overwrite into a buffer full of stale values
  owner = sta.GradOut(feature_uninitialized=True)
  d_sh_b, d_pos_b = torch.full_like(sh, 7.0), torch.zeros_like(pos)
  out = sta.evaluate_sh_at(sh, pos, idx, cam, grad_out=(d_sh_b, d_pos_b, owner))
  out.backward(g)
  assert owner.feature_uninitialized is False
  assert rel_err(d_sh_b, d_sh_a) < 1e-6 and rel_err(d_pos_b, d_pos_a) < 1e-6
  unseen = torch.ones(n, dtype=torch.bool, device="cuda")
  unseen[idx] = False
  assert float(d_sh_b[unseen].abs().max()) == 0.0 if unseen.any() else True
  # a second camera then accumulates on top
  out = sta.evaluate_sh_at(sh, pos, idx, cam, grad_out=(d_sh_b, d_pos_b, owner))
  out.backward(g)
  assert rel_err(d_sh_b, 2 * d_sh_a) < 1e-6
  # plain autograd takes the same overwrite path
  sh2, pos2 = sh.detach().clone().requires_grad_(True), pos.detach().clone().requires_grad_(True)
  sta.evaluate_sh_at(sh2, pos2, idx, cam).backward(g)
  assert rel_err(sh2.grad, d_sh_a) < 1e-6 and rel_err(pos2.grad, d_pos_a) < 1e-6


def test_inverse_map_edges():
  import ctypes as C
  from splat_trainer_amd import _lib
  lib = _lib.load()
  st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
  for n, rows in [(10, [0, 1, 2, 9]), (10, [3]), (7, []), (5, [0, 1, 2, 3, 4]), (4000, list(range(17, 4000, 13)))]:
    idx = torch.tensor(rows, dtype=torch.int64, device="cuda")
    inv = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
    _lib.check(lib.gsr_inverse_map(C.c_void_p(idx.data_ptr()) if rows else None, len(rows), n, C.c_void_p(inv.data_ptr()), st),
               "gsr_inverse_map")
    want = torch.full((n,), -1, dtype=torch.int32)
    want[torch.tensor(rows, dtype=torch.int64)] = torch.arange(len(rows), dtype=torch.int32)
    assert torch.equal(inv.cpu(), want), (n, rows[:5])
