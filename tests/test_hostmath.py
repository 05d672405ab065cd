"""CPU checks of the per-splat maths that the HIP kernels compile (csrc/gsr_math.h), built for the host by
g++ into libgsr_hostmath.so (a test shim, not a product path): K1 in-view test, K2 projection forward and
hand-derived backward, K3 SH basis, K4 tile-hit test -- each against the oracle."""
import ctypes as C

import numpy as np
import torch

from helpers import oracle, small_scene
from splat_trainer_amd import RasterConfig
from splat_trainer_amd._lib import raster_params


def _np(t):
  return np.ascontiguousarray(t.detach().cpu().numpy().astype(np.float32))


def _p(a):
  return a.ctypes.data_as(C.c_void_p)


def _lib(built_libs):
  return C.CDLL(built_libs[1])


def _scene(antialias=False, n=500):
  g, cam = small_scene(n, 96, 64, seed=21, sigma_px=3.0)
  cfg = RasterConfig(antialias=antialias, blur_cov=0.0 if antialias else 0.3)
  return g, cam, cfg


def test_project_forward_matches_oracle(built_libs):
  for aa in (False, True):
    g, cam, cfg = _scene(aa)
    lib = _lib(built_libs)
    M = g.position.shape[0]
    rp = raster_params(cfg)
    g2d = np.zeros((M, 6), np.float32); depth = np.zeros(M, np.float32); ss = np.zeros((M, 2), np.float32)
    T, proj = _np(cam.T_camera_world), _np(cam.projection)
    pos, ls, rot, al = _np(g.position), _np(g.log_scaling), _np(g.rotation), _np(g.alpha_logit)
    lib.hm_project_forward(_p(T), _p(proj), C.byref(rp), C.c_int64(M), _p(pos), _p(ls), _p(rot), _p(al), _p(g2d),
                           _p(depth), _p(ss))
    og, od, oss = oracle.project(g.position.double(), g.log_scaling.double(), g.rotation.double(),
                                 g.alpha_logit.double(), torch.arange(M), cam.T_camera_world.double(),
                                 cam.projection.double(), cfg)
    assert np.allclose(g2d, og.numpy(), rtol=2e-4, atol=1e-5)
    assert np.allclose(depth, od.numpy()[:, 0], rtol=1e-6)
    assert np.allclose(ss, oss.numpy(), rtol=2e-4, atol=1e-5)


def test_project_backward_matches_autograd(built_libs):
  for aa in (False, True):
    g, cam, cfg = _scene(aa, n=300)
    lib = _lib(built_libs)
    M = g.position.shape[0]
    rp = raster_params(cfg)
    torch.manual_seed(0)
    dg = torch.randn(M, 6, dtype=torch.float64)
    dd = torch.randn(M, 1, dtype=torch.float64)
    args = [t.double().clone().requires_grad_(True) for t in (g.position, g.log_scaling, g.rotation, g.alpha_logit)]
    og, od, _ = oracle.project(*args, torch.arange(M), cam.T_camera_world.double(), cam.projection.double(), cfg)
    ((og * dg).sum() + (od * dd).sum()).backward()
    T, proj = _np(cam.T_camera_world), _np(cam.projection)
    pos, ls, rot, al = _np(g.position), _np(g.log_scaling), _np(g.rotation), _np(g.alpha_logit)
    dpos = np.zeros((M, 3), np.float32); dls = np.zeros((M, 3), np.float32)
    drot = np.zeros((M, 4), np.float32); dal = np.zeros(M, np.float32)
    dgn, ddn = _np(dg), _np(dd[:, 0])
    lib.hm_project_backward(_p(T), _p(proj), C.byref(rp), C.c_int64(M), _p(pos), _p(ls), _p(rot), _p(al), _p(dgn),
                            _p(ddn), _p(dpos), _p(dls), _p(drot), _p(dal))
    for got, want, name in ((dpos, args[0].grad, "position"), (dls, args[1].grad, "log_scaling"),
                            (drot, args[2].grad, "rotation"), (dal, args[3].grad[:, 0], "alpha_logit")):
      want = want.numpy()
      scale = np.abs(want).max()
      assert np.abs(got - want).max() <= 2e-4 * scale, (name, aa, np.abs(got - want).max() / scale)


def test_in_view_matches_oracle(built_libs):
  lib = _lib(built_libs)
  torch.manual_seed(4)
  pos = torch.randn(5000, 3) * 4.0 + torch.tensor([0., 0., 3.])
  g, cam, cfg = _scene()
  W, H = cam.image_size
  mask = np.zeros(pos.shape[0], np.uint8)
  T, proj, p = _np(cam.T_camera_world), _np(cam.projection), _np(pos)
  lib.hm_in_view(_p(T), _p(proj), C.c_int64(pos.shape[0]), _p(p), C.c_int(W), C.c_int(H), C.c_float(cam.near_plane),
                 C.c_float(cam.far_plane), C.c_float(48.0), _p(mask))
  idx = oracle.frustum_cull(pos, cam.T_camera_world, cam.projection, cam.image_size, cam.near_plane, cam.far_plane, 48)
  want = np.zeros(pos.shape[0], np.uint8); want[idx.numpy()] = 1
  assert 0 < want.sum() < pos.shape[0]
  assert (mask != want).sum() <= 2      # fp32 boundary ties only


def test_sh_basis_matches_golden(built_libs, golden_dir):
  import os
  lib = _lib(built_libs)
  z = np.load(os.path.join(golden_dir, "rsh_deg0_4.npz"))
  dirs = z["dirs"].astype(np.float32)
  for deg in range(4):
    K = (deg + 1) ** 2
    Y = np.zeros((dirs.shape[0], K), np.float32)
    lib.hm_sh_basis(C.c_int(K), C.c_int64(dirs.shape[0]), _p(np.ascontiguousarray(dirs)), _p(Y))
    assert np.allclose(Y, z[f"deg{deg}"], atol=2e-6), deg


def test_tile_hits_are_conservative_and_tight(built_libs):
  """Every (splat, tile) pair in which some pixel passes the per-pixel test must be hit (conservative);
  and the exact ellipse/rectangle test must prune well below the bounding-box count (tight)."""
  lib = _lib(built_libs)
  g, cam = small_scene(300, 160, 128, seed=8, sigma_px=5.0)
  cfg = RasterConfig()
  M = g.position.shape[0]
  g2d, depth, _ = oracle.project(g.position, g.log_scaling, g.rotation, g.alpha_logit, torch.arange(M),
                                 cam.T_camera_world, cam.projection, cfg)
  W, H = cam.image_size
  tx, ty = (W + 15) // 16, (H + 15) // 16
  hits = np.zeros((M, ty, tx), np.uint8)
  rp = raster_params(cfg)
  gn = _np(g2d)
  lib.hm_tile_hits(C.byref(rp), C.c_int64(M), _p(gn), C.c_int(tx), C.c_int(ty), _p(hits))
  ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
  px, py = xs.reshape(-1).float() + .5, ys.reshape(-1).float() + .5
  dx = px[None] - g2d[:, 0:1]; dy = py[None] - g2d[:, 1:2]
  q = g2d[:, 2:3] * dx * dx + 2 * g2d[:, 3:4] * dx * dy + g2d[:, 4:5] * dy * dy
  alpha = torch.clamp(g2d[:, 5:6] * torch.exp(-0.5 * q), max=cfg.clamp_max_alpha)
  contrib = (q <= cfg.gaussian_scale ** 2) & (alpha >= cfg.alpha_threshold)           # (M, H*W)
  contrib = contrib.reshape(M, H, W)
  pad = torch.zeros(M, ty * 16, tx * 16, dtype=torch.bool); pad[:, :H, :W] = contrib
  need = pad.reshape(M, ty, 16, tx, 16).any(dim=4).any(dim=2).numpy()
  assert not (need & (hits == 0)).any()
  # bounding-box count of the 3-sigma ellipse for comparison
  _, _, counts, _, _ = oracle._tile_lists(g2d, depth, cam.image_size, cfg)
  assert hits.sum() < counts.sum().item()
  assert hits.sum() <= 1.35 * need.sum() + 50
