"""Small names the reference imports from ``taichi_splatting`` around the rasterizer path, so that the import swap of
INTEGRATION.md leaves no dangling import.  None of them does device work.

  TaichiQueue ........ splat_trainer/scene/mlp_scene.py:417 (``run_sync``), scripts/train_scan.py:237,
                       scripts/checkpoint.py:87, scripts/test_split.py:21 (``init``): the reference funnels every Taichi
                       launch through one thread.  The HIP library keeps no global state and enqueues on the caller's
                       current stream, so ``init`` is a no-op and ``run_sync`` calls straight through.
  count_nonfinite .... trainer/trainer.py:54,582 (``taichi_splatting.torch_lib.util``): {path: count} of non-finite
                       entries in a (nested) container of tensors; ``check_finite`` raises instead.
  random_camera, random_3d_gaussians .. scripts/test_split.py:1,24-25 (``taichi_splatting.tests.random_data``): a random
                       test camera and Gaussians inside its view.  The upstream generators are not in the reference
                       tree; these follow the call shape only (arguments used there: image_size; n, camera,
                       alpha_range, scale_factor).
"""
from __future__ import annotations

import math
from typing import Any, Callable, Dict, Optional, Tuple

import torch
import torch.nn.functional as F

from .data_types import CameraParams, Gaussians3D


class TaichiQueue:
  """No launch thread is needed (see module docstring); kept so that call sites stay unchanged."""

  @staticmethod
  def init(*_args, **_kwargs) -> None:
    return None

  @staticmethod
  def run_sync(fn: Callable, *args, **kwargs):
    return fn(*args, **kwargs)

  @staticmethod
  def stop() -> None:
    return None


def _walk(obj: Any, name: str):
  if isinstance(obj, torch.Tensor):
    yield name, obj
  elif isinstance(obj, dict):
    for k, v in obj.items():
      yield from _walk(v, f"{name}.{k}" if name else str(k))
  elif isinstance(obj, (list, tuple)):
    for i, v in enumerate(obj):
      yield from _walk(v, f"{name}[{i}]")
  elif hasattr(obj, "state_dict") and callable(obj.state_dict):
    yield from _walk(obj.state_dict(), name)
  elif hasattr(obj, "__dataclass_fields__"):
    for k in obj.__dataclass_fields__:
      yield from _walk(getattr(obj, k), f"{name}.{k}" if name else k)


def count_nonfinite(obj: Any, name: str = "") -> Dict[str, int]:
  """{path: number of NaN/Inf entries} for every floating tensor reachable from ``obj`` that has any."""
  out: Dict[str, int] = {}
  for path, t in _walk(obj, name):
    if t.is_floating_point() or t.is_complex():
      n = int((~torch.isfinite(t)).sum().item())
      if n > 0:
        out[path] = n
  return out


def check_finite(obj: Any, name: str = "") -> None:
  bad = count_nonfinite(obj, name)
  if bad:
    raise ValueError(f"non-finite entries: {bad}")


def random_camera(image_size: Tuple[int, int] = (640, 480), fov_deg: float = 60.0, distance: float = 4.0,
                  near_plane: float = 0.1, far_plane: float = 100.0, generator: Optional[torch.Generator] = None
                  ) -> CameraParams:
  """Pinhole camera at ``distance`` from the origin in a random direction, looking at the origin (OpenCV axes)."""
  from .synthetic import look_at
  W, H = image_size
  d = F.normalize(torch.randn(3, generator=generator), dim=0)
  eye = d * distance
  up = torch.tensor([0.0, -1.0, 0.0])
  if abs(float(d @ up)) > 0.95:
    up = torch.tensor([1.0, 0.0, 0.0])
  f = W / (2.0 * math.tan(math.radians(fov_deg) / 2.0))
  return CameraParams(T_camera_world=look_at(eye, torch.zeros(3), up), projection=torch.tensor([f, f, W / 2.0, H / 2.0]),
                      image_size=(int(W), int(H)), near_plane=near_plane, far_plane=far_plane)


def random_3d_gaussians(n: int, camera: CameraParams, alpha_range: Tuple[float, float] = (0.1, 0.9),
                        scale_factor: float = 1.0, depth_range: Tuple[float, float] = (2.0, 6.0),
                        generator: Optional[torch.Generator] = None) -> Gaussians3D:
  """``n`` Gaussians at random pixels of ``camera``'s image and random depths, RGB features (N,3), sizes about
  ``scale_factor`` x a tenth of the image at their depth, opacity uniform in ``alpha_range``."""
  W, H = camera.image_size
  fx, fy, cx, cy = (float(v) for v in camera.projection)
  rnd = lambda *s: torch.rand(*s, generator=generator)
  z = depth_range[0] + (depth_range[1] - depth_range[0]) * rnd(n)
  u, v = rnd(n) * W, rnd(n) * H
  cam_pts = torch.stack([(u - cx) * z / fx, (v - cy) * z / fy, z, torch.ones(n)], dim=1)
  world = (torch.linalg.inv(camera.T_camera_world.float().cpu()) @ cam_pts.T).T[:, :3]
  size = scale_factor * 0.1 * W * z / fx
  log_scaling = torch.log(size)[:, None] + 0.3 * torch.randn(n, 3, generator=generator)
  alpha = alpha_range[0] + (alpha_range[1] - alpha_range[0]) * rnd(n, 1)
  return Gaussians3D(position=world.contiguous(), rotation=F.normalize(torch.randn(n, 4, generator=generator), dim=1),
                     log_scaling=log_scaling, alpha_logit=torch.log(alpha / (1 - alpha)), feature=rnd(n, 3))
