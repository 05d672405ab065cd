"""Synthetic scenes of SURVEY.md §8d / BASELINE.md §3 (CPU-generated, seeded, float32).

Scene A "frustum" (configs c1, c2): camera at the origin, every centre projects inside the image.
Scene B "ball + orbit" (c3, c4, c5): points in the unit ball, 8 cameras on a radius-3 circle.
The generator pattern follows the reference's random_gaussians (splat_trainer/scene/io.py:136-147).
"""
from __future__ import annotations

import math
from typing import List, Tuple

import torch
import torch.nn.functional as F

from .data_types import CameraParams, Gaussians3D


def _sh_features(n: int, sh_degree: int, gen: torch.Generator) -> torch.Tensor:
  K = (sh_degree + 1) ** 2
  sh = torch.empty(n, 3, K)
  sh[:, :, 0] = 0.5 * torch.randn(n, 3, generator=gen)
  if K > 1:
    sh[:, :, 1:] = 0.1 * torch.randn(n, 3, K - 1, generator=gen)
  return sh


def scene_a(n: int, width: int, height: int, sh_degree: int = 0, seed: int = 0, sigma_px: float = 2.0
            ) -> Tuple[Gaussians3D, CameraParams]:
  """fov_x 60 deg, identity pose, z~U(2,10), projected sigma ~ sigma_px; feature = SH (N,3,K)."""
  gen = torch.Generator().manual_seed(seed)
  fx = fy = width / (2.0 * math.tan(math.radians(30.0)))
  cx, cy = width / 2.0, height / 2.0
  z = 2.0 + 8.0 * torch.rand(n, generator=gen)
  u = torch.rand(n, generator=gen)
  v = torch.rand(n, generator=gen)
  x = (u * width - cx) * z / fx
  y = (v * height - cy) * z / fy
  position = torch.stack([x, y, z], dim=1)
  log_scaling = torch.log(z * sigma_px / fx)[:, None] + 0.3 * torch.randn(n, 3, generator=gen)
  rotation = F.normalize(torch.randn(n, 4, generator=gen), dim=1)
  alpha_logit = 1.5 * torch.randn(n, 1, generator=gen)
  feature = _sh_features(n, sh_degree, gen)
  cam = CameraParams(T_camera_world=torch.eye(4), projection=torch.tensor([fx, fy, cx, cy]),
                     image_size=(width, height), near_plane=0.1, far_plane=100.0)
  return Gaussians3D(position.float(), rotation.float(), log_scaling.float(), alpha_logit.float(), feature.float()), cam


def look_at(eye: torch.Tensor, target: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
  """world->camera 4x4, OpenCV axes (+z forward, +x right, +y down)."""
  fwd = F.normalize(target - eye, dim=0)
  right = F.normalize(torch.linalg.cross(fwd, up), dim=0)
  down = torch.linalg.cross(fwd, right)
  R = torch.stack([right, down, fwd], dim=0)
  T = torch.eye(4)
  T[:3, :3] = R
  T[:3, 3] = -R @ eye
  return T


def scene_b(n: int, width: int, height: int, sh_degree: int = 3, seed: int = 1, num_cameras: int = 8,
            radius: float = 3.0, sigma_px: float = 1.5) -> Tuple[Gaussians3D, List[CameraParams]]:
  """Unit-ball points; cameras on a circle in the xz-plane looking at the origin, up = -y, fov_x 50 deg."""
  gen = torch.Generator().manual_seed(seed)
  d = F.normalize(torch.randn(n, 3, generator=gen), dim=1)
  r = torch.rand(n, generator=gen) ** (1.0 / 3.0)
  position = d * r[:, None]
  fx = fy = width / (2.0 * math.tan(math.radians(25.0)))
  cx, cy = width / 2.0, height / 2.0
  s_w = sigma_px * radius / fx
  log_scaling = math.log(s_w) + 0.3 * torch.randn(n, 3, generator=gen)
  rotation = F.normalize(torch.randn(n, 4, generator=gen), dim=1)
  alpha_logit = 1.5 * torch.randn(n, 1, generator=gen)
  feature = _sh_features(n, sh_degree, gen)
  cams = []
  for k in range(num_cameras):
    ang = 2.0 * math.pi * k / num_cameras
    eye = torch.tensor([radius * math.sin(ang), 0.0, -radius * math.cos(ang)])
    T = look_at(eye, torch.zeros(3), torch.tensor([0.0, -1.0, 0.0]))
    cams.append(CameraParams(T_camera_world=T, projection=torch.tensor([fx, fy, cx, cy]),
                             image_size=(width, height), near_plane=0.1, far_plane=100.0))
  return Gaussians3D(position.float(), rotation.float(), log_scaling.float(), alpha_logit.float(), feature.float()), cams
