"""Boundary types of the rasterizer path.

These mirror, field for field, what splat-trainer exchanges with ``taichi_splatting`` at
the render boundary (reference call sites: ``splat_trainer/scene/mlp_scene.py:372-427``,
``splat_trainer/trainer/trainer.py:291-320``, ``splat_trainer/controller/point_state.py:34-50``,
``splat_trainer/scene/util.py:11-22``).  tensordict is not available here, so these are
plain dataclasses over torch tensors exposing the attributes/methods the reference uses.
"""
from __future__ import annotations

from dataclasses import dataclass, field, fields, replace as _dc_replace
from typing import Any, Optional, Tuple

import torch

from .tensor_rows import TensorRows


@dataclass(frozen=True)
class RasterConfig:
  """Raster options.  The reference sets ``antialias``, ``blur_cov``, ``compute_visibility``
  and ``compute_point_heuristic`` (trainer.py:305-310, mlp_scene.py:373); every keyword that
  names a field here is routed into the config by ``pop_raster_config`` (scene/util.py:11-22).
  """
  tile_size: int = 16                 # pixels per tile side (one wave64 owns a tile, 4 px per lane)
  margin_tiles: int = 3               # frustum-cull margin around the image, in tiles
  alpha_threshold: float = 1.0 / 255.0
  clamp_max_alpha: float = 0.99
  saturate_threshold: float = 0.9999  # stop compositing a pixel once T < 1 - saturate_threshold
  gaussian_scale: float = 3.0         # splat support: d^T conic d <= gaussian_scale^2
  blur_cov: float = 0.3               # added to the 2D covariance diagonal
  antialias: bool = False             # opacity *= sqrt(det(cov) / det(cov + aa_blur I))
  aa_blur: float = 0.3                # anti-alias filter variance (px^2) used when antialias=True
  compute_visibility: bool = False    # fill points.visibility (sum_pixels T*alpha) in forward
  compute_point_heuristic: bool = False  # fill prune_cost / split_score in backward
  segment_pairs: int = -1             # list segments of a tile: length in (tile, splat) pairs; -1 = chosen per frame,
                                      # 0 = no segmentation at all (rule: csrc/composite.hip segment_thresholds, gsr_segment_thresholds)
  segment_min_pairs: int = 0          # a tile is heavy (its FORWARD pass is segmented too) above this many pairs;
                                      # 0 = chosen per frame from the overlap count

  @property
  def transmittance_eps(self) -> float:
    return 1.0 - self.saturate_threshold


def pop_raster_config(options: dict) -> RasterConfig:
  """Same contract as splat_trainer/scene/util.py:11-22: keys naming RasterConfig fields
  are removed from ``options`` and used to build the config."""
  keys = {f.name for f in fields(RasterConfig)}
  raster_options = {k: v for k, v in options.items() if k in keys}
  for k in raster_options:
    del options[k]
  return RasterConfig(**raster_options)


@dataclass(frozen=True)
class CameraParams:
  """OpenCV-style pinhole (trainer.py:291-301, camera_table.py:167-175):
  +z forward, ``uv = xy / z * f + c`` in pixels, pixel centres at +0.5."""
  T_camera_world: torch.Tensor       # (4, 4)
  projection: torch.Tensor           # (4,)  fx, fy, cx, cy  (pixels)
  image_size: Tuple[int, int]        # (W, H)
  near_plane: float = 0.1
  far_plane: float = 100.0

  def to(self, device=None, dtype=None) -> "CameraParams":
    return _dc_replace(self,
                       T_camera_world=self.T_camera_world.to(device=device, dtype=dtype),
                       projection=self.projection.to(device=device, dtype=dtype))

  @property
  def device(self) -> torch.device:
    return self.T_camera_world.device

  @property
  def camera_position(self) -> torch.Tensor:
    cached = self.__dict__.get("_camera_position")       # the pose is immutable (frozen dataclass): compute once
    if cached is None:
      R = self.T_camera_world[:3, :3]
      t = self.T_camera_world[:3, 3]
      cached = -(R.t() @ t)
      object.__setattr__(self, "_camera_position", cached)
    return cached

  @property
  def focal_length(self) -> torch.Tensor:
    return self.projection[0:2]

  @property
  def principal_point(self) -> torch.Tensor:
    return self.projection[2:4]


class Gaussians3D:
  """mlp_scene.py:401-407.  rotation is a quaternion in **xyzw** order (scene/io.py:102-104).

  Accepts the calls the reference makes on the upstream tensorclass: keyword construction with an (ignored, checked)
  ``batch_size=`` (scene/io.py:106-115), ``Gaussians3D.from_dict(td, batch_dims=1)`` (mlp_scene.py:398),
  ``gaussians.apply(torch.detach)`` (scene/io.py:122), ``to_tensordict`` / ``from_tensordict``
  (scripts/test_split.py:33-34)."""
  FIELDS = ("position", "rotation", "log_scaling", "alpha_logit", "feature")
  __slots__ = FIELDS

  def __init__(self, position: torch.Tensor, rotation: torch.Tensor, log_scaling: torch.Tensor,
               alpha_logit: torch.Tensor, feature: torch.Tensor, batch_size=None):
    self.position = position        # (N, 3)
    self.rotation = rotation        # (N, 4) xyzw
    self.log_scaling = log_scaling  # (N, 3)
    self.alpha_logit = alpha_logit  # (N, 1)
    self.feature = feature          # (N, F) or (N, 3, K) SH coefficients
    if batch_size is not None and tuple(int(b) for b in batch_size) != (int(position.shape[0]),):
      raise ValueError(f"batch_size {tuple(batch_size)} does not match {position.shape[0]} points")

  @property
  def batch_size(self):
    return (self.position.shape[0],)

  @property
  def device(self):
    return self.position.device

  def apply(self, fn, batch_size=None) -> "Gaussians3D":
    """New instance with ``fn`` applied to every field (``gaussians.apply(torch.detach)``, scene/io.py:122)."""
    return Gaussians3D(**{name: fn(getattr(self, name)) for name in self.FIELDS}, batch_size=batch_size)

  def to(self, device=None, dtype=None) -> "Gaussians3D":
    return self.apply(lambda t: t.to(device=device, dtype=dtype))

  def detach(self) -> "Gaussians3D":
    return self.apply(torch.detach)

  def to_dict(self) -> dict:
    return {name: getattr(self, name) for name in self.FIELDS}

  def to_tensordict(self) -> TensorRows:
    """Row container of the five tensors (scripts/test_split.py:33, mlp_scene.py:73)."""
    return TensorRows(self.to_dict())

  @classmethod
  def from_dict(cls, d, batch_dims: int = 1) -> "Gaussians3D":
    if batch_dims != 1:
      raise ValueError("Gaussians3D has one batch dimension (the points)")
    return cls(**{name: d[name] for name in cls.FIELDS})

  from_tensordict = from_dict

  def requires_grad_(self, flag: bool = True) -> "Gaussians3D":
    for name in self.FIELDS:
      getattr(self, name).requires_grad_(flag)
    return self

  def __repr__(self):
    return "Gaussians3D(" + ", ".join(f"{n}={tuple(getattr(self, n).shape)}" for n in self.FIELDS) + ")"


def _index_rows(value: Any, rows: torch.Tensor) -> Any:
  if value is None:
    return None
  if isinstance(value, torch.Tensor):
    return value[rows]
  if hasattr(value, "__getitem__"):
    return value[rows]
  return value


def _detach(value: Any) -> Any:
  if isinstance(value, torch.Tensor):
    return value.detach()
  if hasattr(value, "detach"):
    return value.detach()
  return value            # incl. an unresolved visibility resolver (RenderedPoints): stays lazy


class RenderedPoints:
  """Per-point outputs for the M points that passed the frustum cull.

  ``visibility`` (compute_visibility) is readable as soon as the forward pass has been enqueued;
  ``prune_cost`` and ``split_score`` are owned by this object and filled **in place by backward**
  (compute_point_heuristic) -- the reference reads them after ``loss.backward()``
  (trainer.py:512-514).

  ``visibility`` may be handed in as a zero-argument resolver instead of a tensor: the renderer's backward pass
  delivers the per-splat sums as a by-product of its own reduction, so a frame that is back-propagated before anybody
  looks at the visibility never runs the separate forward-side reduction; a reader that comes first (``points.visible``
  in a regularizer, mlp_scene.py:268-288) triggers it on access.  Same sums in the same order either way: same bits.
  """
  FIELDS = ("idx", "depths", "opacity", "screen_scale", "visibility", "prune_cost", "split_score", "attributes")

  def __init__(self, idx: torch.Tensor, depths: torch.Tensor, opacity: torch.Tensor, screen_scale: torch.Tensor,
               visibility, prune_cost: torch.Tensor, split_score: torch.Tensor, attributes: Any = None):
    self.idx = idx                      # (M,) int64 indices into the scene's N points
    self.depths = depths                # (M, 1) camera-space z (differentiable)
    self.opacity = opacity              # (M,)  (differentiable)
    self.screen_scale = screen_scale    # (M, 2) sigma_major, sigma_minor in pixels
    self._visibility = visibility       # (M,) tensor, or a resolver returning it
    self.prune_cost = prune_cost        # (M,)
    self.split_score = split_score      # (M,)
    self.attributes = attributes        # user payload (mlp_scene.py:423)

  @property
  def visibility(self) -> torch.Tensor:
    v = self._visibility
    if v is not None and not isinstance(v, torch.Tensor):
      v = self._visibility = v()
    return v

  @visibility.setter
  def visibility(self, value):
    self._visibility = value

  @property
  def batch_size(self):
    return (self.idx.shape[0],)

  def _raw(self, name):
    return self._visibility if name == "visibility" else getattr(self, name)

  def to_dict(self) -> dict:
    return {n: getattr(self, n) for n in self.FIELDS}

  def replace(self, **kwargs) -> "RenderedPoints":
    unknown = set(kwargs) - set(self.FIELDS)
    if unknown:
      raise TypeError(f"RenderedPoints.replace: unknown field(s) {sorted(unknown)}")
    return RenderedPoints(**{n: kwargs[n] if n in kwargs else self._raw(n) for n in self.FIELDS})

  def __getitem__(self, rows) -> "RenderedPoints":
    return RenderedPoints(**{n: _index_rows(getattr(self, n), rows) for n in self.FIELDS})

  @property
  def visible_mask(self) -> torch.Tensor:
    return self.visibility > 0

  @property
  def visible(self) -> "RenderedPoints":
    return self[self.visible_mask.nonzero().squeeze(1)]

  @property
  def num_visible(self) -> int:
    return int(self.visible_mask.sum().item())

  def detach(self) -> "RenderedPoints":
    return RenderedPoints(**{n: _detach(self._raw(n)) for n in self.FIELDS})

  def __repr__(self):
    return f"RenderedPoints(M={self.idx.shape[0]})"


@dataclass
class Rendering:
  image: torch.Tensor                               # (H, W, C)
  camera: CameraParams
  points: RenderedPoints
  median_depth_image: Optional[torch.Tensor] = None  # (H, W) when render_median_depth=True
  final_transmittance: Optional[torch.Tensor] = None  # (H, W) T after the last composited splat
  num_overlaps: int = 0                              # O = sum of tile overlaps (reported per frame)

  @property
  def image_size(self) -> Tuple[int, int]:
    return self.camera.image_size

  @property
  def median_ndc_image(self) -> Optional[torch.Tensor]:
    if self.median_depth_image is None:
      return None
    n, f = self.camera.near_plane, self.camera.far_plane
    z = self.median_depth_image.clamp_min(n)
    return (f * (z - n)) / (z * (f - n))

  def detach(self) -> "Rendering":
    return _dc_replace(self, image=self.image.detach(), points=self.points.detach(),
                       median_depth_image=_detach(self.median_depth_image),
                       final_transmittance=_detach(self.final_transmittance))
