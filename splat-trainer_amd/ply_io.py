"""3DGS-compatible PLY reader / writer for Gaussians3D -- the on-disk format either side of the rasterizer
path.  Field layout and conventions follow the reference exactly
(splat_trainer/scene/io.py:13-132):

  x y z | opacity (= alpha_logit) | scale_0..2 (= log_scaling) | rot_0..3 (= normalised quaternion, **wxyz**:
  the in-memory xyzw is rolled by one, io.py:45,102-104) | with_sh: f_dc_0..2 then f_rest_i, CHANNEL-major
  (feature[:, :, 1:] reshaped to (N, 3*(K-1)), io.py:50-61,93-97) | else f_0..f_{F-1}.

The reference goes through the ``plyfile`` package, which is not available here; the binary little-endian PLY
container is written/parsed directly with numpy (header: ``format binary_little_endian 1.0``, one ``vertex``
element of float32 properties).  Pure host-side I/O, no GPU involved.
"""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

from .data_types import Gaussians3D


def _field_names(feature: torch.Tensor, with_sh: bool) -> List[str]:
  names = ["x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
  if with_sh:
    assert feature.ndim == 3, f"Expected ndim=3 sh_feature tensor, got {tuple(feature.shape)}"
    num_sh = feature.shape[2] * feature.shape[1]
    names += ["f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(num_sh - 3)]
  else:
    assert feature.ndim == 2, f"Expected ndim=2 feature tensor, got {tuple(feature.shape)}"
    names += [f"f_{i}" for i in range(feature.shape[1])]
  return names


def to_vertex_array(gaussians: Gaussians3D, with_sh: bool = False) -> np.ndarray:
  """Structured float32 array with the reference's field names and order (io.py:13-67)."""
  g = gaussians
  pos, ls = g.position.detach().cpu(), g.log_scaling.detach().cpu()
  feat, al = g.feature.detach().cpu(), g.alpha_logit.detach().cpu()
  names = _field_names(feat, with_sh)
  vertex = np.zeros(pos.shape[0], dtype=[(n, "<f4") for n in names])
  for i, n in enumerate(["x", "y", "z"]):
    vertex[n] = pos[:, i].numpy()
  for i in range(3):
    vertex[f"scale_{i}"] = ls[:, i].numpy()
  rotation = torch.roll(F.normalize(g.rotation.detach().cpu(), dim=1), 1, dims=(1,))      # xyzw -> wxyz
  for i in range(4):
    vertex[f"rot_{i}"] = rotation[:, i].numpy()
  vertex["opacity"] = al[:, 0].numpy()
  if with_sh:
    sh_dc, sh_rest = feat[:, :, 0], feat[:, :, 1:]
    sh_rest = sh_rest.reshape(sh_rest.shape[0], sh_rest.shape[1] * sh_rest.shape[2])
    for i in range(3):
      vertex[f"f_dc_{i}"] = sh_dc[:, i].numpy()
    for i in range(sh_rest.shape[1]):
      vertex[f"f_rest_{i}"] = sh_rest[:, i].numpy()
  else:
    for i in range(feat.shape[1]):
      vertex[f"f_{i}"] = feat[:, i].numpy()
  return vertex


def from_vertex_array(vertex: np.ndarray, with_sh: bool = False) -> Gaussians3D:
  """io.py:70-117."""
  def get_keys(ks):
    return torch.stack([torch.from_numpy(np.ascontiguousarray(vertex[k]).astype(np.float32)) for k in ks], dim=-1)

  n = vertex.shape[0]
  positions = get_keys(["x", "y", "z"])
  attrs = sorted(vertex.dtype.names)
  log_scaling = get_keys([f"scale_{k}" for k in range(3)])
  if with_sh:
    sh_attrs = [k for k in attrs if k.startswith("f_rest_") or k.startswith("f_dc_")]
    n_sh = len(sh_attrs) // 3
    deg = int(np.sqrt(n_sh))
    assert deg * deg == n_sh, f"SH feature count must be square ({deg} * {deg} != {n_sh}), got {len(sh_attrs)}"
    sh_dc = get_keys([f"f_dc_{k}" for k in range(3)]).view(n, 3, 1)
    if n_sh > 1:
      sh_rest = get_keys([f"f_rest_{k}" for k in range(3 * (n_sh - 1))]).view(n, 3, n_sh - 1)
      features = torch.cat([sh_dc, sh_rest], dim=2)
    else:
      features = sh_dc
  else:
    feature_attrs = [k for k in attrs if k.startswith("f_")]
    features = get_keys([f"f_{k}" for k in range(len(feature_attrs))])
  rotation = get_keys([f"rot_{k}" for k in range(4)])
  rotation = torch.roll(F.normalize(rotation, dim=1), -1, dims=(1,))                       # wxyz -> xyzw
  alpha_logit = get_keys(["opacity"])
  return Gaussians3D(position=positions, rotation=rotation, log_scaling=log_scaling, alpha_logit=alpha_logit,
                     feature=features)


_PLY_TYPES = {"float": "<f4", "float32": "<f4", "double": "<f8", "float64": "<f8", "uchar": "u1", "uint8": "u1",
              "char": "i1", "int8": "i1", "short": "<i2", "int16": "<i2", "ushort": "<u2", "uint16": "<u2",
              "int": "<i4", "int32": "<i4", "uint": "<u4", "uint32": "<u4"}


def write_ply(filename: Union[str, Path], vertex: np.ndarray) -> None:
  header = ["ply", "format binary_little_endian 1.0", f"element vertex {vertex.shape[0]}"]
  header += [f"property float {n}" for n in vertex.dtype.names]
  header += ["end_header"]
  with open(str(filename), "wb") as f:
    f.write(("\n".join(header) + "\n").encode("ascii"))
    f.write(np.ascontiguousarray(vertex).tobytes())


def read_ply(filename: Union[str, Path]) -> np.ndarray:
  """Vertex element of a binary little-endian (or ascii) PLY as a structured array."""
  with open(str(filename), "rb") as f:
    data = f.read()
  end = data.index(b"end_header\n") + len(b"end_header\n")
  lines = data[:end].decode("ascii").split("\n")
  if lines[0].strip() != "ply":
    raise ValueError("not a PLY file")
  fmt, count, props, in_vertex, others = None, 0, [], False, False
  for ln in lines[1:]:
    tok = ln.split()
    if not tok:
      continue
    if tok[0] == "format":
      fmt = tok[1]
    elif tok[0] == "element":
      in_vertex = tok[1] == "vertex"
      if in_vertex:
        count = int(tok[2])
      elif count:
        others = True
    elif tok[0] == "property" and in_vertex:
      if tok[1] == "list":
        raise ValueError("list properties are not supported in the vertex element")
      props.append((tok[2], _PLY_TYPES[tok[1]]))
  dtype = np.dtype(props)
  if fmt == "binary_little_endian":
    return np.frombuffer(data, dtype=dtype, count=count, offset=end).copy()
  if fmt == "ascii":
    rows = np.loadtxt(data[end:].decode("ascii").split("\n")[:count], dtype=np.float64, ndmin=2)
    out = np.zeros(count, dtype=dtype)
    for i, (n, _) in enumerate(props):
      out[n] = rows[:, i]
    return out
  raise ValueError(f"unsupported PLY format {fmt}")


def write_gaussians(filename: Union[str, Path], gaussians: Gaussians3D, with_sh: bool = True) -> None:
  """io.py:119-123."""
  write_ply(filename, to_vertex_array(gaussians, with_sh=with_sh))


def read_gaussians(filename: Union[str, Path], with_sh: bool = True) -> Gaussians3D:
  """io.py:127-132."""
  return from_vertex_array(read_ply(filename), with_sh=with_sh)


def random_gaussians(n: int, sh_degree: int, generator: torch.Generator = None) -> Gaussians3D:
  """io.py:136-147 (used by the reference's own round-trip test)."""
  r = lambda *s: torch.randn(*s, generator=generator)
  return Gaussians3D(position=r(n, 3), rotation=F.normalize(r(n, 4), dim=1), alpha_logit=r(n, 1),
                     log_scaling=r(n, 3) * 4, feature=r(n, 3, (sh_degree + 1) ** 2))
