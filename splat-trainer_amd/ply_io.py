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


_GEOMETRY_COLUMNS = ("x", "y", "z", "opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3")


def _feature_columns(feature: torch.Tensor, with_sh: bool) -> List[str]:
  """Column names of the colour block: f_dc_* then channel-major f_rest_* for SH, f_* for plain features."""
  if with_sh:
    if feature.ndim != 3 or feature.shape[1] != 3:
      raise ValueError(f"with_sh=True needs (N, 3, K) SH coefficients, got {tuple(feature.shape)}")
    return [f"f_dc_{c}" for c in range(3)] + [f"f_rest_{i}" for i in range(3 * (feature.shape[2] - 1))]
  if feature.ndim != 2:
    raise ValueError(f"with_sh=False needs (N, F) features, got {tuple(feature.shape)}")
  return [f"f_{i}" for i in range(feature.shape[1])]


def to_vertex_array(gaussians: Gaussians3D, with_sh: bool = False) -> np.ndarray:
  """Gaussians -> structured float32 vertex array in the 3DGS column order (what io.py:13-67 produces through plyfile).
  The table is assembled as one dense (N, columns) float32 matrix -- geometry block, then the colour block with the
  DC coefficients first and the higher bands channel-major -- and reinterpreted as the record array."""
  cpu = lambda t: t.detach().to("cpu", torch.float32)
  feature = cpu(gaussians.feature)
  quat_wxyz = F.normalize(cpu(gaussians.rotation), dim=1)[:, [3, 0, 1, 2]]                 # memory xyzw -> disk wxyz
  if with_sh:
    colour = torch.cat([feature[:, :, 0], feature[:, :, 1:].flatten(1)], dim=1)
  else:
    colour = feature
  columns = list(_GEOMETRY_COLUMNS) + _feature_columns(feature, with_sh)
  table = torch.cat([cpu(gaussians.position), cpu(gaussians.alpha_logit).reshape(-1, 1), cpu(gaussians.log_scaling),
                     quat_wxyz, colour], dim=1).contiguous()
  assert table.shape[1] == len(columns)
  records = np.dtype([(name, "<f4") for name in columns])
  return table.numpy().view(records).reshape(-1)


def from_vertex_array(vertex: np.ndarray, with_sh: bool = False) -> Gaussians3D:
  """Structured vertex array (any column order, any float width) -> Gaussians; inverse of to_vertex_array
  (io.py:70-117): quaternion back to xyzw, SH rebuilt as (N, 3, K) from f_dc_* and channel-major f_rest_*."""
  present = set(vertex.dtype.names)

  def block(names) -> torch.Tensor:
    missing = [k for k in names if k not in present]
    if missing:
      raise KeyError(f"PLY vertex element lacks {missing}")
    return torch.from_numpy(np.stack([np.asarray(vertex[k], dtype=np.float32) for k in names], axis=1))

  n = vertex.shape[0]
  if with_sh:
    coeffs = sum(1 for k in present if k.startswith("f_dc_") or k.startswith("f_rest_"))
    per_channel = coeffs // 3
    degree_plus_1 = int(round(per_channel ** 0.5))
    if coeffs != 3 * per_channel or degree_plus_1 ** 2 != per_channel:
      raise ValueError(f"{coeffs} SH columns are not 3 x (degree + 1)^2")
    feature = block([f"f_dc_{c}" for c in range(3)]).reshape(n, 3, 1)
    if per_channel > 1:
      rest = block([f"f_rest_{i}" for i in range(3 * (per_channel - 1))]).reshape(n, 3, per_channel - 1)
      feature = torch.cat([feature, rest], dim=2)
  else:
    width = sum(1 for k in present if k.startswith("f_") and k[2:].isdigit())
    feature = block([f"f_{i}" for i in range(width)])
  quat_xyzw = F.normalize(block([f"rot_{i}" for i in range(4)]), dim=1)[:, [1, 2, 3, 0]]   # disk wxyz -> memory xyzw
  return Gaussians3D(position=block(["x", "y", "z"]), rotation=quat_xyzw,
                     log_scaling=block([f"scale_{i}" for i in range(3)]), alpha_logit=block(["opacity"]),
                     feature=feature, batch_size=(n,))


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
