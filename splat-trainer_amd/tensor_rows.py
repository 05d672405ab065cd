"""``TensorRows``: the slice of the ``tensordict.TensorDict`` interface the reference uses on this path.

tensordict is not importable here, and the boundary types must still answer the calls the reference makes on
them: ``points.tensors.select(...).replace(feature=...)`` and ``.to_dict()`` (scene/mlp_scene.py:296,395-396),
``points[split_idx].detach()`` handed to ``split_gaussians_uniform`` (mlp_scene.py:303), which then uses
``points['log_scaling']``, ``points.update(dict(...))`` (in place, returns self), ``points.apply(fn, batch_size=[n])``
and ``points.batch_size`` (gaussians/split.py:44-52,87-113).  One leading batch dimension (rows) only.
"""
from __future__ import annotations

from typing import Callable, Iterable, Mapping, Optional

import torch


class TensorRows(dict):
  """A ``dict`` of tensors that share their first dimension (the rows)."""

  def __init__(self, source: Optional[Mapping] = None, batch_size: Optional[Iterable[int]] = None, **tensors):
    super().__init__()
    for k, v in dict(source or {}, **tensors).items():
      super().__setitem__(k, v)
    rows = {int(v.shape[0]) for v in self.values()}
    if len(rows) > 1:
      raise ValueError(f"TensorRows: tensors disagree on the number of rows: {sorted(rows)}")
    n = rows.pop() if rows else None
    if batch_size is not None:
      want = tuple(int(b) for b in batch_size)
      if len(want) != 1 or (n is not None and want[0] != n):
        raise ValueError(f"TensorRows: batch_size {want} does not match {n} rows")
      n = want[0]
    self._rows = n if n is not None else 0

  # ----------------------------------------------------------------------------------------------- shape
  @property
  def batch_size(self) -> torch.Size:
    return torch.Size([self._rows])

  @property
  def device(self):
    for v in self.values():
      return v.device
    return None

  # -------------------------------------------------------------------------------------------- indexing
  def __getitem__(self, key):
    if isinstance(key, str):
      return super().__getitem__(key)
    return TensorRows({k: v[key] for k, v in self.items()})        # rows by mask / index tensor / slice

  def __setitem__(self, key, value):
    if not isinstance(key, str):
      raise TypeError("TensorRows: only whole columns can be assigned")
    if self and int(value.shape[0]) != self._rows:
      raise ValueError(f"TensorRows: column {key!r} has {value.shape[0]} rows, expected {self._rows}")
    if not self:
      self._rows = int(value.shape[0])
    super().__setitem__(key, value)

  # ---------------------------------------------------------------------------------- tensordict look-alikes
  def select(self, *keys: str) -> "TensorRows":
    return TensorRows({k: super(TensorRows, self).__getitem__(k) for k in keys}, batch_size=self.batch_size)

  def replace(self, *args, **tensors) -> "TensorRows":
    """New container with some columns swapped / added (the old one is left alone)."""
    out = dict(self)
    for a in args:
      out.update(a)
    out.update(tensors)
    return TensorRows(out)

  def update(self, other=(), **tensors) -> "TensorRows":             # in place, returns self (tensordict semantics)
    for k, v in dict(other, **tensors).items():
      self[k] = v
    return self

  def apply(self, fn: Callable[[torch.Tensor], torch.Tensor], batch_size: Optional[Iterable[int]] = None
            ) -> "TensorRows":
    return TensorRows({k: fn(v) for k, v in self.items()}, batch_size=batch_size)

  def to_dict(self) -> dict:
    return dict(self)

  def detach(self) -> "TensorRows":
    return self.apply(torch.detach, batch_size=self.batch_size)

  def clone(self) -> "TensorRows":
    return self.apply(torch.clone, batch_size=self.batch_size)

  def to(self, *args, **kwargs) -> "TensorRows":
    return self.apply(lambda t: t.to(*args, **kwargs), batch_size=self.batch_size)

  def __repr__(self):
    cols = ", ".join(f"{k}: {tuple(v.shape)}" for k, v in self.items())
    return f"TensorRows(batch_size={tuple(self.batch_size)}, {cols})"
