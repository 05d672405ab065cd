"""Data-parallel seam of the rasterizer path: shard the cameras of a batch over ranks (one process per
GPU), sum the Gaussian parameter gradients with ONE fused collective over RCCL/xGMI, and replay the
per-camera point statistics in camera order so densification stays bit-identical to a 1-GPU run.

The reference has no distributed code; the seam is its sequential per-camera loop with in-place grad
accumulation, splat_trainer/trainer/trainer.py:500-514 (``evaluate_backward_with``).  Parameters (and
optimizer state) are fully replicated; a camera's render is not split across GPUs.

Collective choice.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so what matters is that the
message is ONE large contiguous buffer that RCCL can split over all links/channels at once, not many
per-tensor calls each bound by launch latency: all gradients are packed into one contiguous fp32 buffer
(padded to a multiple of world_size; 120 MB at 500k splats/SH3, 708 MB at 3M).  ``mode="all_reduce"``
(default of bench.py) issues a single fused all-reduce; ``mode="reduce_scatter"`` issues
``reduce_scatter_tensor`` + ``all_gather_into_tensor`` on the same buffer (the form a sharded optimizer
step would slot between; gloo, used by the CPU tests, falls back to all_reduce).  Neither could be timed in
this build environment (one GPU); both are kept selectable.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_cameras(num_cameras: int, rank: int, world_size: int) -> List[int]:
  """Camera j of the batch goes to rank j mod world_size."""
  return [j for j in range(num_cameras) if j % world_size == rank]


class GradBucket:
  """One contiguous fp32 buffer viewing the ``.grad`` of every parameter (so the collective runs on a single
  large message and no pack/unpack copies are needed after the first step)."""

  def __init__(self, params: Sequence[torch.Tensor], world_size: int, extra: int = 0):
    self.params = list(params)
    self.world_size = max(world_size, 1)
    sizes = [p.numel() for p in self.params]
    align = 64                                        # floats: every view starts on a 256-byte boundary (the
    starts, off = [], 0                               # kernels use 16-byte vector accesses on gradient rows)
    for n in sizes + [extra]:
      starts.append(off)
      off += ((n + align - 1) // align) * align
    total = off
    self.padded = ((total + self.world_size - 1) // self.world_size) * self.world_size
    dev = self.params[0].device
    self.flat = torch.zeros(self.padded, dtype=torch.float32, device=dev)
    self.views = [self.flat[o:o + n].view_as(p) for p, n, o in zip(self.params, sizes, starts)]
    self.extra = self.flat[starts[-1]:starts[-1] + extra]   # e.g. the per-point `visible` accumulator
    self._starts = starts
    self.attach()

  def attach(self):
    """Point every param's .grad at its slice of the flat buffer (autograd then accumulates in place)."""
    for p, v in zip(self.params, self.views):
      p.grad = v

  def zero(self, except_views: Sequence[int] = ()):
    """Zero-fills the buffer; ``except_views`` lists parameter slots to leave alone (e.g. the feature gradient when
    ``GradOut.feature_uninitialized`` lets the SH backward overwrite it)."""
    if not except_views:
      self.flat.zero_()
      return
    cuts = sorted((self._starts[i], self._starts[i] + self.params[i].numel()) for i in except_views)
    at = 0
    for lo, hi in cuts:
      if lo > at:
        self.flat[at:lo].zero_()
      at = hi
    if at < self.flat.numel():
      self.flat[at:].zero_()

  def all_reduce(self, group=None, mode: str = "all_reduce", async_op: bool = False):
    """Sums the buffer over the ranks.  ``async_op=True`` (all_reduce mode only) returns the work handle instead of
    making the current stream wait."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
      return None
    ws = dist.get_world_size(group)
    if mode == "reduce_scatter" and dist.get_backend(group) != "gloo":
      shard = self.flat.numel() // ws
      out = torch.empty(shard, dtype=torch.float32, device=self.flat.device)
      dist.reduce_scatter_tensor(out, self.flat, op=dist.ReduceOp.SUM, group=group)
      dist.all_gather_into_tensor(self.flat, out, group=group)
      return None
    work = dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if async_op else None


def exchange_sh_factors(collector, camera_slots: Sequence[int], cameras_per_rank: int, sh_features: torch.Tensor,
                        positions: torch.Tensor, d_sh: torch.Tensor, d_pos: Optional[torch.Tensor], group=None,
                        accumulate: bool = True, after=None):
  """All-gathers the colour-gradient factors recorded by a ``ShFactorCollector`` and adds the summed SH coefficient
  gradient of ALL cameras of the batch to ``d_sh`` (N,3,K) -- and the view-direction term to ``d_pos`` (N,3) -- on every
  rank.  ``d_sh`` / ``d_pos`` therefore must NOT be all-reduced afterwards.  ``accumulate=False``: ``d_sh`` is
  overwritten row for row instead (no zero-fill by the caller, no read of the old contents).  ``after``: a
  ``torch.distributed`` work handle to wait for before ``d_pos`` is touched (``GradBucket.all_reduce(async_op=True)``):
  the scatter of the factors and their all-gather are then enqueued while that all-reduce is still in flight.

  Why: at K = 16 the coefficient gradient is 48 of the 59 floats per splat that a gradient all-reduce moves; its
  per-camera factors (3 floats per splat + 3 per camera) are 16x smaller, positions and coefficients are replicated,
  and xGMI traffic, not the rebuild (one pass over the rows), is what bounds data-parallel scaling here.  The sum runs
  over cameras in slot order on every rank: deterministic and identical everywhere.

  ``camera_slots[i]``: slot (0..cameras_per_rank-1) of the i-th recorded camera on this rank; every rank contributes
  exactly ``cameras_per_rank`` slots (unused ones stay zero), so the gather is one fixed-size collective."""
  import ctypes as C
  from . import _lib
  lib = _lib.load()
  N, _, K = sh_features.shape
  block = gather_sh_factors(collector, camera_slots, cameras_per_rank, N, group=group)
  if after is not None:
    after.wait()                     # e.g. the asynchronous all-reduce that delivers d_pos
  stride = (N + 1) * 3
  base = block.data_ptr()
  ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
  _lib.check(lib.gsr_sh_backward_multi(C.c_void_p(base), stride, C.c_void_p(base + 4 * 3 * N), stride, block.shape[0],
                                       ptr(sh_features.detach()), ptr(positions.detach()), N, K, ptr(d_sh), ptr(d_pos),
                                       int(accumulate), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
             "gsr_sh_backward_multi")
  collector.clear()


def gather_sh_factors(collector, camera_slots: Sequence[int], cameras_per_rank: int, num_points: int, group=None):
  """The collective half of ``exchange_sh_factors``: scatters this rank's recorded colour gradients to dense rows and
  all-gathers them in ONE collective.  Every camera slot is a (N+1, 3) block: rows 0..N-1 the colour gradient of the
  scene rows (zero where the camera saw nothing), row N the camera position.  Returns the (ws * cameras_per_rank, N+1, 3)
  block, rank-major, identical on every rank.  Pure torch + torch.distributed (works on gloo/CPU)."""
  if len(collector.items) != len(camera_slots):
    raise ValueError(f"{len(collector.items)} recorded cameras but {len(camera_slots)} slots")
  if len(collector.items) > cameras_per_rank:
    raise ValueError("more recorded cameras than cameras_per_rank")
  dev = collector.items[0][1].device if collector.items else None
  mine = torch.zeros(cameras_per_rank, num_points + 1, 3, dtype=torch.float32, device=dev)
  for slot, (idx, dcol, cam) in zip(camera_slots, collector.items):
    mine[slot].index_copy_(0, idx, dcol)
    mine[slot, num_points].copy_(cam)
  ws = dist.get_world_size(group) if dist.is_initialized() else 1
  if ws == 1:
    return mine
  block = torch.empty(ws * cameras_per_rank, num_points + 1, 3, dtype=torch.float32, device=dev)
  dist.all_gather_into_tensor(block, mine, group=group)
  return block


def gather_point_stats(local: List[dict], num_cameras: int, group=None) -> List[dict]:
  """All-gathers the per-camera point statistics (idx, screen_scale_max, visibility, split_score,
  prune_cost) and returns them for ALL cameras in camera order, identical on every rank.

  PointState.add_rendering (controller/point_state.py:34-50) blends with exp_lerp, which depends on
  the order cameras are applied; replaying in camera order on every rank keeps the controller state
  bit-identical to the sequential loop.  ``local`` holds dicts with a ``camera`` key."""
  if not dist.is_initialized() or dist.get_world_size(group) == 1:
    return sorted(local, key=lambda d: d["camera"])
  ws = dist.get_world_size(group)
  gathered: List[Optional[list]] = [None] * ws
  cpu_local = [{k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in d.items()} for d in local]
  dist.all_gather_object(gathered, cpu_local, group=group)
  flat = [d for part in gathered for d in part]
  flat.sort(key=lambda d: d["camera"])
  assert [d["camera"] for d in flat] == list(range(num_cameras)), "camera shards do not partition the batch"
  return flat


def evaluate_backward_sharded(params: Sequence[torch.Tensor], cameras: Sequence, render_loss_fn: Callable,
                              bucket: Optional[GradBucket] = None, group=None, mode: str = "reduce_scatter"):
  """Multi-GPU form of trainer.py:500-514.  Each rank renders + backprops its cameras
  (``render_loss_fn(camera_index, camera) -> (loss, stats_dict)`` must call ``loss.backward()`` itself or
  return a loss to be backpropagated here), gradients accumulate across the rank's cameras exactly as
  in the reference (no zeroing between cameras), then ONE fused collective sums them over ranks.
  Returns the per-camera stats of the whole batch in camera order."""
  rank = dist.get_rank(group) if dist.is_initialized() else 0
  ws = dist.get_world_size(group) if dist.is_initialized() else 1
  local_stats = []
  for j in shard_cameras(len(cameras), rank, ws):
    with torch.enable_grad():
      loss, stats = render_loss_fn(j, cameras[j])
      if loss is not None and loss.requires_grad:
        loss.backward()
    stats = dict(stats or {})
    stats["camera"] = j
    local_stats.append(stats)
  if bucket is not None:
    bucket.all_reduce(group=group, mode=mode)
  elif ws > 1:
    for p in params:
      if p.grad is not None:
        dist.all_reduce(p.grad, op=dist.ReduceOp.SUM, group=group)
  return gather_point_stats(local_stats, len(cameras), group=group)
