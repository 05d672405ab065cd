"""Data-parallel seam of the rasterizer path: shard the cameras of a batch over ranks (one process per
GPU), sum the Gaussian parameter gradients with ONE fused collective over RCCL/xGMI, and replay the
per-camera point statistics in camera order so densification stays bit-identical to a 1-GPU run.

The reference has no distributed code; the seam is its sequential per-camera loop with in-place grad
accumulation, splat_trainer/trainer/trainer.py:500-514 (``evaluate_backward_with``).  Parameters (and
optimizer state) are fully replicated; a camera's render is not split across GPUs.

Collective choice.  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so what matters is that the
message is ONE large contiguous buffer that RCCL can split over all links/channels at once, not many
per-tensor calls each bound by launch latency: all gradients are packed into one contiguous fp32 buffer
(padded to a multiple of world_size; 120 MB at 500k splats/SH3, 708 MB at 3M).

ONE default everywhere (``DEFAULT_COLLECTIVE = "sh_factor"``, used by bench.py and ``CameraShardedStep``): TWO
collectives per batch and no host round trip -- a fused all-reduce of the geometry gradients + visible accumulator +
in-view count (13 floats per splat), and ONE all-gather of a fixed-size block per camera (6 floats per point: the
colour-gradient factors from which every rank rebuilds the SH coefficient gradient, the two order-dependent controller
scores, the screen-scale maximum; csrc/densify.hip: dp_pack / dp_replay).  The block is dense over the N points, so
its size is known without exchanging visible counts first; ``packed=True`` selects the older form that pads to the
largest visible count of the batch (smaller messages once the frustum cull removes most of the scene, at the price
of a count exchange and a host sync per batch).  ``"all_reduce"`` (the whole 59-float buffer in one fused all-reduce) and
``"reduce_scatter"`` (``reduce_scatter_tensor`` + ``all_gather_into_tensor`` on the same buffer; gloo falls back to
all_reduce) stay selectable for comparison.  None could be timed on xGMI in this build environment (one GPU).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


DEFAULT_COLLECTIVE = "sh_factor"
STAT_FIELDS = ("screen_scale_max", "visibility", "split_score", "prune_cost")


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
    # the extra columns come FIRST: with the (large, last) feature gradient exempt from the zero-fill, everything that
    # does need zeroing is then one contiguous range -- one fill launch per batch instead of two
    for n in [extra] + sizes:
      starts.append(off)
      off += ((n + align - 1) // align) * align
    total = off
    self.padded = ((total + self.world_size - 1) // self.world_size) * self.world_size
    dev = self.params[0].device
    self.flat = torch.zeros(self.padded, dtype=torch.float32, device=dev)
    self.extra = self.flat[:extra]                    # e.g. the per-point `visible` accumulator
    starts = starts[1:]
    self.views = [self.flat[o:o + n].view_as(p) for p, n, o in zip(self.params, sizes, starts)]
    self._starts = starts
    self._used = starts[-1] + sizes[-1] if sizes else extra        # alignment / world-size padding behind it is never read
    self.attach()

  def attach(self):
    """Point every param's .grad at its slice of the flat buffer (autograd then accumulates in place)."""
    for p, v in zip(self.params, self.views):
      p.grad = v

  def zero(self, except_views: Sequence[int] = ()):
    """Zero-fills the buffer; ``except_views`` lists parameter slots to leave alone (their first backward pass of the
    batch overwrites every row: ``GradOut.geometry_uninitialized`` / ``feature_uninitialized``).  The alignment padding
    between slots is zero from construction and never written, so it is only filled through when that saves a launch."""
    if not except_views:
      self.flat.zero_()
      return
    keep = set(except_views)
    todo = [(0, self.extra.numel())] if self.extra.numel() else []
    todo += [(o, o + p.numel()) for i, (p, o) in enumerate(zip(self.params, self._starts)) if i not in keep]
    merged = []
    for lo, hi in sorted(todo):
      if merged and lo - merged[-1][1] < 64:          # neighbours separated by padding only: one fill
        merged[-1][1] = hi
      else:
        merged.append([lo, hi])
    for lo, hi in merged:
      self.flat[lo:hi].zero_()

  def all_reduce(self, group=None, mode: str = "all_reduce", async_op: bool = False, even_single: bool = False):
    """Sums the buffer over the ranks.  ``async_op=True`` (all_reduce mode only) returns the work handle instead of
    making the current stream wait.  ``even_single``: issue the collective on a one-rank group too (tests)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not even_single):
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
                        accumulate: bool = True, after=None, visible_max: Optional[int] = None):
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
  if visible_max is not None and 4 * (visible_max + 1) >= 3 * (N + 1):
    visible_max = None               # packed rows would not be smaller than the dense block
  block = gather_sh_factors(collector, camera_slots, cameras_per_rank, N, group=group, device=positions.device,
                            visible_max=visible_max)
  if after is not None:
    after.wait()                     # e.g. the asynchronous all-reduce that delivers d_pos
  stride = (N + 1) * 3
  base = block.data_ptr()
  ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
  _lib.check(lib.gsr_sh_backward_multi(C.c_void_p(base), stride, C.c_void_p(base + 4 * 3 * N), stride, block.shape[0],
                                       ptr(sh_features.detach()), ptr(positions.detach()), N, K, ptr(d_sh), ptr(d_pos),
                                       int(accumulate), _lib.current_stream_ptr()),
             "gsr_sh_backward_multi")
  collector.clear()


def exchange_counts(local: Sequence[Sequence[int]], slots_per_rank: int, device, group=None) -> List[List[int]]:
  """All-gathers a few integers per camera slot (e.g. camera index, visible count) and returns them for every slot of
  every rank, rank-major, as host ints: the sizes the packed exchanges below are padded to.  One tiny collective and
  one read-back per batch (the only host sync of the exchange); unused slots hold -1."""
  width = len(local[0]) if local else 2
  mine = torch.full((slots_per_rank, width), -1, dtype=torch.int64)
  for s, row in enumerate(local):
    mine[s] = torch.tensor(list(row), dtype=torch.int64)
  ws = dist.get_world_size(group) if dist.is_initialized() else 1
  if ws == 1:
    return mine.tolist()
  mine = mine.to(device)
  out = torch.empty(ws * slots_per_rank, width, dtype=torch.int64, device=device)
  dist.all_gather_into_tensor(out, mine, group=group)
  return out.cpu().tolist()


def gather_sh_factors(collector, camera_slots: Sequence[int], cameras_per_rank: int, num_points: int, group=None,
                      device=None, visible_max: Optional[int] = None):
  """The collective half of ``exchange_sh_factors``: ONE all-gather carries every camera's colour gradients and camera
  position.  Returns the dense (ws * cameras_per_rank, N+1, 3) block the rebuild kernel reads -- rows 0..N-1 the colour
  gradient of the scene rows (zero where the camera saw nothing), row N the camera position -- rank-major, identical on
  every rank.  Pure torch + torch.distributed (works on gloo/CPU).

  ``visible_max`` = None: the dense block itself goes over the wire (12 (N+1) bytes per camera).  ``visible_max`` = the
  largest visible count of the batch (from ``exchange_counts``): packed rows [index, r, g, b] padded to that count go
  over the wire (16 (visible_max+1) bytes per camera) and every rank scatters them into the dense block -- the cheaper
  form once the frustum cull removes more than a quarter of the scene.  ``device``: where the block lives (required
  when this rank recorded no camera)."""
  if len(collector.items) != len(camera_slots):
    raise ValueError(f"{len(collector.items)} recorded cameras but {len(camera_slots)} slots")
  if len(collector.items) > cameras_per_rank:
    raise ValueError("more recorded cameras than cameras_per_rank")
  dev = device if device is not None else (collector.items[0][1].device if collector.items else None)
  if dev is None:
    raise ValueError("gather_sh_factors: pass device= (this rank recorded no camera to take it from)")
  ws = dist.get_world_size(group) if dist.is_initialized() else 1
  N = num_points
  if visible_max is None:
    mine = torch.zeros(cameras_per_rank, N + 1, 3, dtype=torch.float32, device=dev)
    for slot, (idx, dcol, cam, *_) in zip(camera_slots, collector.items):
      mine[slot].index_copy_(0, idx, dcol)
      mine[slot, N].copy_(cam)
    if ws == 1:
      return mine
    block = torch.empty(ws * cameras_per_rank, N + 1, 3, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(block, mine, group=group)
    return block
  # packed: field-major (4, visible_max + 1); entry 0 of every field is the header [count | camera x, y, z]
  L = int(visible_max) + 1
  mine = torch.zeros(cameras_per_rank, 4, L, dtype=torch.float32, device=dev)
  for slot, (idx, dcol, cam, *_) in zip(camera_slots, collector.items):
    m = idx.shape[0]
    if m > visible_max:
      raise ValueError(f"camera with {m} visible rows but visible_max = {visible_max}")
    head = torch.tensor([m], dtype=torch.int32, device=dev)
    mine[slot, 0, :1] = head.view(torch.float32)
    mine[slot, 0, 1:1 + m] = idx.to(torch.int32).view(torch.float32)
    mine[slot, 1:, 0] = cam
    mine[slot, 1:, 1:1 + m] = dcol.t()
  if ws == 1:
    packed = mine
  else:
    packed = torch.empty(ws * cameras_per_rank, 4, L, dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(packed, mine, group=group)
  counts = packed[:, 0, 0].contiguous().view(torch.int32).tolist()      # host sync; tiny
  block = torch.zeros(packed.shape[0], N + 1, 3, dtype=torch.float32, device=dev)
  for c, m in enumerate(counts):
    if m > 0:
      rows = packed[c, 0, 1:1 + m].contiguous().view(torch.int32).long()
      block[c].index_copy_(0, rows, packed[c, 1:, 1:1 + m].t())
    block[c, N] = packed[c, 1:, 0]
  return block


def gather_point_stats(local: List[dict], num_cameras: int, group=None, device=None,
                       counts: Optional[List[List[int]]] = None) -> List[dict]:
  """All-gathers the per-camera point statistics -- ``idx`` and the STAT_FIELDS (screen_scale_max, visibility,
  split_score, prune_cost; a missing field counts as zeros) -- and returns them for ALL cameras in camera order,
  identical on every rank, as tensors on ``device``.

  PointState.add_rendering (controller/point_state.py:34-50) blends with exp_lerp, which depends on the order cameras
  are applied; replaying in camera order on every rank keeps the controller state bit-identical to the sequential
  loop.  ``local`` holds dicts with a ``camera`` key.  Everything stays on the device: one field-major block
  (5, visible_max) per camera slot -- row index bits + the four statistics -- padded to the largest visible count of the
  batch, ONE ``all_gather_into_tensor``; the counts come from ``exchange_counts`` (passed in when the caller already
  has them, e.g. shared with the factor exchange)."""
  ws = dist.get_world_size(group) if dist.is_initialized() else 1
  if ws == 1:
    return sorted(local, key=lambda d: d["camera"])
  cpr = (num_cameras + ws - 1) // ws
  if len(local) > cpr:
    raise ValueError("more local cameras than ceil(num_cameras / world_size)")
  dev = device if device is not None else (local[0]["idx"].device if local else None)
  if dev is None:
    raise ValueError("gather_point_stats: pass device= (this rank rendered no camera to take it from)")
  if counts is None:
    counts = exchange_counts([(d["camera"], d["idx"].shape[0]) for d in local], cpr, dev, group=group)
  L = max(max(m for _, m in counts), 1)
  mine = torch.zeros(cpr, 1 + len(STAT_FIELDS), L, dtype=torch.float32, device=dev)
  for s, d in enumerate(local):
    m = d["idx"].shape[0]
    mine[s, 0, :m] = d["idx"].to(torch.int32).view(torch.float32)
    for f, name in enumerate(STAT_FIELDS):
      if d.get(name) is not None:
        mine[s, 1 + f, :m] = d[name].detach().to(torch.float32)
  block = torch.empty(ws * cpr, 1 + len(STAT_FIELDS), L, dtype=torch.float32, device=dev)
  dist.all_gather_into_tensor(block, mine, group=group)
  out = []
  for s, (cam, m) in enumerate(counts):
    if cam < 0:
      continue
    d = dict(camera=int(cam), idx=block[s, 0, :m].view(torch.int32).long())
    for f, name in enumerate(STAT_FIELDS):
      d[name] = block[s, 1 + f, :m]
    out.append(d)
  out.sort(key=lambda d: d["camera"])
  assert [d["camera"] for d in out] == list(range(num_cameras)), "camera shards do not partition the batch"
  return out


def point_stats_of(camera: int, points) -> dict:
  """The statistics of one rendered camera the controller consumes (point_state.py:34-50), read AFTER backward."""
  return dict(camera=camera, idx=points.idx, screen_scale_max=points.screen_scale.max(1).values,
              visibility=points.visibility, split_score=points.split_score, prune_cost=points.prune_cost)


def replay_point_stats(state, stats: Sequence[dict]):
  """Applies the gathered statistics to a controller_math.PointState in camera order -- the reference's own
  per-camera ``add_rendering`` arithmetic, run on every rank for every camera of the batch."""
  from .data_types import RenderedPoints, Rendering
  for d in stats:
    m = d["idx"].shape[0]
    z = torch.zeros(m, dtype=torch.float32, device=d["idx"].device)
    scale = d["screen_scale_max"] if d.get("screen_scale_max") is not None else z
    pts = RenderedPoints(idx=d["idx"], depths=z[:, None], opacity=z,
                         screen_scale=scale if scale.is_cuda else torch.stack([scale, scale], dim=1),
                         visibility=d["visibility"], prune_cost=d["prune_cost"] if d.get("prune_cost") is not None else z,
                         split_score=d["split_score"] if d.get("split_score") is not None else z)
    state.add_rendering(Rendering(image=None, camera=None, points=pts))
  return state


def evaluate_backward_sharded(params: Sequence[torch.Tensor], cameras: Sequence, render_loss_fn: Callable,
                              bucket: Optional[GradBucket] = None, group=None, mode: str = "all_reduce", device=None):
  """Multi-GPU form of trainer.py:500-514.  Each rank renders + backprops its cameras
  (``render_loss_fn(camera_index, camera) -> (loss, stats_dict)`` must call ``loss.backward()`` itself or
  return a loss to be backpropagated here), gradients accumulate across the rank's cameras exactly as
  in the reference (no zeroing between cameras), then ONE fused collective sums them over ranks.
  Returns the per-camera stats of the whole batch in camera order.  This is the generic form (any render function,
  whole-buffer collective); ``CameraShardedStep`` is the product path with the factor exchange."""
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
  dev = device if device is not None else (params[0].device if len(params) else None)
  return gather_point_stats(local_stats, len(cameras), group=group, device=dev)


def accumulate_local_stats(points, scale_max: torch.Tensor, visible: torch.Tensor, in_view: torch.Tensor):
  """Folds one LOCAL camera into the order-independent reductions of the light exchange: running max of the screen
  scale, summed visibility (= the scene's ``visible`` accumulator, mlp_scene.py:244) and the count of cameras that saw
  the point (point_state.py:40).  ``idx`` rows are unique per camera."""
  idx, vis = points.idx, points.visibility
  scale_max[idx] = torch.maximum(scale_max[idx], points.screen_scale.max(1).values)
  visible.index_add_(0, idx, vis)
  in_view.index_add_(0, idx, (vis > 0).to(visible.dtype))


def exchange_point_scores(state, local: List[dict], num_cameras: int, num_points: int, group=None, device=None,
                          counts: Optional[List[List[int]]] = None):
  """The order-dependent half of the controller statistics: all-gathers every camera's ``split_score`` / ``prune_cost``
  (+ the row indexes unless every camera sees every point) in ONE packed ``all_gather_into_tensor`` and applies the two
  ``exp_lerp`` EMAs to ``state`` in camera order on every rank (point_state.py:49-50).  2 (or 3) floats per visible
  splat and camera instead of the 5 of ``gather_point_stats``; the max / sum / count quantities are order independent
  and travel as reductions (``CameraShardedStep``)."""
  ws = dist.get_world_size(group) if dist.is_initialized() else 1
  if ws == 1:
    for d in sorted(local, key=lambda d: d["camera"]):
      state.add_scores(d["idx"], d["split_score"], d["prune_cost"])
    return state
  cpr = (num_cameras + ws - 1) // ws
  dev = device if device is not None else (local[0]["idx"].device if local else None)
  if dev is None:
    raise ValueError("exchange_point_scores: pass device= (this rank rendered no camera to take it from)")
  if counts is None:
    counts = exchange_counts([(d["camera"], d["idx"].shape[0]) for d in local], cpr, dev, group=group)
  dense = all(m == num_points for cam, m in counts if cam >= 0)          # nothing culled anywhere: rows are 0..N-1
  first = 0 if dense else 1
  L = max(max(m for _, m in counts), 1)
  mine = torch.zeros(cpr, first + 2, L, dtype=torch.float32, device=dev)
  for s, d in enumerate(local):
    m = d["idx"].shape[0]
    if not dense:
      mine[s, 0, :m] = d["idx"].to(torch.int32).view(torch.float32)
    mine[s, first, :m] = d["split_score"].detach().to(torch.float32)
    mine[s, first + 1, :m] = d["prune_cost"].detach().to(torch.float32)
  block = torch.empty(ws * cpr, first + 2, L, dtype=torch.float32, device=dev)
  dist.all_gather_into_tensor(block, mine, group=group)
  order = sorted((cam, s, m) for s, (cam, m) in enumerate(counts) if cam >= 0)
  assert [c for c, _, _ in order] == list(range(num_cameras)), "camera shards do not partition the batch"
  for _, s, m in order:
    idx = None if dense else block[s, 0, :m].view(torch.int32).long()
    state.add_scores(idx, block[s, first, :m], block[s, first + 1, :m])
  return state


def position_term_done(collector, K: int) -> bool:
  """Has the position term of the colour gradient been added to the position gradient by the local backward passes
  (render_gaussians' fused node does, from the Jacobian its forward pass saved), or is it left to the multi-camera
  rebuild (the three-call form: evaluate_sh_at hands on colour gradients only)?  Recorded per camera by the backward
  node itself (ShFactorCollector.items[i][3]); every camera of every rank must have gone the same way."""
  flags = {bool(it[3]) for it in collector.items if len(it) > 3}
  if len(flags) > 1:
    raise RuntimeError("the cameras of one batch mix call forms that add the colour gradient's position term locally with "
                       "forms that leave it to the rebuild: use ONE form of the render call in render_backward")
  if not flags:                       # this rank rendered nothing (or an older collector): the configured behaviour
    return bool(collector.position_term_local) or K == 1
  return flags.pop() or K == 1


class CameraShardedStep:
  """One data-parallel batch of the hot path, the way bench.py and the tests run it: this rank renders and
  back-propagates ITS cameras of the batch (camera j -> rank j mod world) with the gradients accumulating straight into
  one flat buffer, then the default exchange (``DEFAULT_COLLECTIVE``) delivers on every rank the summed gradients of all
  cameras and -- in camera order -- the per-camera controller statistics.  With one rank nothing is communicated.

      step = CameraShardedStep(params, world, rank)          # params: position, log_scaling, rotation, alpha_logit, feature
      stats = step.run(cameras, render_backward)              # render_backward(j, camera, grad_out, sh_collector) -> Rendering
      step.grads                                              # name -> summed gradient tensor
  """
  NAMES = ("position", "log_scaling", "rotation", "alpha_logit", "feature")

  def __init__(self, params: Sequence[torch.Tensor], world_size: int, rank: int, group=None,
               mode: str = DEFAULT_COLLECTIVE, with_stats: bool = True, packed: bool = False,
               exchange_when_single: bool = False, fused_grad_out: bool = True, position_term_local: bool = True,
               sharded_replay: bool = True, early_gather: bool = True):
    """``packed``: exchange rows padded to the batch's largest visible count instead of the dense per-point block
    (needs a count exchange + host sync per batch).  ``exchange_when_single``: run the exchange even with one rank
    (tests: packing, the collectives on a one-rank group, rebuild and replay on a single GPU).
    ``fused_grad_out`` (default): ``render_backward`` hands ``grad_out`` on to ``render_gaussians`` /
    ``project_to_image``, whose first backward pass of a batch initialises the gradient buffers itself (every row
    written, zeros where its camera saw nothing): no zero-fill per batch.  False: the buffers are zero-filled per batch
    and the callback may accumulate into ``param.grad`` any way it likes.
    ``sharded_replay`` (default): the order-dependent controller scores are exchanged by POINTS, not replayed on every
    rank -- all-to-all of each rank's N / G slice of every camera's scores, in-order replay on the slice, all-gather of the
    2-float state -- and the screen-scale maximum travels as a MAX all-reduce (``_exchange_sharded``); False: the round-3
    form, one dense 6 N + 3 block per camera all-gathered and replayed on every rank (``_exchange_dense``).
    ``early_gather`` (default, sharded form with ``render_gaussians``' fused node): the colour-factor all-gather is
    started from inside this rank's last backward pass of the batch, right behind K7 + the per-splat reduction, and runs
    next to the geometry sweep (``gsr_frame_backward_stages``); False: it is issued with the other collectives.
    ``position_term_local`` (default; factor exchange only): every rank adds the position term of its own cameras'
    colour gradient before the all-reduce -- ``render_gaussians(use_sh=True, sh_collector=...)`` does, from the Jacobian
    its forward pass saves -- and the multi-camera rebuild neither recomputes it for all cameras on every rank nor
    reads the coefficient rows.  False: the fused node does not save the Jacobian and the rebuild adds the term.  A
    callback in the three-call form (``evaluate_sh_at`` hands on colour gradients only) needs no flag: every backward
    node records whether it added the term (``position_term_done``) and the rebuild follows.  Every rank must run the
    same form of the render call."""
    self.fused = bool(fused_grad_out)
    from .renderer import GradOut
    from .sh import ShFactorCollector
    self.params = list(params)
    self.world, self.rank, self.group, self.mode, self.with_stats = max(world_size, 1), rank, group, mode, with_stats
    self.packed = packed
    self.sharded = bool(sharded_replay)
    self.early_gather = bool(early_gather)      # start the factor all-gather from inside the last local backward pass
    self.exchange = self.world > 1 or exchange_when_single
    self.factor = mode == "sh_factor" and self.exchange
    self._slots = {}
    N = self.params[0].shape[0]
    # extra: [visible accumulator (mlp_scene.py:244) | number of cameras that saw the point] -- both sums, so they ride
    # in the gradient all-reduce; the screen-scale maximum needs a MAX all-reduce of its own
    self.bucket = GradBucket(self.params[:4] if self.factor else self.params, self.world, extra=2 * N)
    self.scale_max = torch.zeros(N, dtype=torch.float32, device=self.params[0].device) if self.exchange else None
    self.feature_grad = torch.empty_like(self.params[4]) if self.factor else None
    self.collector = ShFactorCollector() if self.factor else None
    if self.collector is not None:
      self.collector.position_term_local = bool(position_term_local)
    v = self.bucket.views
    self.grad_out = GradOut(position=v[0], log_scaling=v[1], rotation=v[2], alpha_logit=v[3],
                            feature=None if self.factor else v[4])

  @property
  def grads(self) -> dict:
    v = self.bucket.views
    return dict(zip(self.NAMES, list(v[:4]) + [self.feature_grad if self.factor else v[4]]))

  @property
  def visible(self) -> torch.Tensor:
    return self.bucket.extra[:self.params[0].shape[0]]

  def run(self, cameras: Sequence, render_backward: Callable, point_state=None) -> List[dict]:
    """``point_state`` (controller_math.PointState): the controller's statistics are updated in place for every camera
    of the batch -- locally with one fused launch per camera on one rank, through the LIGHT exchange on several (max /
    sum / count as reductions, the two EMA inputs per camera all-gathered and replayed in camera order) -- and nothing
    is returned.  Without it (and ``with_stats``) the full per-camera statistics of the batch are returned instead."""
    position, feature = self.params[0], self.params[4]
    dev = position.device
    N = position.shape[0]
    light = point_state is not None
    mine = shard_cameras(len(cameras), self.rank, self.world)
    dense = self.factor and light and not self.packed
    if self.fused:
      # Only the two sum columns are zero-filled: the first backward pass of the batch overwrites every row of the four
      # geometry gradients (and, when it is formed here, of the feature gradient) -- zeros where its camera saw nothing
      self.bucket.zero(except_views=(0, 1, 2, 3) if self.factor else (0, 1, 2, 3, 4))
      self.grad_out.begin_batch(geometry=True, feature=not self.factor)
    else:
      self.bucket.zero()
      self.grad_out.begin_batch(geometry=False, feature=False)
    local = []
    if light and self.exchange and not dense:
      self.scale_max.zero_()
    pack_now = dense and self.sharded
    if pack_now:
      self._sharded_begin(len(cameras))
    for j in mine:
      r = render_backward(j, cameras[j], self.grad_out, self.collector)
      if self.grad_out.geometry_uninitialized:
        raise RuntimeError("render_backward did not route grad_out into the renderer's backward pass (the gradient "
                           "buffers of this batch are uninitialised): pass grad_out on, or construct "
                           "CameraShardedStep(fused_grad_out=False)")
      if pack_now:
        # packed at once: nothing of the frame (its per-point outputs are views of the frame's whole output arena) has to
        # stay alive until the end of the batch
        self._sharded_pack(len(local), r.points)
        local.append(j)
        continue
      if dense:                               # sums ride in the all-reduce; scores + scale go into the camera's block
        local.append(dict(camera=j, idx=r.points.idx, split_score=r.points.split_score, prune_cost=r.points.prune_cost,
                          screen_scale=r.points.screen_scale, visibility=r.points.visibility))
        continue
      if light and self.exchange:
        accumulate_local_stats(r.points, self.scale_max, self.bucket.extra[:N], self.bucket.extra[N:])
        local.append(dict(camera=j, idx=r.points.idx, split_score=r.points.split_score, prune_cost=r.points.prune_cost))
        continue
      if light:                                 # one rank: the reference's own loop; mlp_scene.py:244 rides in the launch
        point_state.add_rendering(r, visible_sum=self.bucket.extra[:N])
        continue
      self.bucket.extra[:N].index_add_(0, r.points.idx, r.points.visibility)      # mlp_scene.py:244
      if self.with_stats:
        local.append(point_stats_of(j, r.points))
    # A buffer no backward pass reached (this rank had no camera in the batch, or none of its cameras reached that buffer)
    # holds nothing: its share of the sum is zero -- and it must BE zero before a collective adds it to the others
    self.grad_out.finish_batch()
    if not self.exchange:
      return [] if light else local
    if dense:
      (self._exchange_sharded if self.sharded else self._exchange_dense)(len(cameras), local, point_state)
      return []
    cpr = (len(cameras) + self.world - 1) // self.world
    counts = exchange_counts([(d["camera"], d["idx"].shape[0]) for d in local] if (self.with_stats or light) else
                             [(j, it[0].shape[0]) for j, it in zip(mine, self.collector.items)] if self.factor else
                             [(j, 0) for j in mine], cpr, dev, group=self.group)
    if self.factor:
      pending = self.bucket.all_reduce(group=self.group, mode="all_reduce", async_op=True)
      exchange_sh_factors(self.collector, list(range(len(mine))), cpr, feature, position, self.feature_grad,
                          None if position_term_done(self.collector, feature.shape[2]) else self.bucket.views[0], group=self.group,
                          accumulate=False, after=pending,
                          visible_max=max(m for _, m in counts))
    else:
      self.bucket.all_reduce(group=self.group, mode=self.mode)
    if light:
      dist.all_reduce(self.scale_max, op=dist.ReduceOp.MAX, group=self.group)
      torch.maximum(point_state.max_scale_px, self.scale_max, out=point_state.max_scale_px)
      point_state.visibility += self.bucket.extra[:N]                 # summed over the ranks by the gradient all-reduce
      point_state.points_in_view += self.bucket.extra[N:].to(point_state.points_in_view.dtype)
      exchange_point_scores(point_state, local, len(cameras), N, group=self.group, device=dev, counts=counts)
      return []
    if not self.with_stats:
      return []
    return gather_point_stats(local, len(cameras), group=self.group, device=dev, counts=counts)

  # ------------------------------------------------------------------------------------------ dense exchange
  def camera_slots(self, num_cameras: int, device) -> torch.Tensor:
    """Row of the gathered block that holds camera c, for c = 0..num_cameras-1 (camera j lives on rank j mod world, in
    that rank's slot j // world; the gather is rank-major)."""
    key = (num_cameras, str(device))
    if key not in self._slots:
      cpr = (num_cameras + self.world - 1) // self.world
      self._slots[key] = torch.tensor([(j % self.world) * cpr + j // self.world for j in range(num_cameras)],
                                      dtype=torch.int32, device=device)
    return self._slots[key]

  def pack_camera_blocks(self, num_cameras: int, local: List[dict], factors: Sequence) -> torch.Tensor:
    """This rank's cameras as fixed-size blocks (densify.dp_pack), (cameras_per_rank, 6 N + 3).  ``factors``: the SH
    collector's items (idx, d_colour, camera position) in the order of ``local``; a ``visibility`` entry in a ``local``
    dict also adds that camera to the two sum columns of the gradient bucket (visible accumulator | in-view count).
    Unused slots carry an empty camera (zero gradient, NaN scores)."""
    from .densify import dp_block_floats, dp_pack
    position = self.params[0]
    N, dev = position.shape[0], position.device
    cpr = (num_cameras + self.world - 1) // self.world
    send = torch.empty(cpr, dp_block_floats(N), dtype=torch.float32, device=dev)
    empty_i = torch.empty(0, dtype=torch.int64, device=dev)
    empty_f = torch.empty(0, dtype=torch.float32, device=dev)
    for s in range(cpr):
      if s < len(local):
        d, (idx, d_colour, cam, *_) = local[s], factors[s]
        dp_pack(send[s], N, idx, d_colour, d["split_score"], d["prune_cost"], d["screen_scale"], cam,
                visibility=d.get("visibility"), sums=self.bucket.extra)
      else:
        dp_pack(send[s], N, empty_i, empty_f.view(0, 3), empty_f, empty_f, empty_f.view(0, 2),
                torch.zeros(3, dtype=torch.float32, device=dev))
    return send

  def all_gather_blocks(self, send: torch.Tensor) -> torch.Tensor:
    """(world * cameras_per_rank, 6 N + 3) block table, rank-major, identical on every rank."""
    if not (dist.is_available() and dist.is_initialized()):
      return send
    recv = torch.empty(self.world * send.shape[0], send.shape[1], dtype=torch.float32, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=self.group)
    return recv

  def _exchange_dense(self, num_cameras: int, local: List[dict], point_state):
    """The default exchange: all-reduce (geometry gradients + visible + in-view count) in flight while the camera blocks
    are packed and all-gathered; then every rank rebuilds the SH gradient of all cameras and replays their controller
    scores in camera order.  Two collectives, no host sync."""
    import ctypes as C
    from . import _lib
    from .densify import dp_replay
    position, feature = self.params[0], self.params[4]
    N, K = position.shape[0], feature.shape[2]
    initialised = dist.is_available() and dist.is_initialized()
    # pack first: it also adds this rank's cameras to the two sum columns the all-reduce carries
    send = self.pack_camera_blocks(num_cameras, local, self.collector.items)
    pending = self.bucket.all_reduce(group=self.group, mode="all_reduce", async_op=True,
                                     even_single=True) if initialised else None
    blocks = self.all_gather_blocks(send)
    if pending is not None:
      pending.wait()                                   # d_pos below adds to the all-reduced position gradient
    width = blocks.shape[1]
    base = blocks.data_ptr()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    # (every rank runs the same render path: the flag is the same everywhere; K = 1 has no position term at all)
    position_done = position_term_done(self.collector, K)
    _lib.check(_lib.load().gsr_sh_backward_multi(C.c_void_p(base), width, C.c_void_p(base + 4 * 3 * N), width,
                                                 blocks.shape[0], ptr(feature.detach()), ptr(position.detach()), N, K,
                                                 ptr(self.feature_grad),
                                                 None if position_done else ptr(self.bucket.views[0]), 0,
                                                 _lib.current_stream_ptr()), "gsr_sh_backward_multi")
    self.collector.clear()
    dp_replay(point_state, blocks, self.camera_slots(num_cameras, position.device), N, sums=self.bucket.extra)

  # ---------------------------------------------------------------------------------------- sharded exchange
  def _sharded_begin(self, num_cameras: int):
    """Send buffers of one batch: (cpr, 3 N + 3) colour-factor blocks and the (G, cpr, 2, L) score slices."""
    from .densify import dp_slice_len
    position = self.params[0]
    N, dev = position.shape[0], position.device
    cpr = (num_cameras + self.world - 1) // self.world
    L = dp_slice_len(N, self.world)
    self._send = (torch.empty(cpr, 3 * N + 3, dtype=torch.float32, device=dev),
                  torch.empty(self.world, cpr, 2, L, dtype=torch.float32, device=dev), cpr, L)
    self.scale_max.zero_()
    # Early factor gather: the fused node's backward pass calls back right behind K7 + the per-splat reduction; the
    # factor block is packed from the gradient rows there and, once this rank's LAST camera of the batch has been packed,
    # the all-gather is started -- it then runs next to that camera's geometry sweep instead of behind it.
    self._early = dict(packed=0, expected=len(shard_cameras(num_cameras, self.rank, self.world)), work=None, blocks=None)
    self.collector.on_rows = self._early_factors if self.early_gather else None

  def _early_factors(self, indexes, grad_rows, camera_pos):
    from .densify import dp_pack_factors_rows
    factors, scores, cpr, L = self._send
    e = self._early
    N = self.params[0].shape[0]
    dp_pack_factors_rows(factors[e["packed"]], N, indexes, grad_rows, camera_pos)
    e["packed"] += 1
    if e["packed"] == e["expected"]:
      for s_ in range(e["expected"], cpr):             # unused slots: an empty camera
        factors[s_].zero_()
      if dist.is_available() and dist.is_initialized():
        e["blocks"] = torch.empty(self.world * cpr, factors.shape[1], dtype=torch.float32, device=factors.device)
        e["work"] = dist.all_gather_into_tensor(e["blocks"], factors, group=self.group, async_op=True)
      else:
        e["blocks"] = factors

  def _sharded_pack(self, slot: int, points):
    """Camera ``slot`` of this rank into the send buffers (``points`` None: an empty camera), right behind its backward
    pass; also adds the camera to the two sum columns the gradient all-reduce carries."""
    from .densify import dp_pack_sharded
    factors, scores, cpr, L = self._send
    N, dev = self.params[0].shape[0], self.params[0].device
    if points is None:
      e_i, e_f = torch.empty(0, dtype=torch.int64, device=dev), torch.empty(0, dtype=torch.float32, device=dev)
      early = self._early["blocks"] is not None          # (the gather has left: its unused slots were zero-filled there)
      dp_pack_sharded(None if early else factors[slot], scores, self.scale_max, N, slot, e_i, e_f.view(0, 3), e_f, e_f,
                      e_f.view(0, 2), torch.zeros(3, dtype=torch.float32, device=dev))
      return
    idx, d_colour, cam = self.collector.items[slot][:3]
    early = self._early["packed"] > slot               # this camera's factor block was packed from its gradient rows already
    dp_pack_sharded(None if early else factors[slot], scores, self.scale_max, N, slot, idx, d_colour, points.split_score,
                    points.prune_cost, points.screen_scale, cam, visibility=points.visibility, sums=self.bucket.extra)

  def _exchange_sharded(self, num_cameras: int, local: List[dict], point_state):
    """The default exchange since round 4.  Per batch, G ranks, ``cpr`` camera slots per rank, L = ceil(N / G):

      1. all_reduce SUM (async)   geometry gradients + visible accumulator + in-view count: 13 floats per point
      2. all_reduce MAX (async)   larger screen-space sigma (point_state.py:37): 1 float per point
      3. all_gather               one (3 N + 3)-float block per camera: the colour-gradient factors every rank rebuilds the
                                  summed SH coefficient gradient from (48 of the 59 gradient floats never cross xGMI)
      4. all_to_all               each camera's split_score / prune_cost cut into G slices: rank r receives, of every
                                  camera of the batch, the L points it owns (2 L floats per camera)
      5. (local) the two exp_lerp EMAs of all cameras IN CAMERA ORDER on the L owned points (point_state.py:49-50)
      6. all_gather               the slice's new state, 2 L floats per rank

    No host sync.  Against the dense form (one 6 N + 3 block per camera to every rank) a rank receives 3 N + 2 N / G
    floats per camera instead of 6 N, and the replay costs N / G instead of N point-visits per camera."""
    import ctypes as C
    from . import _lib
    from .densify import dp_finish, dp_replay_slice
    position, feature = self.params[0], self.params[4]
    N, K, dev = position.shape[0], feature.shape[2], position.device
    G = self.world
    live = dist.is_available() and dist.is_initialized()
    factors, scores, cpr, L = self._send
    for s in range(len(local), cpr):         # an unused slot carries an empty camera: zero gradient, NaN scores
      self._sharded_pack(s, None)
    self._send = None
    # (the packs above also added this rank's cameras to the two sum columns the gradient all-reduce carries)
    # ORDER OF ISSUE -- the same on every rank, whatever happened during its backward passes: the factor gather FIRST (a
    # rank whose last camera went through the fused node issued it from inside that backward pass; every other rank --
    # no camera in this batch, a callback in the three-call form, early_gather off -- issues it here, before anything
    # else), then SUM all-reduce, MAX all-reduce, score all-to-all, state gather.
    early = self._early
    self.collector.on_rows = None
    work_gather = None
    if early["blocks"] is not None:                    # left from inside the last backward pass
      blocks, work_gather = early["blocks"], early["work"]
    elif live:
      blocks = torch.empty(G * cpr, 3 * N + 3, dtype=torch.float32, device=dev)
      work_gather = dist.all_gather_into_tensor(blocks, factors, group=self.group, async_op=True)
    else:
      blocks = factors
    pending = self.bucket.all_reduce(group=self.group, mode="all_reduce", async_op=True, even_single=True) if live else None
    pending_max = dist.all_reduce(self.scale_max, op=dist.ReduceOp.MAX, group=self.group, async_op=True) if live else None
    if live:
      recv = torch.empty_like(scores)
      dist.all_to_all_single(recv, scores, group=self.group)
    else:
      recv = scores
    mine = torch.empty(2, L, dtype=torch.float32, device=dev)
    dp_replay_slice(point_state, recv, self.rank, num_cameras, N, mine)
    if live:
      gathered = torch.empty(G, 2, L, dtype=torch.float32, device=dev)
      dist.all_gather_into_tensor(gathered.view(G * 2, L), mine, group=self.group)
    else:
      gathered = mine.view(1, 2, L)
    if work_gather is not None:
      work_gather.wait()
    if pending is not None:
      pending.wait()                                   # d_pos below adds to the all-reduced position gradient
    if pending_max is not None:
      pending_max.wait()
    width = blocks.shape[1]
    base = blocks.data_ptr()
    ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())
    position_done = position_term_done(self.collector, K)
    if blocks.is_cuda:
      _lib.check(_lib.load().gsr_sh_backward_multi(C.c_void_p(base), width, C.c_void_p(base + 4 * 3 * N), width,
                                                   blocks.shape[0], ptr(feature.detach()), ptr(position.detach()), N, K,
                                                   ptr(self.feature_grad),
                                                   None if position_done else ptr(self.bucket.views[0]), 0,
                                                   _lib.current_stream_ptr()), "gsr_sh_backward_multi")
    self.collector.clear()
    dp_finish(point_state, gathered, N, scale_max=self.scale_max, sums=self.bucket.extra)
    self.last_blocks = blocks                          # (tests look at what was gathered)
