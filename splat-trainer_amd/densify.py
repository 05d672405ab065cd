"""Densify / prune on the device: the two operations the reference's densification round
spends its time in, as HIP kernels behind the C ABI (csrc/densify.hip).

  select_n(values, n, descending) ... bool mask of the n smallest / largest entries, ties broken by ascending index --
                                      deterministic replacement of ``take_n`` = ``argsort(t)[:n]`` -> mask
                                      (splat_trainer/controller/target_controller.py:150-160)
  compact_rows(keep_mask, columns) ... ``column[keep_mask]`` followed by ``cat(appended rows)`` for MANY columns in one
                                      pass (scene.split_and_prune, splat_trainer/scene/mlp_scene.py:301-310)

There is no CPU path: CUDA tensors and the HIP library are required.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib


_stream = _lib.current_stream_ptr


def _p(t: Optional[torch.Tensor]):
  return None if t is None or t.numel() == 0 else t.data_ptr()     # plain int: the prototypes declare c_void_p


def select_n(values: torch.Tensor, n: int, descending: bool = False) -> torch.Tensor:
  """Bool mask with exactly ``min(n, N)`` entries set: the n smallest (``descending=False``) or largest values; among
  equal values the lowest indexes win (a stable argsort's choice); NaN orders above +inf."""
  lib = _lib.load()
  if not values.is_cuda:
    raise _lib.GsplatHipError("select_n runs only on a HIP device (got a CPU tensor); there is no CPU fallback")
  if values.dim() != 1:
    raise ValueError("select_n takes a 1-D tensor")
  if n < 0:
    raise ValueError(f"n must be >= 0, got {n}")
  v = values.detach().to(torch.float32).contiguous()
  N = v.shape[0]
  mask = torch.empty(N, dtype=torch.bool, device=v.device)
  if N == 0:
    return mask
  ws_bytes = lib.gsr_select_workspace_bytes(N)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=v.device)
  _lib.check(lib.gsr_select_n(_p(v), N, int(n), 1 if descending else 0, _p(mask), _p(ws), ws_bytes, _stream()),
             "gsr_select_n")
  return mask


def compact_rows(keep_mask: torch.Tensor, columns: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor]]],
                 n_tail: Optional[int] = None) -> List[torch.Tensor]:
  """For every ``(column, tail)``: ``cat([column[keep_mask], tail])`` -- ``tail = None`` appends ``n_tail`` zero rows.
  All columns move in ONE gather launch (after one count + scan and one read-back of the kept count, which sizes the
  outputs).  Columns are 4-byte-element tensors with N rows; a tail has the column's row shape."""
  lib = _lib.load()
  if not keep_mask.is_cuda or keep_mask.dtype != torch.bool or keep_mask.dim() != 1:
    raise _lib.GsplatHipError("compact_rows needs a 1-D CUDA bool mask (no CPU fallback)")
  N = keep_mask.shape[0]
  dev = keep_mask.device
  if n_tail is None:
    tails = [t for _, t in columns if t is not None]
    n_tail = int(tails[0].shape[0]) if tails else 0
  cols = []
  for col, tail in columns:
    if not col.is_cuda or col.element_size() != 4 or col.shape[0] != N:
      raise ValueError("columns must be CUDA tensors of 4-byte elements with one row per mask entry")
    col = col.detach().contiguous()
    if tail is not None:
      if tuple(tail.shape[1:]) != tuple(col.shape[1:]) or tail.shape[0] != n_tail:
        raise ValueError(f"tail of shape {tuple(tail.shape)} does not fit a column of row shape {tuple(col.shape[1:])}")
      tail = tail.detach().to(col.dtype).contiguous()
    cols.append((col, tail))
  if len(cols) > _lib.MAX_COLUMNS:
    raise ValueError(f"at most {_lib.MAX_COLUMNS} columns per call")
  keep = keep_mask.contiguous()
  blocks = max((N + 255) // 256, 1)
  offsets = torch.empty(blocks, dtype=torch.int32, device=dev)
  total = torch.empty(1, dtype=torch.int32, device=dev)
  ws_bytes = lib.gsr_compact_workspace_bytes(N)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
  stream = _stream()
  _lib.check(lib.gsr_compact_offsets(_p(keep), N, _p(offsets), _p(total), _p(ws), ws_bytes, stream),
             "gsr_compact_offsets")
  kept = int(total.item())                  # the one host sync: the outputs' size is data dependent
  outs = [torch.empty((kept + n_tail,) + tuple(col.shape[1:]), dtype=col.dtype, device=dev) for col, _ in cols]
  if kept + n_tail == 0 or not cols:
    return outs
  arr = (_lib.GsrColumnC * len(cols))()
  for i, ((col, tail), out) in enumerate(zip(cols, outs)):
    width = 1
    for d in col.shape[1:]:
      width *= int(d)
    arr[i] = _lib.GsrColumnC(col.data_ptr() if N else None, out.data_ptr(),
                             tail.data_ptr() if (tail is not None and n_tail) else None, max(width, 1))
  _lib.check(lib.gsr_compact_columns(_p(keep), N, _p(offsets), kept, n_tail, arr, len(cols), stream),
             "gsr_compact_columns")
  return outs


def point_state_add(state, points, split_alpha: float = 0.01, prune_alpha: float = 0.1, visible_sum=None):
  """``PointState.add_rendering`` (splat_trainer/controller/point_state.py:34-50) for one camera in one launch, in place
  on the state's device tensors.  ``points``: RenderedPoints (idx, screen_scale (M,2) or (M,), visibility, split_score,
  prune_cost).  ``visible_sum``: optional second (N,) accumulator that receives ``+= visibility`` on the same rows (the
  scene's ``visible`` statistic, mlp_scene.py:244)."""
  return point_state_update(state, points.idx, screen_scale=points.screen_scale, visibility=points.visibility,
                            split_score=points.split_score, prune_cost=points.prune_cost, split_alpha=split_alpha,
                            prune_alpha=prune_alpha, visible_sum=visible_sum)


def point_state_update(state, idx, screen_scale=None, visibility=None, split_score=None, prune_cost=None,
                       split_alpha: float = 0.01, prune_alpha: float = 0.1, visible_sum=None):
  """The fused update with every input group optional (``None`` leaves that part of the state alone); ``idx = None``:
  the rows are the points 0..M-1.  The data-parallel exchange uses it to replay only the two order-dependent EMAs per
  camera (distributed.exchange_point_stats)."""
  lib = _lib.load()
  given = [t for t in (screen_scale, visibility, split_score, prune_cost) if t is not None]
  if not given:
    return state
  M = int(idx.shape[0]) if idx is not None else int(given[0].shape[0])
  if M == 0:
    return state
  # (float32 contiguous inputs -- what the renderer delivers -- pass through untouched: this runs once per camera)
  f32 = lambda t: t if (t is None or (t.dtype is torch.float32 and t.is_contiguous())) else \
      t.detach().to(torch.float32).contiguous()
  scale = f32(screen_scale)
  cols = 2 if (scale is not None and scale.dim() == 2) else 1
  for t in (state.prune_cost, state.split_score, state.max_scale_px, state.visibility):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
      raise _lib.GsplatHipError("point_state_update needs contiguous float32 CUDA state tensors (no CPU fallback)")
  if state.points_in_view.dtype != torch.int16:
    raise ValueError("points_in_view must be int16 (point_state.py:27)")
  if visible_sum is not None and not (visible_sum.is_cuda and visible_sum.dtype == torch.float32 and
                                      visible_sum.is_contiguous() and visible_sum.shape == state.visibility.shape):
    raise ValueError("visible_sum must be a contiguous float32 CUDA tensor shaped like state.visibility")
  _lib.check(lib.gsr_point_state_add(_p(idx if idx.is_contiguous() else idx.contiguous()) if idx is not None else None, _p(scale), cols,
                                     _p(f32(visibility)), _p(f32(split_score)), _p(f32(prune_cost)), M,
                                     float(split_alpha), float(prune_alpha), _p(state.prune_cost), _p(state.split_score),
                                     _p(state.max_scale_px), _p(state.points_in_view), _p(state.visibility),
                                     _p(visible_sum), _stream()),
             "gsr_point_state_add")
  return state


# ------------------------------------------------------------------------------ data-parallel exchange blocks
def dp_block_floats(num_points: int) -> int:
  """Floats of one camera's exchange block (include/gsplat_hip.h: GSR_DP_BLOCK_FLOATS)."""
  return 6 * int(num_points) + 3


def dp_pack(block: torch.Tensor, num_points: int, idx: torch.Tensor, d_colour: torch.Tensor, split_score: torch.Tensor,
            prune_cost: torch.Tensor, screen_scale: torch.Tensor, camera_pos: torch.Tensor,
            visibility: Optional[torch.Tensor] = None, sums: Optional[torch.Tensor] = None) -> torch.Tensor:
  """Fills one camera's fixed-size exchange block (layout in gsplat_hip.h) from the rows ``idx`` the camera saw: colour
  gradient (0 elsewhere), camera position, split_score / prune_cost (NaN elsewhere), larger screen-space sigma (0
  elsewhere).  One launch (two when the camera did not see every point).  ``visibility`` + ``sums`` (2 N floats): the
  camera's visibility and its saw-the-point count are added to ``sums[:N]`` / ``sums[N:]`` in the same launch.  On CPU
  tensors -- the gloo tests of the exchange logic -- the same is done with torch ops."""
  N = int(num_points)
  M = int(idx.shape[0])
  if block.numel() != dp_block_floats(N) or block.dtype != torch.float32 or not block.is_contiguous():
    raise ValueError("dp_pack: block must be a contiguous float32 tensor of 6 N + 3 elements")
  f32 = lambda t: t.detach().to(torch.float32).contiguous()
  scale = f32(screen_scale)
  if not block.is_cuda:
    block[:3 * N] = 0
    block[3 * N + 3:5 * N + 3] = float("nan")
    block[5 * N + 3:] = 0
    block[:3 * N].view(N, 3).index_copy_(0, idx, f32(d_colour))
    block[3 * N:3 * N + 3] = f32(camera_pos)
    block[3 * N + 3:4 * N + 3].index_copy_(0, idx, f32(split_score))
    block[4 * N + 3:5 * N + 3].index_copy_(0, idx, f32(prune_cost))
    block[5 * N + 3:].index_copy_(0, idx, scale.max(1).values if scale.dim() == 2 else scale)
    if visibility is not None:
      sums[:N].index_add_(0, idx, f32(visibility))
      sums[N:].index_add_(0, idx, (visibility > 0).to(torch.float32))
    return block
  lib = _lib.load()
  cols = 2 if scale.dim() == 2 else 1
  _lib.check(lib.gsr_dp_pack(_p(idx.contiguous()) if M < N else None, _p(f32(d_colour)), _p(f32(split_score)),
                             _p(f32(prune_cost)), _p(scale), cols, _p(f32(camera_pos)), M, N, _p(block),
                             _p(f32(visibility)) if visibility is not None else None,
                             _p(sums[:N]) if visibility is not None else None,
                             _p(sums[N:]) if visibility is not None else None, _stream()), "gsr_dp_pack")
  return block


def dp_replay(state, blocks: torch.Tensor, slots: torch.Tensor, num_points: int, split_alpha: float = 0.01,
              prune_alpha: float = 0.1, sums: Optional[torch.Tensor] = None):
  """Applies, for all cameras in camera order (``slots[c]`` = row of ``blocks`` holding camera c), the two
  order-dependent EMAs of PointState.add_rendering to the rows each camera saw and folds the screen-scale maximum --
  one pass over the points.  ``sums`` (2 N floats, all-reduced over the ranks): visibility and saw-the-point counts of
  the batch, added to ``state.visibility`` / ``state.points_in_view`` in the same pass."""
  N = int(num_points)
  if blocks.dim() != 2 or blocks.shape[1] < dp_block_floats(N) or not blocks.is_contiguous():
    raise ValueError("dp_replay: blocks must be (slots, >= 6 N + 3) contiguous")
  if not blocks.is_cuda:
    from .controller_math import exp_lerp
    for s in slots.tolist():
      b = blocks[s]
      split, prune = b[3 * N + 3:4 * N + 3], b[4 * N + 3:5 * N + 3]
      seen = ~torch.isnan(split)
      state.split_score[seen] = exp_lerp(split_alpha, state.split_score[seen], split[seen])
      state.prune_cost[seen] = exp_lerp(prune_alpha, state.prune_cost[seen], prune[seen])
      torch.maximum(state.max_scale_px, b[5 * N + 3:6 * N + 3], out=state.max_scale_px)
    if sums is not None:
      state.visibility += sums[:N]
      state.points_in_view += sums[N:].to(state.points_in_view.dtype)
    return state
  lib = _lib.load()
  _lib.check(lib.gsr_dp_replay(_p(blocks), int(blocks.shape[1]), _p(slots), int(slots.shape[0]), N, float(split_alpha),
                               float(prune_alpha), _p(state.split_score), _p(state.prune_cost), _p(state.max_scale_px),
                               _p(sums[:N]) if sums is not None else None, _p(sums[N:]) if sums is not None else None,
                               _p(state.visibility) if sums is not None else None,
                               _p(state.points_in_view) if sums is not None else None, _stream()), "gsr_dp_replay")
  return state


# ---------------------------------------------------------------------------------- sharded exchange (round 4)
def dp_slice_len(num_points: int, num_ranks: int) -> int:
  """L = ceil(N / G): rank r owns the points [r L, (r + 1) L) of the order-dependent controller state."""
  return (int(num_points) + int(num_ranks) - 1) // int(num_ranks)


def dp_pack_sharded(factors: torch.Tensor, scores: torch.Tensor, scale_max: torch.Tensor, num_points: int, slot: int,
                    idx: torch.Tensor, d_colour: torch.Tensor, split_score: torch.Tensor, prune_cost: torch.Tensor,
                    screen_scale: torch.Tensor, camera_pos: torch.Tensor, visibility: Optional[torch.Tensor] = None,
                    sums: Optional[torch.Tensor] = None):
  """One camera of this rank into the sharded exchange's send buffers (layouts in gsplat_hip.h: gsr_dp_pack_sharded):
  ``factors`` (3 N + 3,) the camera's all-gather block (colour gradient rows, 0 where unseen, + camera position);
  ``scores`` (G, slots_per_rank, 2, L) the all-to-all send buffer, this camera's split_score / prune_cost (NaN where
  unseen) in slot ``slot``; ``scale_max`` (N,) running maximum of the larger screen-space sigma over this rank's cameras;
  ``visibility`` + ``sums`` as in dp_pack.  CPU tensors (the gloo tests): the same with torch ops."""
  N, M = int(num_points), int(idx.shape[0])
  G, cpr, _, L = scores.shape
  if (factors is not None and factors.numel() != 3 * N + 3) or L != dp_slice_len(N, G) or scale_max.numel() != N:
    raise ValueError("dp_pack_sharded: buffer shapes do not match (N, ranks)")
  f32 = lambda t: t.detach().to(torch.float32).contiguous()
  scale = f32(screen_scale)
  if not scores.is_cuda:
    if factors is not None:                 # (None: the block was packed earlier, dp_pack_factors_rows)
      factors[:3 * N] = 0
      factors[:3 * N].view(N, 3).index_copy_(0, idx, f32(d_colour))
      factors[3 * N:] = f32(camera_pos)
    dense = torch.full((2, G * L), float("nan"))
    dense[0, idx], dense[1, idx] = f32(split_score), f32(prune_cost)
    scores[:, slot] = dense.view(2, G, L).permute(1, 0, 2)
    sc = scale.max(1).values if scale.dim() == 2 else scale
    scale_max[idx] = torch.maximum(scale_max[idx], sc)
    if visibility is not None:
      sums[:N].index_add_(0, idx, f32(visibility))
      sums[N:].index_add_(0, idx, (visibility > 0).to(torch.float32))
    return
  cols = 2 if scale.dim() == 2 else 1
  _lib.check(_lib.load().gsr_dp_pack_sharded(
      _p(idx.contiguous()) if M < N else None, _p(f32(d_colour)) if factors is not None else None, _p(f32(split_score)),
      _p(f32(prune_cost)), _p(scale), cols, _p(f32(camera_pos)), M, N, int(G), int(cpr), int(slot), _p(factors), _p(scores),
      _p(scale_max),
      _p(f32(visibility)) if visibility is not None else None, _p(sums[:N]) if visibility is not None else None,
      _p(sums[N:]) if visibility is not None else None, _stream()), "gsr_dp_pack_sharded")


def dp_pack_factors_rows(factors: torch.Tensor, num_points: int, idx: torch.Tensor, grad_rows: torch.Tensor,
                         camera_pos: torch.Tensor):
  """A camera's (3 N + 3)-float factor block straight from its packed gradient rows (M, 16) -- columns 8..10 are the
  colour gradient -- i.e. before the backward sweep has copied them out (the early factor gather of the exchange)."""
  N, M = int(num_points), int(idx.shape[0])
  if factors.numel() != 3 * N + 3:
    raise ValueError("dp_pack_factors_rows: factors must hold 3 N + 3 floats")
  if not factors.is_cuda:
    factors[:3 * N] = 0
    factors[:3 * N].view(N, 3).index_copy_(0, idx, grad_rows[:, 8:11].to(torch.float32))
    factors[3 * N:] = camera_pos.to(torch.float32)
    return
  _lib.check(_lib.load().gsr_dp_pack_factors_rows(_p(idx.contiguous()), _p(grad_rows), _p(camera_pos.detach().to(torch.float32).contiguous()),
                                                  M, N, _p(factors), _stream()), "gsr_dp_pack_factors_rows")


def dp_replay_slice(state, recv: torch.Tensor, rank: int, num_cameras: int, num_points: int, slice_out: torch.Tensor,
                    split_alpha: float = 0.01, prune_alpha: float = 0.1):
  """The two order-dependent EMAs of PointState.add_rendering (point_state.py:49-50) of ALL cameras of the batch, in
  camera order, on THIS rank's slice of the points: ``recv`` (G, slots_per_rank, 2, L) is what the all-to-all delivered
  (camera c came from rank c % G, slot c // G); the slice's new (split_score, prune_cost) go to ``slice_out`` (2, L),
  the all-gather send buffer.  ``state`` itself is not modified (dp_finish writes the gathered slices back)."""
  G, cpr, _, L = recv.shape
  N = int(num_points)
  lo = rank * L
  count = max(0, min(L, N - lo))
  if not recv.is_cuda:
    from .controller_math import exp_lerp
    s, p = state.split_score[lo:lo + count].clone(), state.prune_cost[lo:lo + count].clone()
    for c in range(num_cameras):
      cell = recv[c % G, c // G]
      seen = ~torch.isnan(cell[0, :count])
      s[seen] = exp_lerp(split_alpha, s[seen], cell[0, :count][seen])
      p[seen] = exp_lerp(prune_alpha, p[seen], cell[1, :count][seen])
    slice_out[0, :count], slice_out[1, :count] = s, p
    return
  _lib.check(_lib.load().gsr_dp_replay_slice(_p(recv), int(G), int(cpr), N, int(rank), int(num_cameras), float(split_alpha),
                                             float(prune_alpha), _p(state.split_score), _p(state.prune_cost),
                                             _p(slice_out), _stream()), "gsr_dp_replay_slice")


def dp_finish(state, gathered: torch.Tensor, num_points: int, scale_max: Optional[torch.Tensor] = None,
              sums: Optional[torch.Tensor] = None):
  """Writes the all-gathered state slices (G, 2, L) back into ``state.split_score`` / ``state.prune_cost`` and folds the
  MAX-all-reduced screen scale and the SUM-all-reduced visibility / in-view count into the state."""
  G, _, L = gathered.shape
  N = int(num_points)
  if not gathered.is_cuda:
    flat = gathered.permute(1, 0, 2).reshape(2, G * L)
    state.split_score.copy_(flat[0, :N])
    state.prune_cost.copy_(flat[1, :N])
    if scale_max is not None:
      torch.maximum(state.max_scale_px, scale_max, out=state.max_scale_px)
    if sums is not None:
      state.visibility += sums[:N]
      state.points_in_view += sums[N:].to(state.points_in_view.dtype)
    return state
  _lib.check(_lib.load().gsr_dp_finish(_p(gathered), int(G), N, _p(state.split_score), _p(state.prune_cost),
                                       _p(scale_max), _p(state.max_scale_px) if scale_max is not None else None,
                                       _p(sums[:N]) if sums is not None else None, _p(sums[N:]) if sums is not None else None,
                                       _p(state.visibility) if sums is not None else None,
                                       _p(state.points_in_view) if sums is not None else None, _stream()), "gsr_dp_finish")
  return state
