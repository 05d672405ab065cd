"""Host side of the rasterizer path: the three calls splat-trainer makes at the render boundary.

    project_to_image(gaussians, camera_params, config) -> (gaussians2d, depth, indexes)
    render_projected(indexes, gaussians2d, features, depth, camera_params, config, **options) -> Rendering
    render_gaussians(gaussians, camera_params, config, use_sh=..., ...) -> Rendering

Same names, argument meaning and outputs as the taichi_splatting calls at
splat_trainer/scene/mlp_scene.py:375-378,415-419 and splat_trainer/scripts/test_split.py:30.
Everything numerical happens in hand-written HIP kernels reached through the C ABI
(include/gsplat_hip.h); PyTorch supplies device memory, the current stream and autograd plumbing.
There is no CPU path: CPU tensors or a missing libgsplat_hip.so raise.

Re-entrancy (the reference serialises taichi launches through TaichiQueue, train_scan.py:237,
while the viewer thread renders concurrently, splatview.py:230-254): every call owns its
workspaces and only enqueues on ``torch.cuda.current_stream()``; the library keeps no global
state, so concurrent callers on different streams/threads are safe and no run_sync is needed.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import threading
from typing import Optional, Tuple

import torch

from . import _lib
from .data_types import CameraParams, Gaussians3D, RasterConfig, RenderedPoints, Rendering
from . import sh as _sh
from .sh import evaluate_sh_at

ROW_FLOATS = 16          # packed per-splat row (64 bytes; include/gsplat_hip.h GSR_ROW_FLOATS)
PARTIAL_FLOATS = 12


class KernelTimer:
  """HIP-event timing of individual kernels on the stream they are launched on (bench.py's roofline leg).
  Off by default (``renderer.KERNEL_TIMER = None``); recording costs two event records per timed launch."""

  def __init__(self):
    self.events = {}
    self._open = {}
    self._pool = []

  def reserve(self, n: int):
    """Creates ``n`` events up front (the runtime materialises an event at its first record, in pool-sized batches:
    a few hundred microseconds that would otherwise land inside somebody's timed step)."""
    for _ in range(n):
      e = torch.cuda.Event(enable_timing=True)
      e.record(_lib.current_stream())
      self._pool.append(e)

  def _event(self):
    e = self._pool.pop() if self._pool else torch.cuda.Event(enable_timing=True)
    e.record(_lib.current_stream())
    return e

  def begin(self, name: str):
    self._open[name] = self._event()

  def _materialized(self):
    """An event whose native handle exists (pool events were recorded once when they were reserved)."""
    if self._pool:
      return self._pool.pop()
    e = torch.cuda.Event(enable_timing=True)
    e.record(_lib.current_stream())
    return e

  def pair(self, name: str):
    """Two events for a native caller to record around a launch itself (frame driver): returns their raw handles.
    (Not recorded here: on a host-bound frame every torch-side record is a few microseconds of the step.)"""
    a, b = self._materialized(), self._materialized()
    self.events.setdefault(name, []).append((a, b))
    return a.cuda_event, b.cuda_event

  def end(self, name: str):
    self.events.setdefault(name, []).append((self._open.pop(name), self._event()))

  def summary(self) -> dict:
    """name -> (launches, average milliseconds).  Synchronises."""
    torch.cuda.synchronize()
    return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / max(len(v), 1)) for k, v in self.events.items()}

  def reset(self):
    self.events.clear()
    self._open.clear()


KERNEL_TIMER: Optional[KernelTimer] = None


def _ptr(t):
  """ctypes pointer of a tensor's data, or of a raw device address (an int: buffers of the frame arena that are only ever
  handed to kernels are kept as addresses, see _run_frame), or None."""
  if t is None:
    return None
  if isinstance(t, int):
    return t or None
  if t.numel() == 0:
    return None
  return t.data_ptr()          # (the prototypes declare c_void_p: ctypes takes the plain int, no wrapper object per argument)


_stream = _lib.current_stream_ptr


def _require_device(*tensors: torch.Tensor):
  for t in tensors:
    if not t.is_cuda:
      raise _lib.GsplatHipError(
          "the rasterizer path runs only on a HIP device (got a CPU tensor); there is no CPU fallback")
    if t.dtype not in (torch.float32, torch.float16, torch.bfloat16, torch.int64, torch.int32):
      raise TypeError(f"unsupported dtype {t.dtype}; the path computes in float32 (half inputs are widened)")


def _f32c(t: torch.Tensor) -> torch.Tensor:
  if t.dtype is torch.float32 and t.is_contiguous():
    return t.detach()
  return t.detach().to(torch.float32).contiguous()


# ------------------------------------------------------------------------------------------- K1 + K2
def frustum_cull(position: torch.Tensor, camera_params: CameraParams, config: RasterConfig) -> torch.Tensor:
  """K1: ascending int64 indices of the points whose centre lies in the margin-expanded frustum."""
  lib = _lib.load()
  _require_device(position)
  pos = _f32c(position)
  N = pos.shape[0]
  dev = pos.device
  W, H = camera_params.image_size
  T = _f32c(camera_params.T_camera_world)
  proj = _f32c(camera_params.projection)
  indexes = torch.empty(N, dtype=torch.int64, device=dev)
  count = torch.empty(1, dtype=torch.int32, device=dev)
  ws_bytes = lib.gsr_cull_workspace_bytes(N)
  ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
  _lib.check(lib.gsr_frustum_cull(_ptr(pos), N, _ptr(T), _ptr(proj), int(W), int(H),
                                  float(camera_params.near_plane), float(camera_params.far_plane),
                                  float(config.margin_tiles * config.tile_size), _ptr(indexes), _ptr(count),
                                  _ptr(ws), ws_bytes, _stream()), "gsr_frustum_cull")
  M = int(count.item())          # host sync #1: the index tensor's size is data dependent
  return indexes[:M]


# ------------------------------------------------------------------------------- data-dependent sizes
_TLS = threading.local()
# GSPLAT_HIP_NO_SPECULATION=1: enqueue nothing before the size it depends on has reached the host (A/B measurements;
# results are bit-identical either way, tests/test_gpu_render.py::test_speculative_emit_and_early_colours_do_not_change_results)
SPECULATE = os.environ.get("GSPLAT_HIP_NO_SPECULATION", "0") != "1"


def _readback_slot(device):
  """(pinned int32[8] landing buffer, event, the buffer as a ctypes array) of this thread for ``device``; the event has
  been recorded once so that its native handle exists (native callers record it themselves)."""
  pool = _TLS.__dict__.setdefault("readback", {})
  key = device.index
  if key not in pool:
    event = torch.cuda.Event()
    event.record(_lib.current_stream())
    host = torch.empty(8, dtype=torch.int32).pin_memory()
    pool[key] = (host, event, (C.c_int32 * 8).from_address(host.data_ptr()))    # + the same words as a ctypes array
  return pool[key]


# GSPLAT_HIP_SIDE_STREAM=1: the frame driver puts the depth sort on a second stream next to the fused K2 + K3 sweep.
# Measured (same box, median ms/step, off -> on): c1 0.511 -> 0.583, c2 0.958 -> 0.979, c3 2.705 -> 2.682: the cross-stream
# hand-off costs more than the nine small sort launches hide except at millions of splats; off by default.
SIDE_STREAM = os.environ.get("GSPLAT_HIP_SIDE_STREAM", "0") == "1"


def _side_stream(device):
  """(stream, fork event, join event) of this thread for ``device``: the frame driver puts the depth sort on the side
  stream while the fused projection + colour sweep runs on the current one (csrc/frame.hip).  Created once; the events
  have been recorded once so that their native handles exist."""
  pool = _TLS.__dict__.setdefault("side", {})
  key = device.index
  if key not in pool:
    stream = torch.cuda.Stream(device=device)
    fork, join = torch.cuda.Event(), torch.cuda.Event()
    fork.record(_lib.current_stream())
    join.record(_lib.current_stream())
    pool[key] = (stream, fork, join)
  return pool[key]


def _start_readback(words: torch.Tensor):
  """Begins the device->host copy of a few int32 words on the current stream and returns ``wait() -> list``.
  The two sizes the path cannot know in advance (visible splats, tile overlaps) come back this way: whatever is
  enqueued between this call and ``wait()`` runs while the host blocks on the COPY's event, not on the stream, so the
  GPU does not idle through the host's round trip.  One pinned landing buffer + event per thread and device."""
  host, event, _ = _readback_slot(words.device)
  n = words.numel()
  host[:n].copy_(words, non_blocking=True)
  event.record(_lib.current_stream())

  def wait():
    event.synchronize()
    return host[:n].tolist()
  return wait


class GradOut:
  """Optional fused gradient accumulation, and the ONE owner of the "who initialises which buffer" question.

  When given to project_to_image / evaluate_sh_at / render_gaussians, the backward kernels add ("+=") their rows straight
  into these N-sized buffers -- typically the parameters' ``.grad`` tensors, which is where the reference accumulates over
  the cameras of a batch (trainer.py:500-514, mlp_scene.py:155-161) -- and autograd receives no gradient for those inputs.
  This removes the zero-fill + dense add of N-sized temporaries per camera.  Default (None): plain autograd.

  A batch may declare buffers UNINITIALISED (``begin_batch``, or the two constructor flags): their contents are not worth
  keeping, so the first backward node that reaches one may overwrite every row of it (zeros where its camera saw nothing)
  instead of the caller zero-filling it and the node reading it back -- ``feature`` is by far the largest buffer (3K of the
  3K + 11 floats per splat).  Which node that is depends on autograd's order (the SH node of the three-call form runs
  BEFORE the projection's, and only ADDS its position term), so the nodes do not test flags themselves; they ask:

    ``claim_overwrite(node, names)``   "I write every row of these buffers."  True: all of them were uninitialised and are
                                       now this node's to overwrite.  False: at least one holds earlier terms -- the ones
                                       still uninitialised are zero-filled here and the node accumulates into all.
    ``claim_accumulate(node, names)``  "I add to these buffers."  Those still uninitialised are zero-filled first.
    ``finish_batch()``                 Whoever consumes the gradients (optimizer step, all-reduce) calls it: buffers no node
                                       reached (a rank without a camera) are zero-filled; returns their names.

  ``debug=True`` (or GSPLAT_HIP_DEBUG_GRADOUT=1) records every claim in ``log`` -- (node, action, buffers) in backward
  order -- and raises on a protocol violation: a buffer overwritten after it had been written in the same batch, or a
  buffer consumed (``finish_batch``) that a node had claimed for overwrite but some other node then also overwrote.
  ``geometry_uninitialized`` / ``feature_uninitialized`` remain as properties over the same state (older callers set them)."""

  GEOMETRY = ("position", "log_scaling", "rotation", "alpha_logit")
  NAMES = GEOMETRY + ("feature",)

  def __init__(self, position=None, log_scaling=None, rotation=None, alpha_logit=None, feature=None,
               feature_uninitialized: bool = False, geometry_uninitialized: bool = False, debug: Optional[bool] = None):
    self.position, self.log_scaling, self.rotation = position, log_scaling, rotation
    self.alpha_logit, self.feature = alpha_logit, feature
    self.debug = (os.environ.get("GSPLAT_HIP_DEBUG_GRADOUT", "0") == "1") if debug is None else bool(debug)
    self._fresh = set()                   # buffers declared uninitialised that no node has written yet
    self._overwritten = {}                # debug: buffer -> node that overwrote it in this batch
    self.log = []
    self.begin_batch(geometry=geometry_uninitialized, feature=feature_uninitialized)

  # ---- declaration / consumption (the caller's side)
  def begin_batch(self, geometry: bool = True, feature: bool = True):
    """Declares the four geometry buffers and / or ``feature`` uninitialised for the batch that starts now."""
    self._fresh = set(self.GEOMETRY if geometry else ()) | ({"feature"} if feature else set())
    self._overwritten = {}
    self.log = []
    return self

  def finish_batch(self):
    """Zero-fills every buffer no backward node reached; returns their names (empty in the usual case)."""
    left = sorted(self._fresh)
    for n in left:
      t = getattr(self, n)
      if t is not None:
        t.zero_()
    if left:
      self._note("finish_batch", "zero-fill", left)
    self._fresh.clear()
    return left

  # ---- claims (the backward nodes' side)
  def claim_overwrite(self, node: str, names) -> bool:
    names = tuple(names)
    fresh = [n for n in names if n in self._fresh]
    if len(fresh) == len(names):
      for n in names:
        self._fresh.discard(n)
        if self.debug:
          if n in self._overwritten:
            raise _lib.GsplatHipError(f"GradOut: {node} overwrites {n}, which {self._overwritten[n]} already wrote in this batch")
          self._overwritten[n] = node
      self._note(node, "overwrite", names)
      return True
    self.claim_accumulate(node, names)
    return False

  def claim_accumulate(self, node: str, names):
    fresh = [n for n in names if n in self._fresh]
    for n in fresh:
      t = getattr(self, n)
      if t is not None:
        t.zero_()
      self._fresh.discard(n)
    if fresh:
      self._note(node, "zero-fill", fresh)
    self._note(node, "accumulate", tuple(names))
    return fresh                          # (a node that was handed a buffer other than this object's own zero-fills that one itself)

  def _note(self, node, action, names):
    if self.debug:
      self.log.append((node, action, tuple(names)))

  # ---- the two flags older callers use, over the same state
  @property
  def geometry_uninitialized(self) -> bool:
    return any(n in self._fresh for n in self.GEOMETRY)

  @geometry_uninitialized.setter
  def geometry_uninitialized(self, value: bool):
    (self._fresh.update if value else self._fresh.difference_update)(self.GEOMETRY)

  @property
  def feature_uninitialized(self) -> bool:
    return "feature" in self._fresh

  @feature_uninitialized.setter
  def feature_uninitialized(self, value: bool):
    (self._fresh.add if value else self._fresh.discard)("feature")

  def take_geometry_uninitialized(self) -> bool:
    """= claim_overwrite over the four geometry buffers (kept for callers of the round-3 protocol)."""
    return self.claim_overwrite("caller", self.GEOMETRY)

  def ensure_geometry_initialized(self):
    """= claim_accumulate over the four geometry buffers."""
    self.claim_accumulate("caller", self.GEOMETRY)

  def _check(self, name, like):
    t = getattr(self, name)
    if t is None or t.shape != like.shape or t.dtype != torch.float32 or not t.is_contiguous() or t.device != like.device:
      raise ValueError(f"GradOut.{name} must be a contiguous float32 tensor shaped like the parameter")
    return t


class _ProjectFn(torch.autograd.Function):
  """K1 + K2 as one autograd node.  The projection kernel is enqueued right behind the cull with the visible
  count still on the device, so the GPU keeps working while the host reads the count back (the one unavoidable
  sync: the size of ``indexes`` is data dependent)."""

  @staticmethod
  def forward(ctx, position, log_scaling, rotation, alpha_logit, T, proj, cull_args, params, grad_out, prefetch):
    lib = _lib.load()
    pos, ls, rot, al = _f32c(position), _f32c(log_scaling), _f32c(rotation), _f32c(alpha_logit)
    N, dev = pos.shape[0], pos.device
    W, H, near, far, margin = cull_args
    stream = _stream()
    indexes_full = torch.empty(N, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    ws_bytes = lib.gsr_cull_workspace_bytes(N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    g2d_full = torch.empty(N, 6, dtype=torch.float32, device=dev)
    depth_full = torch.empty(N, 1, dtype=torch.float32, device=dev)
    keys_full = _u32(N, dev) if prefetch is not None else None     # the depth sort's keys, when the caller rasterizes next
    key_range = _depth_key_range(near, far)
    _lib.check(lib.gsr_frustum_cull(_ptr(pos), N, _ptr(T), _ptr(proj), W, H, near, far, margin, _ptr(indexes_full),
                                    _ptr(count), _ptr(ws), ws_bytes, stream), "gsr_frustum_cull")
    _lib.check(lib.gsr_project_forward(_ptr(pos), _ptr(ls), _ptr(rot), _ptr(al), _ptr(indexes_full), N, _ptr(T),
                                       _ptr(proj), C.byref(params), _ptr(g2d_full), _ptr(depth_full), _ptr(count),
                                       _ptr(keys_full), key_range[0], key_range[1], stream), "gsr_project_forward")
    wait = _start_readback(count)  # host sync #1 (K2 is already running)
    sh_job = prefetch.pop("sh", None) if prefetch is not None else None
    if sh_job is not None and N > 0:
      # the colours only need `indexes`: K3 goes in behind K2 with the count still on the device and keeps the GPU
      # busy for the ~40 us the host needs to read M back and get to its next launch
      sh_out = _sh.launch_forward_counted(*sh_job[:3], indexes_full, count, sh_job[3])
    M = int(wait()[0])
    if sh_job is not None and N > 0:
      prefetch["sh_out"] = (sh_out[0][:M], sh_out[1][:M] if sh_out[1] is not None else None)
    indexes, g2d, depth = indexes_full[:M], g2d_full[:M], depth_full[:M]
    if prefetch is not None and M > 0:
      # the caller will rasterize next: the depth sort needs only `depth`, so it is enqueued now and runs while the host
      # works its way to render_projected (the GPU would otherwise idle behind the sync)
      prefetch["order"] = _launch_depth_order(depth.reshape(-1), M, key_range, keys=keys_full[:M])
    ctx.save_for_backward(pos, ls, rot, al, indexes, T, proj)
    ctx.set_materialize_grads(False)       # an unused output (depth, usually) arrives as None, not as a zero-filled tensor
    ctx.params = params
    ctx.grad_out = grad_out
    ctx.in_dtypes = (position.dtype, log_scaling.dtype, rotation.dtype, alpha_logit.dtype)
    ctx.mark_non_differentiable(indexes)
    return g2d, depth, indexes

  @staticmethod
  def backward(ctx, d_g2d, d_depth, _d_indexes):
    lib = _lib.load()
    pos, ls, rot, al, indexes, T, proj = ctx.saved_tensors
    M, N = indexes.shape[0], pos.shape[0]
    go = ctx.grad_out
    if go is not None:
      d_pos, d_ls = go._check("position", pos), go._check("log_scaling", ls)
      d_rot, d_al = go._check("rotation", rot), go._check("alpha_logit", al)
      go.claim_accumulate("project_to_image.backward", go.GEOMETRY)      # rows of `indexes` are added to; the rest must be defined
    else:
      live = M > 0 and (d_g2d is not None or d_depth is not None)
      alloc = torch.empty_like if (M == N and live) else torch.zeros_like   # every row is written when nothing was culled
      d_pos, d_ls, d_rot, d_al = alloc(pos), alloc(ls), alloc(rot), alloc(al)
    if M > 0 and (d_g2d is not None or d_depth is not None):
      dg = _f32c(d_g2d) if d_g2d is not None else torch.zeros(M, 6, dtype=torch.float32, device=pos.device)
      dd = _f32c(d_depth) if d_depth is not None else None
      _lib.check(lib.gsr_project_backward(_ptr(pos), _ptr(ls), _ptr(rot), _ptr(al), _ptr(indexes), M, _ptr(T),
                                          _ptr(proj), C.byref(ctx.params), _ptr(dg), _ptr(dd), _ptr(d_pos),
                                          _ptr(d_ls), _ptr(d_rot), _ptr(d_al), 1 if go is not None else 0,
                                          _stream()), "gsr_project_backward")
    if go is not None:
      return None, None, None, None, None, None, None, None, None, None
    dt = ctx.in_dtypes
    return d_pos.to(dt[0]), d_ls.to(dt[1]), d_rot.to(dt[2]), d_al.to(dt[3]), None, None, None, None, None, None


def project_to_image(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig,
                     grad_out: Optional[GradOut] = None, prefetch: Optional[dict] = None
                     ) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
  """K1 cull + K2 projection.  Returns ``gaussians2d (M,6) = [u v A B C opacity]``, ``depth (M,1)``,
  ``indexes (M,) int64`` (ascending).  Differentiable wrt position / log_scaling / rotation /
  alpha_logit; gradients land in N-sized tensors, zero outside ``indexes``.
  ``prefetch={}``: the depth sort render_projected needs is enqueued right away (it only depends on ``depth``) and handed
  back as ``prefetch["depth_order"]`` for ``render_projected(..., _depth_order=prefetch["depth_order"])``."""
  _require_device(gaussians.position, gaussians.log_scaling, gaussians.rotation, gaussians.alpha_logit)
  T = _f32c(camera_params.T_camera_world)
  proj = _f32c(camera_params.projection)
  params = _lib.raster_params(config)
  W, H = camera_params.image_size
  cull_args = (int(W), int(H), float(camera_params.near_plane), float(camera_params.far_plane),
               float(config.margin_tiles * config.tile_size))
  g2d, depth, indexes = _ProjectFn.apply(gaussians.position, gaussians.log_scaling, gaussians.rotation,
                                         gaussians.alpha_logit, T, proj, cull_args, params, grad_out, prefetch)
  if prefetch is not None and "order" in prefetch:
    prefetch["depth_order"] = (prefetch.pop("order"), depth, depth._version)   # pass as render_projected(_depth_order=)
  return g2d, depth, indexes


# ------------------------------------------------------------------------------------------- K4..K7
class _RasterState:
  """Per-frame buffers shared by forward and backward (owned by the autograd node / the Rendering)."""
  __slots__ = ("M", "O", "C", "W", "H", "params", "rows", "order", "count", "offsets", "sorted_splat",
               "sorted_inst", "tile_range", "vis_partial", "pair_vis", "final_T", "last", "median", "visibility",
               "prune_cost", "split_score", "screen_scale", "want_median", "compute_visibility", "needs_grad",
               "segments", "segment_buffers", "seg_pairs", "seg_min", "image", "key_range", "vis_ready", "vis_capacity",
               "near_far")

  def materialize_visibility(self) -> torch.Tensor:
    """points.visibility = per-splat sum of the forward pass's per-pair partials.  A frame that is back-propagated gets
    it from the backward pass's own reduction (one column of the packed gradient rows, same sums in the same order:
    same bits); whoever reads it EARLIER -- a regularizer on ``points.visible`` before ``loss.backward()``,
    mlp_scene.py:268-288 -- triggers the stand-alone reduction here, once."""
    if not self.vis_ready:
      _lib.check(_lib.load().gsr_reduce_visibility(_ptr(self.vis_partial), _ptr(self.offsets), _ptr(self.count),
                                                   _ptr(self.order), self.M, _ptr(self.visibility), self.vis_capacity,
                                                   None, _stream()), "gsr_reduce_visibility")
      self.vis_ready = True
    return self.visibility


def _u32(n: int, device) -> torch.Tensor:
  return torch.empty(max(n, 1), dtype=torch.int32, device=device)   # raw storage for uint32 arrays


_KEY_RANGES = {}


def _depth_key_range(near: float, far: float):
  """(bias, max_key, significant bits) of the depth sort keys of a camera: keys are taken relative to the near plane
  (binning.hip: gsr_depth_key_range), so the stable radix sort only has to cover bits(far) - bits(near)."""
  key = (float(near), float(far))
  hit = _KEY_RANGES.get(key)
  if hit is None:
    bias, top = C.c_uint32(0), C.c_uint32(0)
    _lib.check(_lib.load().gsr_depth_key_range(key[0], key[1], C.byref(bias), C.byref(top)), "gsr_depth_key_range")
    hit = _KEY_RANGES[key] = (bias.value, top.value, max(1, int(top.value).bit_length()))
    if len(_KEY_RANGES) > 4096:
      _KEY_RANGES.clear()
  return hit


def _launch_depth_order(depth: Optional[torch.Tensor], M: int, key_range, keys: Optional[torch.Tensor] = None) -> torch.Tensor:
  """depth keys + stable radix sort of the M splats over the keys' significant bits (ties keep ascending index); returns
  order (M,) int32.  ``key_range`` = _depth_key_range(near, far); ``keys``: the keys when K2 has already written them
  (consumed as sort scratch)."""
  lib = _lib.load()
  dev = keys.device if keys is not None else depth.device
  stream = _stream()
  bias, max_key, key_bits = key_range
  keys_a = keys if keys is not None else _u32(M, dev)
  keys_b, vals_a, vals_b = _u32(M, dev), _u32(M, dev), _u32(M, dev)
  sort_bytes = lib.gsr_sort_workspace_bytes(M)
  sort_ws = torch.empty(sort_bytes, dtype=torch.uint8, device=dev)
  if keys is None:
    _lib.check(lib.gsr_depth_keys(_ptr(depth), M, bias, max_key, _ptr(keys_a), stream), "gsr_depth_keys")
  where = _lib.check(lib.gsr_sort_pairs_u32(_ptr(keys_a), _ptr(vals_a), _ptr(keys_b), _ptr(vals_b), M, 1, 0, key_bits,
                                            _ptr(sort_ws), sort_bytes, None, stream), "gsr_sort_pairs_u32(depth)")
  return vals_b if where == 1 else vals_a


def _seg_ref(st: "_RasterState"):
  return C.byref(st.segments) if st.segments is not None else None


def _blank_frame(st: _RasterState, dev) -> torch.Tensor:
  """The frame of an empty splat list (M = 0): colour 0, transmittance 1, no per-point outputs."""
  M, C_, W, H = st.M, st.C, st.W, st.H
  st.O = 0
  st.rows = torch.empty(0, ROW_FLOATS, dtype=torch.float32, device=dev)
  st.prune_cost, st.split_score = torch.zeros(2, M, dtype=torch.float32, device=dev).unbind(0)
  st.screen_scale = torch.zeros(0, 2, dtype=torch.float32, device=dev)
  st.final_T = torch.ones(H, W, dtype=torch.float32, device=dev)
  st.last = torch.zeros(H, W, dtype=torch.int32, device=dev)
  st.median = torch.zeros(H, W, dtype=torch.float32, device=dev) if st.want_median else None
  st.visibility = torch.zeros(M, dtype=torch.float32, device=dev)
  st.vis_ready = True
  st.segments, st.segment_buffers = None, None
  return torch.zeros(H, W, C_, dtype=torch.float32, device=dev)


def _composite_backward_rows(st: _RasterState, d_image: Optional[torch.Tensor], dev) -> torch.Tensor:
  """K7 + the per-splat reduction: the packed (M,16) gradient rows of the frame
  (du dv dA dB | dC dop prune split | df0 df1 df2 visibility | 0 0 0 0), zeros when nothing was composited."""
  lib = _lib.load()
  live = st.M > 0 and st.O > 0 and d_image is not None
  if not live:
    return torch.zeros(st.M, ROW_FLOATS, dtype=torch.float32, device=dev)
  if st.vis_partial is None:
    raise _lib.GsplatHipError("backward called on a rendering made without gradient state")
  stream = _stream()
  dimg = _f32c(d_image)
  partial = torch.empty(st.O, PARTIAL_FLOATS, dtype=torch.float32, device=dev)
  timer = KERNEL_TIMER
  if timer is not None:
    timer.begin("composite_backward")
  _lib.check(lib.gsr_composite_backward(_ptr(st.rows), _ptr(st.sorted_splat), _ptr(st.sorted_inst),
                                        _ptr(st.pair_vis), _ptr(st.tile_range), st.W, st.H, st.C,
                                        C.byref(st.params), _ptr(st.final_T), _ptr(st.last), _ptr(dimg),
                                        _ptr(st.image), _ptr(partial), _seg_ref(st), stream),
             "gsr_composite_backward")
  if timer is not None:
    timer.end("composite_backward")
  grows = torch.empty(st.M, ROW_FLOATS, dtype=torch.float32, device=dev)      # every row is written whole
  _lib.check(lib.gsr_reduce_gradients(_ptr(partial), _ptr(st.vis_partial), _ptr(st.offsets), _ptr(st.count),
                                      _ptr(st.order), st.M, _ptr(grows), stream), "gsr_reduce_gradients")
  return grows


def _vis_out(st: _RasterState, live: bool):
  """Where the backward pass leaves points.visibility (None: nothing to deliver, or it has been reduced already)."""
  if live and st.compute_visibility and not st.vis_ready:
    st.vis_ready = True
    return st.visibility
  return None


class _RasterFn(torch.autograd.Function):
  """K4..K7 of the three-call form: the caller's (M,6) / (M,C) / (M,1) tensors are packed into the (M,16) rows the
  kernels read, and the packed gradient rows are taken apart again for the caller's graph."""

  @staticmethod
  def forward(ctx, g2d, feats, depth, st: _RasterState, order):
    g, f, d = _f32c(g2d), _f32c(feats), _f32c(depth).reshape(-1)
    if st.M == 0:
      image = _blank_frame(st, g.device)
    else:
      # the native frame driver in projected mode: pack -> depth sort -> K4 -> K5 -> K6 behind one call
      st.order = order
      frame = _lib.GsrFrameC(None, None, None, None, None, st.M, 1, st.W, st.H, None, None, None, st.near_far[0],
                             st.near_far[1], st.params, 0, int(st.want_median), int(st.compute_visibility),
                             int(bool(st.needs_grad)), st.seg_pairs, st.seg_min, 0, g.data_ptr(), d.data_ptr(),
                             f.data_ptr(), st.C, order.data_ptr() if order is not None else None)
      _, _, _, image = _run_frame(frame, st, g.device, st.M, projected=True)
    ctx.st = st
    ctx.in_dtypes = (g2d.dtype, feats.dtype)      # e.g. fp16 colours from an autocast MLP (mlp_scene.py:362)
    return image

  @staticmethod
  def backward(ctx, d_image):
    lib = _lib.load()
    st: _RasterState = ctx.st
    dev = d_image.device
    live = st.M > 0 and st.O > 0
    alloc = torch.empty if live else torch.zeros                # the unpack sweep writes every row
    d_g2d = alloc(st.M, 6, dtype=torch.float32, device=dev)
    d_feat = alloc(st.M, st.C, dtype=torch.float32, device=dev)
    if live:
      grows = _composite_backward_rows(st, d_image, dev)
      # prune_cost / split_score (/ visibility) are written straight into the tensors the Rendering already holds
      _lib.check(lib.gsr_unpack_grad_rows(_ptr(st.rows), _ptr(grows), st.M, st.C, _ptr(d_g2d), _ptr(d_feat), _ptr(st.prune_cost),
                                          _ptr(st.split_score), _ptr(_vis_out(st, live)), _stream()),
                 "gsr_unpack_grad_rows")
    return d_g2d.to(ctx.in_dtypes[0]), d_feat.to(ctx.in_dtypes[1]), None, None, None


def _arena_view(arena: torch.Tensor, offset: int, shape, dtype=torch.float32) -> torch.Tensor:
  """A contiguous tensor of ``shape`` at byte ``offset`` of the uint8 arena (one as_strided call on a typed alias)."""
  typed = arena.view(dtype)
  size = typed.element_size()
  if len(shape) == 1:
    strides = (1,)
  elif len(shape) == 2:
    strides = (int(shape[1]), 1)
  else:
    strides = (int(shape[1]) * int(shape[2]), int(shape[2]), 1)
  return torch.as_strided(typed, shape, strides, offset // size)


def _pair_capacity(dev_index, N: int):
  """Capacity of the frame's pair buffers: 1.25 x the largest recent pair count of this thread on this device (grows at
  once, decays 1.5 % per frame), in steps of 1/16 of its power of two so that the arenas keep their sizes -- and the
  caching allocator its blocks -- from frame to frame; a first frame starts from 4 pairs per scene row."""
  guesses = _TLS.__dict__.setdefault("overlap_guess", {})
  raw = guesses.get(dev_index, 0) if SPECULATE else 0
  if raw <= 0:
    raw = max(4 * N, 1 << 16)
  step = 1 << max(raw.bit_length() - 5, 0)
  return guesses, raw, min((raw + step - 1) // step * step, 0x7fffffff)


def _run_frame(frame: "_lib.GsrFrameC", st: _RasterState, dev, rows_bound: int, projected: bool):
  """Enqueues one frame through the native driver (csrc/frame.hip) and fills ``st`` with views of its output arena.
  ``rows_bound``: the scene's N (one-call form: the visible count M comes back from the device) or the exact number of
  splats (projected mode).  Returns (out arena, plan, M)."""
  lib = _lib.load()
  guesses, raw_guess, capacity = _pair_capacity(dev.index, rows_bound)
  plan, res = _lib.GsrFramePlanC(), _lib.GsrFrameResultC()
  timer = KERNEL_TIMER
  N, W, H, C_ = rows_bound, st.W, st.H, st.C
  while True:
    frame.pair_capacity = capacity
    _lib.check(lib.gsr_frame_plan(C.byref(frame), C.byref(plan)), "gsr_frame_plan")
    out = torch.empty(plan.out_bytes, dtype=torch.uint8, device=dev)
    work = torch.empty(plan.work_bytes, dtype=torch.uint8, device=dev)
    ev = timer.pair("composite_forward") if timer is not None else (None, None)
    host, ready, host_words = _readback_slot(dev)
    _lib.check(lib.gsr_frame_forward(C.byref(frame), C.byref(plan), C.c_void_p(out.data_ptr()),
                                     C.c_void_p(work.data_ptr()), C.byref(res), C.c_void_p(host.data_ptr()),
                                     C.c_void_p(ready.cuda_event), ev[0], ev[1], _stream()), "gsr_frame_forward")
    del work                                   # scratch: the allocator may hand it on (stream order keeps it safe)
    # the frame's only host wait: on the copy the driver issued right behind the scan, with the emit, the tile sort and
    # the composite already enqueued behind it -- the device works on while the host shapes the tensors and moves on
    ready.synchronize()
    M, O, overflow = host_words[0], host_words[1], host_words[2]
    if overflow or O < 0:      # the guard fires before a 32-bit wrap can go unnoticed (screen-filling splats at 4K)
      raise _lib.GsplatHipError("tile overlap count reached 2^31: the (tile, splat) lists are addressed with 32 bits")
    if O <= capacity:
      break
    if timer is not None:
      timer.events["composite_forward"].pop()            # the frame is run again with exact sizes
    capacity = O
  guesses[dev.index] = min(max(O + O // 4 + 4096, raw_guess - raw_guess // 64), 0x7fffffff)   # grows at once, decays slowly
  if projected:
    M = N
  # Buffers only kernels ever see are kept as device addresses (no tensor objects: ten views a frame cost the host more
  # than the small frames' kernels take); the arena itself stays alive through st.segment_buffers.
  base = out.data_ptr()
  out_f = out.view(torch.float32)
  V = lambda off, shape, strides: torch.as_strided(out_f, shape, strides, off >> 2)
  st.M, st.O = M, O
  st.rows = V(plan.rows, (M, ROW_FLOATS), (ROW_FLOATS, 1))
  st.screen_scale = V(plan.screen_scale, (M, 2), (2, 1))
  if res.order >= 0:
    st.order = base + res.order
  st.count, st.offsets = base + plan.count, base + plan.offsets
  st.sorted_splat, st.sorted_inst = base + res.sorted_splat, base + res.sorted_inst
  st.tile_range = base + plan.tile_range
  keep = st.compute_visibility or st.needs_grad
  st.vis_partial = base + plan.vis_partial if keep else None
  st.pair_vis = base + plan.pair_vis if keep else None
  st.final_T, st.last = V(plan.final_T, (H, W), (W, 1)), base + plan.last
  st.median = V(plan.median, (H, W), (W, 1)) if st.want_median else None
  st.visibility = V(plan.visibility, (M,), (1,))
  st.prune_cost, st.split_score = V(plan.prune_cost, (M,), (1,)), V(plan.split_score, (M,), (1,))
  st.vis_capacity = capacity
  if not st.compute_visibility:
    st.visibility.zero_()
  st.vis_ready = not (st.compute_visibility and st.needs_grad)    # without gradients the driver reduced it already
  st.segments = _lib.GsrSegmentsC.from_buffer_copy(res.segments) if res.has_segments else None
  st.segment_buffers = (out,)
  image = V(plan.image, (H, W, C_), (W * C_, C_, 1))
  st.image = base + plan.image if st.needs_grad else None
  return out, plan, M, image


class _FrameFn(torch.autograd.Function):
  """The one-call form with SH colours as ONE autograd node.  Forward: the native frame driver (csrc/frame.hip) enqueues
  K1 cull -> fused K2 + K3 (one packed 64-byte row per visible splat) -> K4 -> K5 -> K6 behind a single call, with the
  visible count and the pair count left on the device; the host reads both back once, afterwards.  Backward: K7 -> packed
  gradient rows -> one sweep in splat order for the geometry gradients and the per-point outputs -> SH coefficient
  gradient.  Outputs: image, gaussians2d (M,6) and depth (M,1) (views of the rows; differentiable, so regularizers on
  points.opacity / points.depths reach the parameters), indexes."""

  @staticmethod
  def forward(ctx, position, log_scaling, rotation, alpha_logit, feature, T, proj, cam_pos, cull_args, st: _RasterState,
              grad_out, sh_out, want_pos_grad):
    lib = _lib.load()
    pos, ls, rot, al = _f32c(position), _f32c(log_scaling), _f32c(rotation), _f32c(alpha_logit)
    sh, cam = _f32c(feature), _f32c(cam_pos)
    N, K, dev = pos.shape[0], sh.shape[2], pos.device
    W, H, near, far, margin = cull_args
    num_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    if N == 0:
      indexes = torch.empty(0, dtype=torch.int64, device=dev)
      st.M = 0
      image = _blank_frame(st, dev)
      rows = st.rows
      ctx.save_for_backward(pos, ls, rot, al, sh, indexes, T, proj, cam)
      ctx.set_materialize_grads(False)
      ctx.st, ctx.jac, ctx.grad_out, ctx.sh_out = st, None, grad_out, sh_out
      ctx.in_dtypes = (position.dtype, log_scaling.dtype, rotation.dtype, alpha_logit.dtype, feature.dtype)
      ctx.mark_non_differentiable(indexes)
      return image, rows[:, 0:6], rows[:, 10:11], indexes
    frame = _lib.GsrFrameC(pos.data_ptr(), ls.data_ptr(), rot.data_ptr(), al.data_ptr(), sh.data_ptr(), N, K, W, H,
                           T.data_ptr(), proj.data_ptr(), cam.data_ptr(), near, far, st.params,
                           int(bool(want_pos_grad and K > 1)), int(st.want_median), int(st.compute_visibility),
                           int(bool(st.needs_grad)), st.seg_pairs, st.seg_min, 0, None, None, None, 3, None)
    if SIDE_STREAM:
      side, fork, join = _side_stream(dev)
      frame.side_stream, frame.event_fork, frame.event_join = side.cuda_stream, fork.cuda_event, join.cuda_event
    out, plan, M, image = _run_frame(frame, st, dev, N, projected=False)
    indexes = _arena_view(out, plan.indexes, (M,), torch.int64)
    rows = st.rows
    ctx.save_for_backward(pos, ls, rot, al, sh, indexes, T, proj, cam)
    ctx.set_materialize_grads(False)       # unused outputs (gaussians2d / depth, usually) arrive as None
    ctx.st, ctx.jac = st, (_arena_view(out, plan.jacobian, (M, 9)) if plan.jacobian >= 0 else None)
    ctx.grad_out, ctx.sh_out = grad_out, sh_out
    ctx.in_dtypes = (position.dtype, log_scaling.dtype, rotation.dtype, alpha_logit.dtype, feature.dtype)
    ctx.mark_non_differentiable(indexes)
    return image, rows[:, 0:6], rows[:, 10:11], indexes

  @staticmethod
  def backward(ctx, d_image, d_g2d, d_depth, _d_indexes):
    lib = _lib.load()
    pos, ls, rot, al, sh, indexes, T, proj, cam = ctx.saved_tensors
    st: _RasterState = ctx.st
    M, N, K, dev = indexes.shape[0], pos.shape[0], sh.shape[2], pos.device
    go, sh_out = ctx.grad_out, ctx.sh_out
    collector = sh_out if isinstance(sh_out, _sh.ShFactorCollector) else None
    nothing = (None,) * 13
    # Geometry gradients: added to caller-owned buffers (mode 1), or every scene row written -- zeros where the camera saw
    # nothing -- when the destination holds nothing worth keeping (fresh tensors for autograd, or buffers the caller
    # declared uninitialised): no zero-fill and no read-modify-write (mode 2; mode 0 + zero-fill when the camera saw
    # less than an eighth of the scene)
    dense = N > 0 and (M == N or 8 * M >= N)
    if go is not None:
      d_pos, d_ls = go._check("position", pos), go._check("log_scaling", ls)
      d_rot, d_al = go._check("rotation", rot), go._check("alpha_logit", al)
      mode = 1
      if dense and go.claim_overwrite("render_gaussians.backward", go.GEOMETRY):
        mode = 2                           # every scene row written: no zero-fill, no read-modify-write
      elif not dense:
        go.claim_accumulate("render_gaussians.backward", go.GEOMETRY)
    else:
      alloc = torch.empty_like if dense else torch.zeros_like
      d_pos, d_ls, d_rot, d_al = alloc(pos), alloc(ls), alloc(rot), alloc(al)
      mode = 2 if dense else 0
    # the SH coefficient gradient: to the factor collector (data-parallel), into caller-owned buffers, or returned
    want_sh = collector is not None or sh_out is not None or ctx.needs_input_grad[4]
    d_sh, owner, sh_mode = None, None, 0
    if collector is None and want_sh:
      owner = sh_out[2] if (sh_out is not None and len(sh_out) > 2) else None
      d_sh = sh_out[0] if sh_out is not None else torch.empty(N, 3, K, dtype=torch.float32, device=dev)
      if sh_out is None:
        overwrite = True
        if not dense:
          d_sh.zero_()
      elif owner is None:
        overwrite = False                  # caller-owned buffer without an owner object: plain accumulation
      elif dense:
        overwrite = owner.claim_overwrite("render_gaussians.backward", ("feature",))
      else:
        owner.claim_accumulate("render_gaussians.backward", ("feature",))
        overwrite = False
      # 1: every row of d_sh is written (zeros where this camera saw nothing): no zero-fill, no RMW; 2: rows of `indexes` accumulate
      sh_mode = 1 if (overwrite and dense) else 2
    if N > 0:
      # the backward half of the frame behind ONE native call (csrc/frame.hip: gsr_frame_backward):
      # K7 -> packed gradient rows -> (scene row -> visible rank map) -> geometry sweep -> SH coefficient gradient
      live = M > 0 and st.O > 0 and d_image is not None
      if live and st.vis_partial is None:
        raise _lib.GsplatHipError("backward called on a rendering made without gradient state")
      inv = torch.empty(N, dtype=torch.int32, device=dev) if (M < N and dense) else None
      partial = torch.empty(st.O, PARTIAL_FLOATS, dtype=torch.float32, device=dev) if live else None
      grows = torch.empty(M, ROW_FLOATS, dtype=torch.float32, device=dev) if M > 0 else None
      dcol = torch.empty(M, 3, dtype=torch.float32, device=dev) if want_sh else None
      dimg = _f32c(d_image) if live else None
      dg = _f32c(d_g2d) if d_g2d is not None else None
      dd = _f32c(d_depth).reshape(-1) if d_depth is not None else None
      timer = KERNEL_TIMER
      ev = timer.pair("composite_backward") if (timer is not None and live) else (None, None)
      seg = C.addressof(st.segments) if st.segments is not None else None
      args = _lib.GsrFrameBackwardC(
          _ptr(pos), _ptr(ls), _ptr(rot), _ptr(al), _ptr(sh), N, K, st.W, st.H, st.C, _ptr(T), _ptr(proj), _ptr(cam),
          st.params, M, st.O, _ptr(indexes), _ptr(st.rows), _ptr(st.order), _ptr(st.count), _ptr(st.offsets),
          _ptr(st.sorted_splat), _ptr(st.sorted_inst), _ptr(st.pair_vis), _ptr(st.vis_partial), _ptr(st.tile_range),
          _ptr(st.final_T), _ptr(st.last), _ptr(st.image), _ptr(ctx.jac), seg, _ptr(dimg), _ptr(dg), _ptr(dd),
          _ptr(partial), _ptr(grows), _ptr(inv), _ptr(dcol), _ptr(d_pos), _ptr(d_ls), _ptr(d_rot), _ptr(d_al), mode,
          _ptr(d_sh), sh_mode, _ptr(st.prune_cost), _ptr(st.split_score), _ptr(_vis_out(st, live)))
      early = getattr(collector, "on_rows", None) if collector is not None else None
      if early is not None and M > 0:
        # data-parallel: K7 + reduction first; the caller packs the colour-gradient factors straight from the packed rows
        # and starts their all-gather, which then runs next to the sweep enqueued below
        _lib.check(lib.gsr_frame_backward_stages(C.byref(args), 1, ev[0], ev[1], _stream()), "gsr_frame_backward_stages(1)")
        early(indexes, grows, cam)
        _lib.check(lib.gsr_frame_backward_stages(C.byref(args), 2, None, None, _stream()), "gsr_frame_backward_stages(2)")
      else:
        _lib.check(lib.gsr_frame_backward(C.byref(args), ev[0], ev[1], _stream()), "gsr_frame_backward")
    else:
      dcol = torch.zeros(0, 3, dtype=torch.float32, device=dev) if want_sh else None
    if collector is not None:              # data-parallel factor exchange: keep only the colour gradient
      # (4th entry: has the position term of this camera's colour gradient been added to d_position already? -- by the
      # sweep above, from the Jacobian the forward pass saved; K = 1 has no such term)
      collector.items.append((indexes, dcol, cam, K == 1 or ctx.jac is not None))
    if go is not None:
      return (None, None, None, None, d_sh.to(ctx.in_dtypes[4]) if (d_sh is not None and sh_out is None) else None) + \
          nothing[5:]
    dt = ctx.in_dtypes
    return (d_pos.to(dt[0]), d_ls.to(dt[1]), d_rot.to(dt[2]), d_al.to(dt[3]),
            d_sh.to(dt[4]) if (d_sh is not None and sh_out is None) else None) + nothing[5:]


def render_projected(indexes: torch.Tensor, gaussians2d: torch.Tensor, features: torch.Tensor,
                     depth: torch.Tensor, camera_params: CameraParams, config: RasterConfig,
                     render_median_depth: bool = False, _depth_order=None, **_unused) -> Rendering:
  """K4 tile binning -> K5 radix sort -> K6 composite; autograd backward = K7 (+ per-point heuristics).

  ``features`` is (M, C) with C in {1, 2, 3}.  ``points.prune_cost`` / ``points.split_score`` of the
  returned Rendering are filled in place when ``loss.backward()`` runs (trainer.py:512-514 reads them
  afterwards); ``points.visibility`` is available right after the forward pass (reg_loss,
  mlp_scene.py:268-288, needs ``points.visible`` before backward)."""
  _require_device(gaussians2d, features, depth)
  if features.dim() != 2 or not (1 <= features.shape[1] <= 3):
    raise ValueError(f"features must be (M, C) with C in 1..3, got {tuple(features.shape)}")
  if config.tile_size != 16:
    raise ValueError("the HIP kernels are specialised for tile_size=16")
  W, H = camera_params.image_size
  st = _RasterState()
  st.M, st.C, st.W, st.H = int(gaussians2d.shape[0]), int(features.shape[1]), int(W), int(H)
  _init_state(st, camera_params, config, render_median_depth)
  st.needs_grad = torch.is_grad_enabled() and (gaussians2d.requires_grad or features.requires_grad)
  order = None
  if _depth_order is not None and _depth_order[1] is depth and _depth_order[2] == depth._version:
    order = _depth_order[0]                 # enqueued by project_to_image(prefetch=...) for exactly this tensor
  image = _RasterFn.apply(gaussians2d, features, depth, st, order)
  return _rendering_of(st, image, indexes, gaussians2d, depth, camera_params)


def _init_state(st: _RasterState, camera_params: CameraParams, config: RasterConfig, render_median_depth: bool):
  st.params = _lib.raster_params(config)
  st.want_median = bool(render_median_depth)
  st.compute_visibility = bool(config.compute_visibility or config.compute_point_heuristic)
  st.vis_partial = st.pair_vis = st.segments = st.segment_buffers = st.rows = st.image = None
  st.vis_ready, st.vis_capacity = True, 0
  st.seg_pairs, st.seg_min = int(config.segment_pairs), int(config.segment_min_pairs)
  st.near_far = (float(camera_params.near_plane), float(camera_params.far_plane))
  st.key_range = _depth_key_range(*st.near_far)


def _rendering_of(st: _RasterState, image, indexes, gaussians2d, depth, camera_params) -> Rendering:
  # visibility: the tensor when it exists already, otherwise the state's resolver (see _RasterState.materialize_visibility)
  points = RenderedPoints(idx=indexes, depths=depth, opacity=gaussians2d[:, 5], screen_scale=st.screen_scale,
                          visibility=st.visibility if st.vis_ready else st.materialize_visibility,
                          prune_cost=st.prune_cost, split_score=st.split_score)
  return Rendering(image=image, camera=camera_params, points=points, median_depth_image=st.median,
                   final_transmittance=st.final_T, num_overlaps=st.O)


def render_gaussians(gaussians: Gaussians3D, camera_params: CameraParams, config: Optional[RasterConfig] = None,
                     use_sh: bool = False, render_median_depth: bool = False, grad_out: Optional[GradOut] = None,
                     sh_collector=None, **options) -> Rendering:
  """One-call form (splat_trainer/scripts/test_split.py:30): project -> colour -> rasterize.
  ``use_sh``: ``gaussians.feature`` is (N, 3, K) SH coefficients evaluated towards the camera (one fused autograd node,
  ``_FrameFn``); otherwise it is an (N, C) per-point colour and the three calls are chained."""
  config = config or RasterConfig()
  feature = gaussians.feature
  if use_sh and feature.is_cuda and feature.dim() == 3 and feature.shape[1] == 3 and feature.shape[2] in (1, 4, 9, 16):
    return _render_frame(gaussians, camera_params, config, render_median_depth, grad_out, sh_collector)
  prefetch = {}
  if use_sh:
    sh_out = None
    if sh_collector is not None:            # data-parallel: exchange colour-gradient factors, not d_sh (sh.py)
      sh_out = sh_collector
    elif grad_out is not None:
      sh_out = (grad_out._check("feature", feature), grad_out._check("position", gaussians.position), grad_out)
    g2d, depth, indexes = project_to_image(gaussians, camera_params, config, grad_out=grad_out, prefetch=prefetch)
    feats = evaluate_sh_at(feature, gaussians.position, indexes, camera_params.camera_position, grad_out=sh_out)
  else:
    if grad_out is not None:
      raise ValueError("grad_out with use_sh=False: gather the features yourself or use plain autograd")
    g2d, depth, indexes = project_to_image(gaussians, camera_params, config, grad_out=grad_out, prefetch=prefetch)
    feats = feature[indexes]
  return render_projected(indexes, g2d, feats, depth, camera_params, config,
                          render_median_depth=render_median_depth, _depth_order=prefetch.get("depth_order"), **options)


def _render_frame(gaussians: Gaussians3D, camera_params: CameraParams, config: RasterConfig, render_median_depth: bool,
                  grad_out: Optional[GradOut], sh_collector) -> Rendering:
  """render_gaussians(use_sh=True) through the fused node."""
  position, feature = gaussians.position, gaussians.feature
  _require_device(position, gaussians.log_scaling, gaussians.rotation, gaussians.alpha_logit, feature)
  if config.tile_size != 16:
    raise ValueError("the HIP kernels are specialised for tile_size=16")
  sh_out = None
  if sh_collector is not None:              # data-parallel: exchange colour-gradient factors, not d_sh (sh.py)
    sh_out = sh_collector
  elif grad_out is not None:
    sh_out = (grad_out._check("feature", feature), grad_out._check("position", position), grad_out)
  W, H = camera_params.image_size
  st = _RasterState()
  st.M, st.C, st.W, st.H, st.O = 0, 3, int(W), int(H), 0
  _init_state(st, camera_params, config, render_median_depth)
  st.needs_grad = torch.is_grad_enabled() and (grad_out is not None or sh_collector is not None or any(
      t.requires_grad for t in (position, gaussians.log_scaling, gaussians.rotation, gaussians.alpha_logit, feature)))
  cull_args = (int(W), int(H), float(camera_params.near_plane), float(camera_params.far_plane),
               float(config.margin_tiles * config.tile_size))
  # the colour gradient's position term: from the Jacobian the forward pass saves.  Also in data-parallel mode (factor
  # collector): every rank adds the term of its own cameras before the all-reduce, so the multi-camera rebuild does not
  # have to recompute it for all cameras on every rank (it would re-read every coefficient row for that)
  want_pos_grad = _sh.wants_position_grad(position, sh_out) or (
      sh_collector is not None and sh_collector.position_term_local and torch.is_grad_enabled() and
      (position.requires_grad or grad_out is not None))
  image, g2d, depth, indexes = _FrameFn.apply(position, gaussians.log_scaling, gaussians.rotation, gaussians.alpha_logit,
                                              feature, _f32c(camera_params.T_camera_world),
                                              _f32c(camera_params.projection), camera_params.camera_position, cull_args,
                                              st, grad_out, sh_out, want_pos_grad)
  return _rendering_of(st, image, indexes, g2d, depth, camera_params)
