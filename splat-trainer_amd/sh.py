"""evaluate_sh_at: view-dependent colour from spherical-harmonic coefficients (K3).

Call site in the reference: splat_trainer/scene/transfer_sh.py:49
``colors = evaluate_sh_at(self.sh_features, positions, indexes, cam_pos)   # N, 3``
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
  if t is None or t.numel() == 0:
    return None
  return t.data_ptr()          # plain int: the prototypes declare c_void_p


_stream = _lib.current_stream_ptr


class ShFactorCollector:
  """Data-parallel helper: instead of forming the (N,3,K) coefficient gradient per camera, the backward pass of
  ``evaluate_sh_at`` only records the per-camera colour gradient (M,3) here; ``distributed.exchange_sh_factors``
  later all-gathers the 16x smaller factors over the ranks and one fused kernel rebuilds the summed coefficient
  gradient on every rank (csrc/geometry.hip: sh_bwd_multi_kernel)."""

  def __init__(self):
    self.items = []            # (indexes (M,), d_colour (M,3), camera_pos (3,), position term added locally?) in backward order
    # Configuration, the same on every rank (distributed.CameraShardedStep sets it): True = every rank adds the position
    # term of its OWN cameras' colour gradient to the position gradient before the all-reduce (render_gaussians' fused
    # node does, from the Jacobian its forward pass saves), so the multi-camera rebuild leaves the position gradient
    # alone and never reads the coefficient rows; False = the rebuild recomputes the term for all cameras on every rank
    # (what the three-call form needs: evaluate_sh_at hands on colour gradients only).
    self.position_term_local = False
    # Optional hook (distributed.CameraShardedStep): called by the fused node's backward pass as on_rows(indexes (M,),
    # grad_rows (M,16), camera_pos) right behind K7 + the per-splat reduction, BEFORE the geometry sweep is enqueued --
    # the colour-gradient factors are columns 8..10 of the rows, so their exchange can start there.
    self.on_rows = None

  def clear(self):
    self.items.clear()


def as_kernel_inputs(sh_features, positions, camera_pos):
  """The float32 contiguous views the kernels read (no copies when the caller already holds such tensors)."""
  return (sh_features.detach().to(torch.float32).contiguous(), positions.detach().to(torch.float32).contiguous(),
          camera_pos.detach().to(torch.float32).contiguous())


def wants_position_grad(positions, grad_out) -> bool:
  return torch.is_grad_enabled() and not isinstance(grad_out, ShFactorCollector) and \
      (positions.requires_grad or (grad_out is not None and grad_out[1] is not None))


def launch_forward_counted(sh_features, positions, camera_pos, indexes_full, count_dev, want_pos_grad: bool):
  """K3 over an index buffer whose fill count is still on the device: N-sized outputs, rows past the count are left
  unwritten.  render_gaussians enqueues this right behind K1 + K2 so the GPU has work while the host reads the count
  back; the (out, jac) pair it returns is narrowed to M rows and handed to evaluate_sh_at(_precomputed=...)."""
  lib = _lib.load()
  sh, pos, cam = as_kernel_inputs(sh_features, positions, camera_pos)
  N, K = indexes_full.shape[0], sh.shape[2]
  out = torch.empty(N, 3, dtype=torch.float32, device=sh.device)
  jac = torch.empty(N, 9, dtype=torch.float32, device=sh.device) if (want_pos_grad and K > 1 and N > 0) else None
  _lib.check(lib.gsr_sh_forward(_ptr(sh), _ptr(pos), _ptr(indexes_full), N, K, _ptr(cam), _ptr(out), _ptr(jac),
                                _ptr(count_dev), _stream()), "gsr_sh_forward")
  return out, jac


class _SHFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, sh_features, positions, indexes, camera_pos, grad_out, want_pos_grad, precomputed):
    lib = _lib.load()
    sh, pos, cam = as_kernel_inputs(sh_features, positions, camera_pos)
    idx = indexes.contiguous()
    M, K = idx.shape[0], sh.shape[2]
    if precomputed is not None:
      out, jac = precomputed     # launch_forward_counted ran behind the projection, before M was known on the host
    else:
      out = torch.empty(M, 3, dtype=torch.float32, device=sh.device)
      # d colour / d position is cheap to form while the coefficient row is in registers; saving it (36 B per
      # splat) spares the backward pass a second sweep over the 12K-byte rows
      jac = torch.empty(M, 9, dtype=torch.float32, device=sh.device) if (want_pos_grad and K > 1 and M > 0) else None
      _lib.check(lib.gsr_sh_forward(_ptr(sh), _ptr(pos), _ptr(idx), M, K, _ptr(cam), _ptr(out), _ptr(jac), None,
                                    _stream()), "gsr_sh_forward")
    ctx.save_for_backward(sh, pos, idx, cam)
    ctx.jac = jac
    ctx.grad_out = grad_out
    ctx.in_dtypes = (sh_features.dtype, positions.dtype)
    return out

  @staticmethod
  def backward(ctx, d_out):
    lib = _lib.load()
    sh, pos, idx, cam = ctx.saved_tensors
    N, _, K = sh.shape
    M = idx.shape[0]
    go = ctx.grad_out
    if isinstance(go, ShFactorCollector):    # data-parallel factor exchange: keep only the colour gradient
      # (4th entry False: this node hands on colour gradients only -- the position term of the colour gradient is left to
      # the multi-camera rebuild, which the exchange then runs WITH the position gradient as a target; K = 1 has no term)
      go.items.append((idx, d_out.detach().to(torch.float32).contiguous(), cam, K == 1))
      return None, None, None, None, None, None, None
    g = d_out.detach().to(torch.float32).contiguous() if M > 0 else None
    owner = go[2] if (go is not None and len(go) > 2) else None
    dense = N > 0 and (M == N or 8 * M >= N)
    if go is None:
      overwrite = True
    elif owner is None:
      overwrite = False
    elif dense:
      overwrite = owner.claim_overwrite("evaluate_sh_at.backward", ("feature",))
    else:
      # (too few visible rows for the dense overwrite: zero-fill + accumulate; a d_sh that is not the owner's own buffer
      # -- grad_out=(d_sh, d_pos, owner) -- is zero-filled below)
      overwrite = bool(owner.claim_accumulate("evaluate_sh_at.backward", ("feature",))) and owner.feature is not go[0]
    if go is not None:                       # fused "+=" into caller-owned buffers (see renderer.GradOut)
      d_sh, d_pos = go[0], go[1]
      if owner is not None and d_pos is not None and K > 1:
        # this pass ADDS to d_pos and autograd runs it before the projection's backward pass
        owner.claim_accumulate("evaluate_sh_at.backward", owner.GEOMETRY)
    else:
      d_sh = torch.empty(N, 3, K, dtype=torch.float32, device=pos.device)
      d_pos = torch.zeros_like(pos) if ctx.needs_input_grad[1] else None
    if overwrite and dense:
      # every row of d_sh is written (zeros where this camera saw nothing): no zero-fill, no read-modify-write
      inv = None
      if M < N:
        inv = torch.empty(N, dtype=torch.int32, device=pos.device)
        _lib.check(lib.gsr_inverse_map(_ptr(idx), M, N, _ptr(inv), _stream()), "gsr_inverse_map")
      _lib.check(lib.gsr_sh_backward_dense(_ptr(g), _ptr(sh), _ptr(pos), _ptr(inv), M, N, K, _ptr(cam), _ptr(ctx.jac),
                                           _ptr(d_sh), _ptr(d_pos), _stream()), "gsr_sh_backward_dense")
    else:
      if overwrite:
        d_sh.zero_()
      if M > 0:
        _lib.check(lib.gsr_sh_backward(_ptr(g), _ptr(sh), _ptr(pos), _ptr(idx), M, K, _ptr(cam), _ptr(ctx.jac),
                                       _ptr(d_sh), _ptr(d_pos), 1, _stream()), "gsr_sh_backward")
    if go is not None:
      return None, None, None, None, None, None, None
    return (d_sh.to(ctx.in_dtypes[0]), d_pos.to(ctx.in_dtypes[1]) if d_pos is not None else None,
            None, None, None, None, None)


def evaluate_sh_at(sh_features: torch.Tensor, positions: torch.Tensor, indexes: torch.Tensor,
                   camera_pos: torch.Tensor, grad_out=None, _precomputed=None) -> torch.Tensor:
  """``sh_features (N,3,K)``, ``positions (N,3)``, ``indexes (M,) int64``, ``camera_pos (3,)`` -> ``(M,3)``.

  colour_c = 0.5 + sum_k sh[idx, c, k] * Y_k(normalize(positions[idx] - camera_pos)), K in {1,4,9,16}
  (degrees 0..3, basis order k = n(n+1)+m as splat_trainer/scene/mlp/rsh.py).  Differentiable wrt
  ``sh_features`` and, through the view direction, ``positions``; the caller clamps (transfer_sh.py:50).
  ``grad_out=(d_sh, d_positions[, owner])``: optional fused accumulation, see ``renderer.GradOut`` (when
  ``owner.feature_uninitialized`` is set, ``d_sh`` is overwritten row for row instead of added to, and the flag is cleared)."""
  for t in (sh_features, positions, indexes, camera_pos):
    if not t.is_cuda:
      raise _lib.GsplatHipError("evaluate_sh_at runs only on a HIP device; there is no CPU fallback")
  if sh_features.dim() != 3 or sh_features.shape[1] != 3 or sh_features.shape[2] not in (1, 4, 9, 16):
    raise ValueError(f"sh_features must be (N,3,K) with K in (1,4,9,16), got {tuple(sh_features.shape)}")
  if indexes.dtype != torch.int64:
    raise TypeError("indexes must be int64")
  want_pos_grad = wants_position_grad(positions, grad_out)
  return _SHFn.apply(sh_features, positions, indexes, camera_pos, grad_out, want_pos_grad, _precomputed)
