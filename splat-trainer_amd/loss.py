"""fused_ssim: the loss stage next to the rasterizer path.

Same name and call shape as the CUDA-only ``fused_ssim`` package the reference imports
(splat_trainer/trainer/trainer.py:17,112: ``partial(fused_ssim, padding="valid")``; trainer/evaluation.py:42), so
``from splat_trainer_amd import fused_ssim`` is an import swap.  Mean SSIM with an 11x11 Gaussian window
(sigma 1.5, zero padding, C1 = 0.01^2, C2 = 0.03^2) of ``img1`` (the prediction, differentiable) against ``img2``,
both (B, C, H, W) in any memory format -- the reference passes channels_last views of (H, W, 3) images
(trainer.py:450-452); they are read through their strides, no copy is made.  HIP kernels behind the C ABI
(csrc/ssim.hip); no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib


def _ptr(t):
  return None if t is None or t.numel() == 0 else t.data_ptr()     # plain int: the prototypes declare c_void_p


_stream = _lib.current_stream_ptr


def _strides(t: torch.Tensor):
  return (C.c_int64 * 4)(*t.stride())


class _SSIMFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, img1, img2, crop, train):
    lib = _lib.load()
    x = img1.detach().to(torch.float32)
    y = img2.detach().to(torch.float32)
    B, Cc, H, W = x.shape
    dev = x.device
    mean = torch.empty(1, dtype=torch.float32, device=dev)
    maps = torch.empty(3, B, Cc, H, W, dtype=torch.float32, device=dev) if train else None
    ws_bytes = lib.gsr_ssim_workspace_bytes(B, Cc, H, W)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    m = (maps[0], maps[1], maps[2]) if train else (None, None, None)
    _lib.check(lib.gsr_ssim_forward(_ptr(x), _ptr(y), _strides(x), _strides(y), B, Cc, H, W, crop, _ptr(mean),
                                    _ptr(m[0]), _ptr(m[1]), _ptr(m[2]), _ptr(ws), ws_bytes, _stream()),
               "gsr_ssim_forward")
    ctx.save_for_backward(x, y, maps)
    ctx.in_dtype = img1.dtype
    return mean[0]

  @staticmethod
  def backward(ctx, g):
    lib = _lib.load()
    x, y, maps = ctx.saved_tensors
    if maps is None:
      raise _lib.GsplatHipError("fused_ssim(train=False) result cannot be back-propagated")
    B, Cc, H, W = x.shape
    gs = g if (g.dtype is torch.float32 and g.is_contiguous()) else g.detach().to(torch.float32).contiguous()   # one scalar
    d = torch.empty_strided(x.shape, x.stride(), dtype=torch.float32, device=x.device)
    _lib.check(lib.gsr_ssim_backward(_ptr(x), _ptr(y), _strides(x), _strides(y), _strides(d), B, Cc, H, W,
                                     _ptr(maps[0]), _ptr(maps[1]), _ptr(maps[2]), _ptr(gs), _ptr(d), _stream()),
               "gsr_ssim_backward")
    return d.to(ctx.in_dtype), None, None, None


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor, padding: str = "same", train: bool = True) -> torch.Tensor:
  if padding not in ("same", "valid"):
    raise ValueError("padding must be 'same' or 'valid'")
  if img1.dim() != 4 or img1.shape != img2.shape:
    raise ValueError(f"expected two (B,C,H,W) tensors of equal shape, got {tuple(img1.shape)} and {tuple(img2.shape)}")
  if not (img1.is_cuda and img2.is_cuda):
    raise _lib.GsplatHipError("fused_ssim runs only on a HIP device; there is no CPU fallback")
  crop = 5 if padding == "valid" else 0
  if img1.shape[2] <= 2 * crop or img1.shape[3] <= 2 * crop:
    raise ValueError("image too small for padding='valid' (needs more than 10 pixels per side)")
  train = bool(train and torch.is_grad_enabled() and img1.requires_grad)
  return _SSIMFn.apply(img1, img2, crop, train)


class _PixelLossFn(torch.autograd.Function):
  @staticmethod
  def forward(ctx, image, target, kind, lo, hi):
    lib = _lib.load()
    plain = lambda a: a.dtype is torch.float32 and a.is_contiguous()
    x = image.detach() if plain(image) else image.detach().to(torch.float32).contiguous()
    t = target if (plain(target) and target.shape == x.shape) else target.detach().to(torch.float32).expand_as(x).contiguous()
    n = x.numel()
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    ws_bytes = lib.gsr_pixel_loss_workspace_bytes(n)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    _lib.check(lib.gsr_pixel_loss_forward(_ptr(x), _ptr(t), n, kind, lo, hi, _ptr(out), _ptr(ws), ws_bytes, _stream()),
               "gsr_pixel_loss_forward")
    ctx.save_for_backward(x, t)
    ctx.args = (kind, lo, hi, image.dtype)
    return out[0]

  @staticmethod
  def backward(ctx, g):
    lib = _lib.load()
    x, t = ctx.saved_tensors
    kind, lo, hi, dtype = ctx.args
    gs = g.detach().to(torch.float32).reshape(1).contiguous()
    d = torch.empty_like(x)
    _lib.check(lib.gsr_pixel_loss_backward(_ptr(x), _ptr(t), x.numel(), kind, lo, hi, _ptr(gs), _ptr(d), _stream()),
               "gsr_pixel_loss_backward")
    return d.to(dtype), None, None, None, None


def _pixel_loss(image, target, kind, clamp):
  if not (image.is_cuda and target.is_cuda):
    raise _lib.GsplatHipError("the fused pixel losses run only on a HIP device; there is no CPU fallback")
  lo, hi = (float(clamp[0]), float(clamp[1])) if clamp is not None else (-3.0e38, 3.0e38)
  return _PixelLossFn.apply(image, target, kind, lo, hi)


def clamped_mse_loss(image: torch.Tensor, target: torch.Tensor, clamp=(0.0, 1.0)) -> torch.Tensor:
  """``F.mse_loss(image.clamp(*clamp), target)`` (trainer.py:472-475 on the post-activation image,
  scene/color_model.py:154-160) in one forward pass + one backward pass; ``clamp=None``: no clamp."""
  return _pixel_loss(image, target, 0, clamp)


def clamped_l1_loss(image: torch.Tensor, target: torch.Tensor, clamp=(0.0, 1.0)) -> torch.Tensor:
  """``F.l1_loss(image.clamp(*clamp), target)`` (trainer.py:470-471), fused the same way."""
  return _pixel_loss(image, target, 1, clamp)


class _MSLossFn(torch.autograd.Function):
  """The reference's loss mix behind one native call per direction (csrc/ssim.hip: gsr_msloss_forward / _backward)."""

  @staticmethod
  def forward(ctx, image, target, weights, levels, lo, hi):
    lib = _lib.load()
    x = image.detach()
    t = target.detach()
    H, W, Cc = x.shape
    ws_bytes = lib.gsr_msloss_workspace_bytes(H, W, Cc, levels)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x.device)
    metrics = torch.empty(3 + levels, dtype=torch.float32, device=x.device)
    _lib.check(lib.gsr_msloss_forward(_ptr(x), _ptr(t), H, W, Cc, levels, weights[0], weights[1], weights[2], lo, hi,
                                      _ptr(metrics), _ptr(ws), ws_bytes, _stream()), "gsr_msloss_forward")
    ctx.save_for_backward(x, t, ws)
    ctx.args = (weights, levels, lo, hi, ws_bytes)
    ctx.mark_non_differentiable(metrics)
    return metrics[0], metrics

  @staticmethod
  def backward(ctx, g, _g_metrics):
    lib = _lib.load()
    x, t, ws = ctx.saved_tensors
    weights, levels, lo, hi, ws_bytes = ctx.args
    H, W, Cc = x.shape
    gs = g.detach().to(torch.float32).reshape(1).contiguous()
    d = torch.empty_like(x)
    _lib.check(lib.gsr_msloss_backward(_ptr(x), _ptr(t), H, W, Cc, levels, weights[0], weights[1], weights[2], lo, hi,
                                       _ptr(gs), _ptr(ws), ws_bytes, _ptr(d), _stream()), "gsr_msloss_backward")
    return d, None, None, None, None, None


def reference_loss(image: torch.Tensor, target: torch.Tensor, l1_weight: float = 0.0, mse_weight: float = 10.0,
                   ssim_weight: float = 1.0, ssim_levels: int = 4, clamp=(0.0, 1.0), fused: bool = True,
                   return_metrics: bool = False):
  """The reference's per-camera loss mix, ``Trainer.compute_losses`` without ``reg_loss`` (trainer.py:448-488): L1 and
  MSE of the image against the target, and the multi-scale SSIM loss -- ``fused_ssim(padding="valid")`` on the
  channels_last view of the (H, W, 3) image and on ``ssim_levels - 1`` successive 2 x 2 average poolings of it, the mean
  of the levels' ``1 - ssim`` -- each times its weight (defaults: config/trainer/default.yaml:52-54 for the weights,
  trainer/config.py:70 for the levels; all three terms are formed whatever their weight, as the reference does).
  ``image`` is the UNclamped rendering; the reference receives it clamped from ``MLPScene.render`` (mlp_scene.py:421-423),
  which ``clamp`` restates.  The reference's ``.item()`` calls (logged metrics) are not part of the arithmetic;
  ``return_metrics`` also returns the device tensor [loss, l1, mse, ssim_0 .. ssim_{levels-1}] they would read.

  ``fused`` (default): ONE native call per direction (pyramids of both images + the pixel losses in one pass, SSIM per
  level, one combining pass back through the poolings and the clamp) instead of ~60 torch launches around fused_ssim;
  needs a contiguous float32 (H, W, C <= 4) image, ``ssim_levels`` <= 4 and more than 10 pixels per side on the coarsest
  level, else -- and with ``fused=False`` -- the composition below (the form the fused path is tested against)."""
  import torch.nn.functional as F
  ok = (fused and clamp is not None and image.is_cuda and target.is_cuda and image.dim() == 3 and image.shape == target.shape and
        image.dtype is torch.float32 and target.dtype is torch.float32 and image.is_contiguous() and target.is_contiguous() and
        1 <= image.shape[2] <= 4 and 1 <= ssim_levels <= 4 and
        min(image.shape[0], image.shape[1]) >> (ssim_levels - 1) > 10)
  if ok:
    loss, metrics = _MSLossFn.apply(image, target, (float(l1_weight), float(mse_weight), float(ssim_weight)),
                                    int(ssim_levels), float(clamp[0]), float(clamp[1]))
    return (loss, metrics) if return_metrics else loss
  l1 = clamped_l1_loss(image, target, clamp)
  mse = clamped_mse_loss(image, target, clamp)
  img = image.clamp(*clamp) if clamp is not None else image
  ref = target.unsqueeze(0).permute(0, 3, 1, 2)
  pred = img.unsqueeze(0).permute(0, 3, 1, 2)
  ssims = [fused_ssim(pred, ref, padding="valid")]
  for _ in range(1, ssim_levels):
    pred = F.avg_pool2d(pred, kernel_size=2, stride=2)
    ref = F.avg_pool2d(ref, kernel_size=2, stride=2)
    ssims.append(fused_ssim(pred, ref, padding="valid"))
  loss = l1 * l1_weight + mse * mse_weight + (sum(1.0 - s for s in ssims) / ssim_levels) * ssim_weight
  if return_metrics:
    return loss, torch.stack([loss.detach(), l1.detach(), mse.detach()] + [s.detach() for s in ssims])
  return loss
