"""CPU oracle for the loss stage next to the rasterizer path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  The reference computes SSIM with the third-party CUDA package ``fused-ssim``
(/root/reference/pyproject.toml:23, unpinned version; call sites splat_trainer/trainer/trainer.py:17,112,450-462 and
splat_trainer/trainer/evaluation.py:7,42), which is neither vendored nor installed here, and the reference holds no
SSIM fixtures.  This file restates the published algorithm that package implements (Wang et al. 2004 SSIM as used by
3D Gaussian Splatting): an 11x11 Gaussian window (sigma = 1.5, separable, normalised), zero padding, per channel,

    mu1 = G*x, mu2 = G*y, s1 = G*x^2 - mu1^2, s2 = G*y^2 - mu2^2, s12 = G*xy - mu1 mu2
    ssim = (2 mu1 mu2 + C1)(2 s12 + C2) / ((mu1^2 + mu2^2 + C1)(s1 + s2 + C2)),   C1 = 0.01^2, C2 = 0.03^2

``padding="same"`` averages the whole map, ``padding="valid"`` averages the map cropped by 5 pixels on every side
(the reference always uses "valid").  Written with differentiable torch ops: autograd is the gradient oracle.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

C1 = 0.01 ** 2
C2 = 0.03 ** 2
WINDOW = 11
SIGMA = 1.5


def gaussian_window(dtype=torch.float64) -> torch.Tensor:
  x = torch.arange(WINDOW, dtype=dtype) - WINDOW // 2
  g = torch.exp(-(x * x) / (2 * SIGMA * SIGMA))
  return g / g.sum()


def ssim_map(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
  """(B,C,H,W) x2 -> (B,C,H,W) SSIM map with zero ("same") padding."""
  B, C, H, W = img1.shape
  g = gaussian_window(img1.dtype).to(img1.device)
  w2d = (g[:, None] * g[None, :])[None, None].expand(C, 1, WINDOW, WINDOW).contiguous()

  def blur(t):
    return F.conv2d(t, w2d, padding=WINDOW // 2, groups=C)

  mu1, mu2 = blur(img1), blur(img2)
  s1 = blur(img1 * img1) - mu1 * mu1
  s2 = blur(img2 * img2) - mu2 * mu2
  s12 = blur(img1 * img2) - mu1 * mu2
  return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def fused_ssim(img1: torch.Tensor, img2: torch.Tensor, padding: str = "same") -> torch.Tensor:
  """Mean SSIM, same call shape as the package the reference imports (trainer.py:112: padding="valid")."""
  m = ssim_map(img1, img2)
  if padding == "valid":
    m = m[:, :, 5:-5, 5:-5]
  return m.mean()


def multiscale_ssim_loss(pred_hwc: torch.Tensor, ref_hwc: torch.Tensor, levels: int = 4, ssim=fused_ssim):
  """trainer.py:450-462 (compute_ssim_loss): 1 - SSIM at ``levels`` scales, 2x average pooling in between.
  Returns (loss, ssim at full resolution)."""
  ref = ref_hwc.unsqueeze(0).permute(0, 3, 1, 2)
  pred = pred_hwc.unsqueeze(0).permute(0, 3, 1, 2)
  s = ssim(pred, ref, padding="valid")
  loss = 1.0 - s
  for _ in range(1, levels):
    pred = F.avg_pool2d(pred, kernel_size=2, stride=2)
    ref = F.avg_pool2d(ref, kernel_size=2, stride=2)
    loss = loss + (1.0 - ssim(pred, ref, padding="valid"))
  return loss / levels, s
