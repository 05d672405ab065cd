"""Fused SSIM.  CPU: the oracle against a direct definition-level evaluation and known
properties.  GPU: the HIP kernels through the C ABI against the oracle (fp64) -- value and gradient, both paddings,
planar and channels_last inputs, odd sizes -- and the reference's multi-scale loss (trainer.py:450-462)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import ssim_oracle


def _naive_ssim_at(x, y, py, px):
  """SSIM of one pixel straight from the definition (zero padding), pure Python loops."""
  H, W = x.shape
  g = ssim_oracle.gaussian_window(torch.float64)
  m1 = m2 = m11 = m22 = m12 = 0.0
  for dy in range(-5, 6):
    for dx in range(-5, 6):
      yy, xx = py + dy, px + dx
      if 0 <= yy < H and 0 <= xx < W:
        w = (g[dy + 5] * g[dx + 5]).item()
        a, b = x[yy, xx].item(), y[yy, xx].item()
        m1 += w * a; m2 += w * b; m11 += w * a * a; m22 += w * b * b; m12 += w * a * b
  s1, s2, s12 = m11 - m1 * m1, m22 - m2 * m2, m12 - m1 * m2
  return ((2 * m1 * m2 + ssim_oracle.C1) * (2 * s12 + ssim_oracle.C2)) / ((m1 * m1 + m2 * m2 + ssim_oracle.C1) * (s1 + s2 + ssim_oracle.C2))


def test_oracle_matches_definition_and_properties():
  torch.manual_seed(0)
  x = torch.rand(1, 2, 19, 23, dtype=torch.float64)
  y = (x + 0.2 * torch.randn_like(x)).clamp(0, 1)
  m = ssim_oracle.ssim_map(x, y)
  for (py, px) in [(0, 0), (9, 11), (18, 22), (5, 3), (13, 20)]:
    assert abs(m[0, 1, py, px].item() - _naive_ssim_at(x[0, 1], y[0, 1], py, px)) < 1e-12
  assert abs(ssim_oracle.fused_ssim(x, x, "valid").item() - 1.0) < 1e-12          # identical images
  assert ssim_oracle.fused_ssim(x, y, "valid").item() < 1.0
  assert abs(ssim_oracle.fused_ssim(x, y, "same").item() - ssim_oracle.fused_ssim(y, x, "same").item()) < 1e-12
  g = ssim_oracle.gaussian_window()
  assert abs(g.sum().item() - 1) < 1e-15 and g.argmax().item() == 5
  xg = x.clone().requires_grad_(True)
  assert torch.autograd.gradcheck(lambda t: ssim_oracle.fused_ssim(t, y, "valid"), (xg,), eps=1e-6, atol=1e-7)


def test_fused_ssim_refuses_cpu():
  import splat_trainer_amd as sta
  with pytest.raises(sta.GsplatHipError):
    sta.fused_ssim(torch.rand(1, 3, 32, 32), torch.rand(1, 3, 32, 32))
  with pytest.raises(ValueError):
    sta.fused_ssim(torch.rand(1, 3, 32, 32).to("meta") if False else torch.rand(3, 32, 32), torch.rand(3, 32, 32))


@pytest.mark.gpu
@pytest.mark.parametrize("shape,padding,layout", [((1, 3, 64, 80), "valid", "nchw"), ((2, 1, 33, 47), "same", "nchw"),
                                                  ((1, 3, 100, 37), "valid", "hwc"), ((1, 3, 16, 16), "same", "hwc"),
                                                  ((1, 3, 12, 200), "valid", "nchw")])
def test_hip_ssim_matches_oracle(shape, padding, layout):
  import splat_trainer_amd as sta
  torch.manual_seed(sum(shape))
  B, C, H, W = shape
  x = torch.rand(B, C, H, W)
  y = (x + 0.15 * torch.randn(B, C, H, W)).clamp(0, 1)
  if layout == "hwc":      # the reference's channels_last view of an (H, W, C) image (trainer.py:450-452)
    xd = x[0].permute(1, 2, 0).contiguous().cuda().requires_grad_(True)
    yd = y[0].permute(1, 2, 0).contiguous().cuda()
    a, b = xd.unsqueeze(0).permute(0, 3, 1, 2), yd.unsqueeze(0).permute(0, 3, 1, 2)
    assert not a.is_contiguous()
  else:
    xd = x.clone().cuda().requires_grad_(True)
    yd = y.cuda()
    a, b = xd, yd
  got = sta.fused_ssim(a, b, padding=padding)
  xo = x.double().requires_grad_(True)
  want = ssim_oracle.fused_ssim(xo, y.double(), padding)
  assert got.dim() == 0 and abs(got.item() - want.item()) < 2e-6
  (got * 3.0).backward()
  (want * 3.0).backward()
  gd = xd.grad if layout != "hwc" else xd.grad.permute(2, 0, 1).unsqueeze(0)
  err = (gd.cpu().double() - xo.grad).abs().max().item() / xo.grad.abs().max().item()
  assert err < 1e-4, err
  # deterministic (no atomics: two runs are bit-identical) and usable without grad
  nograd = sta.fused_ssim(a.detach(), b, padding=padding, train=False)
  assert nograd.item() == sta.fused_ssim(a.detach(), b, padding=padding, train=False).item()
  assert abs(nograd.item() - got.item()) < 1e-6
  assert sta.fused_ssim(a, b, padding=padding).item() == got.item()


@pytest.mark.gpu
def test_reference_multiscale_loss_flow():
  """trainer.py:450-462 with the import swapped: 3 levels, 2x average pooling, padding='valid', HWC image."""
  import splat_trainer_amd as sta
  from functools import partial
  torch.manual_seed(5)
  H, W = 270, 480
  ref = torch.rand(H, W, 3)
  pred = (ref + 0.1 * torch.randn(H, W, 3)).clamp(0, 1)
  ssim_hip = partial(sta.fused_ssim, padding="valid")                      # trainer.py:112

  def compute_ssim_loss(pred, ref, levels, ssim):                          # trainer.py:450-462, verbatim structure
    ref = ref.unsqueeze(0).permute(0, 3, 1, 2).to(memory_format=torch.channels_last)
    pred = pred.unsqueeze(0).permute(0, 3, 1, 2).to(memory_format=torch.channels_last)
    s = ssim(pred, ref)
    loss = 1.0 - s
    for _ in range(1, levels):
      pred = F.avg_pool2d(pred, kernel_size=2, stride=2)
      ref = F.avg_pool2d(ref, kernel_size=2, stride=2)
      loss = loss + (1.0 - ssim(pred, ref))
    return loss / levels, s.item()

  pd = pred.clone().cuda().requires_grad_(True)
  loss, s0 = compute_ssim_loss(pd, ref.cuda(), 3, ssim_hip)
  loss.backward()
  po = pred.clone().double().requires_grad_(True)
  oloss, os0 = ssim_oracle.multiscale_ssim_loss(po, ref.double(), levels=3)
  oloss.backward()
  assert abs(loss.item() - oloss.item()) < 2e-6 and abs(s0 - os0.item()) < 2e-6
  err = (pd.grad.cpu().double() - po.grad).abs().max().item() / po.grad.abs().max().item()
  assert err < 1e-4, err


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(37, 53, 3), (1080, 1920, 3), (5,)])
def test_fused_pixel_losses_match_torch(shape):
  """clamped_mse_loss / clamped_l1_loss (csrc/ssim.hip: pixel_loss_*) vs F.mse_loss / F.l1_loss on the clamped image,
  value and gradient, incl. pixels outside [0, 1] (zero gradient) and exactly on the clamp bounds (gradient passes)."""
  import torch.nn.functional as F
  import splat_trainer_amd as sta
  gen = torch.Generator(device="cuda").manual_seed(1)
  x0 = torch.rand(shape, device="cuda", generator=gen) * 1.6 - 0.3
  x0.view(-1)[0], x0.view(-1)[1] = 0.0, 1.0
  t = torch.rand(shape, device="cuda", generator=gen)
  for fused, ref in ((sta.clamped_mse_loss, F.mse_loss), (sta.clamped_l1_loss, F.l1_loss)):
    a = x0.clone().requires_grad_(True)
    b = x0.clone().requires_grad_(True)
    la = fused(a, t) * 3.0
    lb = ref(b.clamp(0, 1), t) * 3.0
    la.backward()
    lb.backward()
    assert torch.allclose(la, lb, rtol=2e-6, atol=1e-8), (la.item(), lb.item())
    assert torch.allclose(a.grad, b.grad, rtol=1e-6, atol=1e-12)
  c = x0.clone().requires_grad_(True)
  assert torch.allclose(sta.clamped_mse_loss(c, t, clamp=None), F.mse_loss(x0, t), rtol=2e-6)


@pytest.mark.gpu
def test_reference_loss_mix_matches_the_torch_expression():
  """loss.reference_loss = Trainer.compute_losses without reg_loss (trainer.py:448-488): L1 + MSE + multi-scale SSIM
  with the reference's weights, against the same expression in fp64 torch + the SSIM oracle; value within 2e-6, gradient
  within 1e-4 of its largest entry (pixels outside [0, 1] included: the clamp passes no gradient there)."""
  import splat_trainer_amd as sta
  torch.manual_seed(9)
  H, W = 270, 480
  ref = torch.rand(H, W, 3)
  pred = ref + 0.1 * torch.randn(H, W, 3) + 0.05                       # some pixels leave [0, 1]
  for weights in (dict(l1_weight=0.0, mse_weight=10.0, ssim_weight=1.0, ssim_levels=4),
                  dict(l1_weight=1.0, mse_weight=0.5, ssim_weight=0.2, ssim_levels=3),
                  dict(l1_weight=0.3, mse_weight=2.0, ssim_weight=1.0, ssim_levels=4, fused=False),
                  dict(l1_weight=0.3, mse_weight=2.0, ssim_weight=1.0, ssim_levels=1)):
    pd = pred.clone().cuda().requires_grad_(True)
    got, metrics = sta.reference_loss(pd, ref.cuda(), return_metrics=True, **weights)
    (got * 1.5).backward()
    weights = {k: v for k, v in weights.items() if k != "fused"}
    po = pred.clone().double().requires_grad_(True)
    img = po.clamp(0, 1)
    ssim_l, _ = ssim_oracle.multiscale_ssim_loss(img, ref.double(), levels=weights["ssim_levels"])
    want = (F.l1_loss(img, ref.double()) * weights["l1_weight"] + F.mse_loss(img, ref.double()) * weights["mse_weight"] +
            ssim_l * weights["ssim_weight"])
    (want * 1.5).backward()
    assert abs(got.item() - want.item()) < 2e-6 * max(1.0, abs(want.item())), (got.item(), want.item())
    err = (pd.grad.cpu().double() - po.grad).abs().max().item() / po.grad.abs().max().item()
    assert err < 1e-4, err
    # the metrics the reference logs with .item(): l1, mse and the full-resolution ssim
    m = metrics.cpu().double()
    assert abs(m[0] - want.item()) < 2e-6 * max(1.0, abs(want.item()))
    assert abs(m[1] - F.l1_loss(img, ref.double()).item()) < 2e-6 and abs(m[2] - F.mse_loss(img, ref.double()).item()) < 2e-6
    assert abs(m[3] - ssim_oracle.fused_ssim(img.detach().unsqueeze(0).permute(0, 3, 1, 2), ref.double().unsqueeze(0).permute(0, 3, 1, 2), "valid").item()) < 2e-6


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(135, 181, 3), (97, 203, 1), (1080, 1920, 3)])
def test_fused_reference_loss_equals_its_composition(shape):
  """The one-call fused loss against the torch composition around fused_ssim (odd sizes: the poolings drop the last row /
  column, 1080p: the bench's shape): value within 2e-6, gradient within 2e-5 of its largest entry, and two runs bit-equal."""
  import splat_trainer_amd as sta
  gen = torch.Generator(device="cuda").manual_seed(3)
  ref = torch.rand(shape, device="cuda", generator=gen)
  pred = ref + 0.1 * torch.randn(shape, device="cuda", generator=gen) + 0.03
  levels = 4 if min(shape[0], shape[1]) >= 97 else 3
  outs = []
  for fused in (True, False, True):
    p = pred.clone().requires_grad_(True)
    loss = sta.reference_loss(p, ref, l1_weight=0.2, mse_weight=10.0, ssim_weight=1.0, ssim_levels=levels, fused=fused)
    loss.backward()
    outs.append((loss.detach(), p.grad))
  assert abs(outs[0][0].item() - outs[1][0].item()) < 2e-6 * max(1.0, abs(outs[1][0].item()))
  err = (outs[0][1] - outs[1][1]).abs().max().item() / outs[1][1].abs().max().item()
  assert err < 2e-5, err
  assert outs[0][0].item() == outs[2][0].item() and torch.equal(outs[0][1], outs[2][1])
