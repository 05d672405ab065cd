"""Occupancy statistics of (tile, splat) pairs on the bench workload: how much of a 16x16 tile a pair really touches."""
import sys, torch
sys.path.insert(0, ".")
import bench
import splat_trainer_amd as sta
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
g, cams, w = bench.make_workload(wl, 1)
dev = "cuda"
cfg = sta.RasterConfig()
g = g.to(dev)
g2d, depth, idx = sta.project_to_image(g, cams[0].to(dev), cfg)
u, v, A, B, C, op = g2d.unbind(1)
# support: q <= min(9, 2 ln(255 op))
qmax = torch.minimum(torch.full_like(op, 9.0), 2 * torch.log(255 * op.clamp_min(1e-8)))
ok = qmax > 0
det = A * C - B * B
hx = torch.sqrt(qmax.clamp_min(0) * C / det)
hy = torch.sqrt(qmax.clamp_min(0) * A / det)
W, H = w["w"], w["h"]
def cells(sz):
  x0 = ((u - hx - 0.5) / sz).floor().clamp(0, (W - 1) // sz); x1 = ((u + hx - 0.5) / sz).floor().clamp(0, (W - 1) // sz)
  y0 = ((v - hy - 0.5) / sz).floor().clamp(0, (H - 1) // sz); y1 = ((v + hy - 0.5) / sz).floor().clamp(0, (H - 1) // sz)
  inside = (u + hx > 0) & (u - hx < W) & (v + hy > 0) & (v - hy < H) & ok
  return ((x1 - x0 + 1) * (y1 - y0 + 1) * inside).sum().item()
n16, n8, n4 = cells(16), cells(8), cells(4)
area = (torch.pi * qmax.clamp_min(0) / torch.sqrt(det) * ok).sum().item()
print(f"{wl}: splats {u.numel()}  bbox tile pairs {n16:.0f}  8x8 cells {n8:.0f} ({n8 / (4 * n16):.2%} of quadrants)  "
      f"4x4 cells {n4:.0f} ({n4 / (16 * n16):.2%})  ellipse px {area:.0f} ({area / (256 * n16):.2%} of tile px)")
print("median hx, hy:", hx[ok].median().item(), hy[ok].median().item(), " opacity median", op.median().item())
