"""How many 8x8 quadrants of a 16x16 tile does a (tile, splat) pair touch?  (bounding-box estimate, bench workloads)"""
import sys, torch
sys.path.insert(0, ".")
import bench
import splat_trainer_amd as sta
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
g, cams, w = bench.make_workload(wl, 1)
g2d, depth, idx = sta.project_to_image(g.to("cuda"), cams[0].to("cuda"), sta.RasterConfig())
u, v, A, B, C, op = g2d.unbind(1)
qmax = torch.minimum(torch.full_like(op, 9.0), 2 * torch.log(255 * op.clamp_min(1e-8))).clamp_min(0)
det = A * C - B * B
hx, hy = torch.sqrt(qmax * C / det), torch.sqrt(qmax * A / det)
W, H = w["w"], w["h"]
def axis(c, h, n):
  c0 = ((c - h - 0.5) / 8).floor().clamp(0, (n - 1) // 8).long(); c1 = ((c + h - 0.5) / 8).floor().clamp(0, (n - 1) // 8).long()
  t0, t1 = c0 // 2, c1 // 2
  ntile = t1 - t0 + 1
  single = ((c0 % 2 == 1).long() + (c1 % 2 == 0).long())
  single = torch.where((t0 == t1) & (c0 != c1), torch.zeros_like(single), torch.where((t0 == t1), torch.ones_like(single), single))
  return ntile, single, ntile - single
ok = (qmax > 0) & (u + hx > 0) & (u - hx < W) & (v + hy > 0) & (v - hy < H)
nx, sx, fx = axis(u, hx, W); ny, sy, fy = axis(v, hy, H)
m = ok.long()
one = (sx * sy * m).sum().item(); lr = (fx * sy * m).sum().item(); tb = (sx * fy * m).sum().item(); four = (fx * fy * m).sum().item()
tot = one + lr + tb + four
print(f"{wl}: pairs {tot}  1 quadrant {one / tot:.1%}  2 side by side (one half) {lr / tot:.1%}  2 stacked (both halves, one side) {tb / tot:.1%}  all 4 {four / tot:.1%}")
halves = one + lr + 2 * tb + 2 * four
print(f"halves evaluated per pair {halves / tot:.2f}; of those with a single quadrant: {(one + 2 * tb) / halves:.1%}")
