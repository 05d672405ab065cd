"""Monte-Carlo footprint model of the composite kernels' work per (tile, splat) pair (no GPU needed): random splats with the
benchmark scenes' statistics (projected sigma 2.0 px for c2 / 1.5 px for c3, log-normal per-axis spread 0.3, random
rotation, 0.3 px^2 blur, opacity = sigmoid(1.5 N(0,1))) laid over a 16 x 16 tile grid.  Reproduces the measured frames'
pairs per splat (2.87 vs 2.93 on c2), tile halves per pair (1.40) and contributing pixels per pair (44 vs 41), and from
there predicts what a window mapping would evaluate -- the numbers DESIGN.md section 4 ("Round 4") quotes for the two
windowed K7 variants before they were built:
    part 1: 64-pixel windows (8x8 / 16x4 / 4x16) over (pixel box of the splat) x tile  -> windows per pair
    part 2: 8-wide windows sliding in x inside a tile half (register-resident state)   -> evaluation time per pair
    python tools/footprint_model.py"""
import numpy as np
rng = np.random.default_rng(0)
def run(s_px, n=40000, label=""):
    # random 3D gaussians -> 2D cov (orthographic approx), + 0.3 blur
    s = s_px * np.exp(0.3 * rng.standard_normal((n, 3)))
    q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    R = np.stack([1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w),
                  2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], 1).reshape(n, 2, 3)
    M = R * s[:, None, :]
    cov = M @ M.transpose(0, 2, 1) + 0.3 * np.eye(2)
    op = 1 / (1 + np.exp(-1.5 * rng.standard_normal(n)))
    qlim = np.minimum(9.0, 2 * np.log(np.maximum(op * 255, 1e-9)))
    keep = qlim > 0
    cov, qlim = cov[keep], qlim[keep]; n = len(qlim)
    det = cov[:, 0, 0] * cov[:, 1, 1] - cov[:, 0, 1] ** 2
    A = cov[:, 1, 1] / det; B = -cov[:, 0, 1] / det; C = cov[:, 0, 0] / det
    u = rng.uniform(32, 48, n); v = rng.uniform(32, 48, n)   # centre in tile (2,2) of a 5x5 tile patch
    rx = np.sqrt(qlim * cov[:, 0, 0]); ry = np.sqrt(qlim * cov[:, 1, 1])
    tot = dict(pairs=0, contrib=0, halves=0, quads=0, win=0, win_b=0, bboxpx=0)
    hist = np.zeros(8, int)
    gx = np.arange(80) + 0.5
    for i in range(n):
        dx = gx[None, :] - u[i]; dy = gx[:, None] - v[i]
        qq = A[i] * dx * dx + 2 * B[i] * dx * dy + C[i] * dy * dy
        hit = qq <= qlim[i]
        if not hit.any(): continue
        # analytic bbox in pixel indices (pixel centres inside [u-rx, u+rx])
        X0 = int(np.ceil(u[i] - rx[i] - 0.5)); X1 = int(np.floor(u[i] + rx[i] - 0.5))
        Y0 = int(np.ceil(v[i] - ry[i] - 0.5)); Y1 = int(np.floor(v[i] + ry[i] - 0.5))
        for ty in range(5):
            for tx in range(5):
                blk = hit[16*ty:16*ty+16, 16*tx:16*tx+16]
                c = blk.sum()
                if c == 0: continue
                tot['pairs'] += 1; tot['contrib'] += c
                tot['halves'] += int(blk[:8].any()) + int(blk[8:].any())
                tot['quads'] += int(blk[:8,:8].any()) + int(blk[:8,8:].any()) + int(blk[8:,:8].any()) + int(blk[8:,8:].any())
                bx0 = max(X0 - 16*tx, 0); bx1 = min(X1 - 16*tx, 15); by0 = max(Y0 - 16*ty, 0); by1 = min(Y1 - 16*ty, 15)
                w = bx1 - bx0 + 1; h = by1 - by0 + 1
                tot['bboxpx'] += w * h
                if w <= 8 and h <= 8: nw = 1
                elif h <= 4 or w <= 4: nw = 1
                else:
                    n88 = (1 if w <= 8 else 2) * (1 if h <= 8 else 2)
                    nw = min(n88, -(-h // 4), -(-w // 4))
                tot['win'] += nw; hist[nw] += 1
    p = tot['pairs']
    print(label, "pairs/splat %.2f contrib/pair %.1f bbox px/pair %.1f halves/pair %.2f quads/pair %.2f windows/pair %.2f hist %s" % (
        p / n, tot['contrib'] / p, tot['bboxpx'] / p, tot['halves'] / p, tot['quads'] / p, tot['win'] / p, (hist[1:5] / p).round(3)))
print("== part 1: LDS-state windows"); run(2.0, 20000, "c2-like (2.0px)")
run(1.5, 20000, "c3-like (1.5px)")

print("== part 2: register-resident 8-wide windows (ns per pair: packed half-step 114, flexible step 87)")
PK, FX = 114.0, 87.0   # ns per packed half-step / per flexible unpacked 64-px step
def run2(s_px, n=20000, label=""):
    s = s_px * np.exp(0.3 * rng.standard_normal((n, 3)))
    q = rng.standard_normal((n, 4)); q /= np.linalg.norm(q, axis=1, keepdims=True)
    x, y, z, w = q.T
    R = np.stack([1-2*(y*y+z*z), 2*(x*y-z*w), 2*(x*z+y*w), 2*(x*y+z*w), 1-2*(x*x+z*z), 2*(y*z-x*w)], 1).reshape(n, 2, 3)
    M = R * s[:, None, :]
    cov = M @ M.transpose(0, 2, 1) + 0.3 * np.eye(2)
    op = 1 / (1 + np.exp(-1.5 * rng.standard_normal(n)))
    qlim = np.minimum(9.0, 2 * np.log(np.maximum(op * 255, 1e-9)))
    keep = qlim > 0
    cov, qlim = cov[keep], qlim[keep]; n = len(qlim)
    det = cov[:, 0, 0] * cov[:, 1, 1] - cov[:, 0, 1] ** 2
    A = cov[:, 1, 1] / det; B = -cov[:, 0, 1] / det; C = cov[:, 0, 0] / det
    u = rng.uniform(32, 48, n); v = rng.uniform(32, 48, n)
    rx = np.sqrt(qlim * cov[:, 0, 0]); ry = np.sqrt(qlim * cov[:, 1, 1])
    pairs = 0; now = 0.0; flex = 0.0; kinds = np.zeros(4)
    gx = np.arange(80) + 0.5
    for i in range(n):
        dx = gx[None, :] - u[i]; dy = gx[:, None] - v[i]
        hit = A[i] * dx * dx + 2 * B[i] * dx * dy + C[i] * dy * dy <= qlim[i]
        if not hit.any(): continue
        X0 = int(np.ceil(u[i] - rx[i] - 0.5)); X1 = int(np.floor(u[i] + rx[i] - 0.5))
        Y0 = int(np.ceil(v[i] - ry[i] - 0.5)); Y1 = int(np.floor(v[i] + ry[i] - 0.5))
        for ty in range(5):
            for tx in range(5):
                blk = hit[16*ty:16*ty+16, 16*tx:16*tx+16]
                if not blk.any(): continue
                pairs += 1
                nh = int(blk[:8].any()) + int(blk[8:].any())
                bx0 = max(X0 - 16*tx, 0); bx1 = min(X1 - 16*tx, 15); by0 = max(Y0 - 16*ty, 0); by1 = min(Y1 - 16*ty, 15)
                w = bx1 - bx0 + 1; h = by1 - by0 + 1
                ncol = 1 if (bx0 >> 3) == (bx1 >> 3) else 2
                nrow = 1 if (by0 >> 3) == (by1 >> 3) else 2   # halves by bbox (>= nh by exact test)
                c_now = nh * PK
                opts = [c_now]
                if w <= 8: opts.append(nh * FX)          # X-flex window per half
                if h <= 8: opts.append(ncol * FX)        # Y-flex window per column
                best = min(opts)
                kinds[opts.index(best) if best < c_now else 0] += 1
                now += c_now; flex += best
    print(label, "pairs %d  eval now %.1f ns/pair  flex %.1f ns/pair  (-%.1f ns)  choice share packed/xflex/yflex %s" % (
        pairs, now / pairs, flex / pairs, (now - flex) / pairs, (kinds[:3] / pairs).round(3)))
run2(2.0, 20000, "c2-like")
run2(1.5, 20000, "c3-like")
