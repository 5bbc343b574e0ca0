"""Developer diagnostic (GPU box): how many (4x4 block, splat) pairs does the block-mask rule of csrc/tile_mask.h list on the map of
a short SLAM run, against (a) the exact ellipse-vs-block test and (b) the blocks that hold at least one pixel with alpha >= 1/255?
The compositors' work is proportional to the listed pairs, so the ratios bound what a tighter rule could buy.  Also prints the
anisotropy of the projected splats (the current rule is exact for isotropic ones).   usage: python tools/mask_rule_probe.py [frames]"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda:0"
torch.manual_seed(0); random.seed(0); np.random.seed(0)
H, W = 480, 640
cfg = default_config(device=dev, height=H, width=W, mapping={"seed_fraction": 150000 / (0.95 * H * W)})
slam = SLAM(cfg, SyntheticSequence(cfg, frames + 1, 150000, seed=0))
for i in range(frames):
    slam.step(i)
g = slam.gaussians
pose = slam.estimate_pose_list[frames - 1]
with torch.no_grad():
    w2c = get_camera_from_tensor(pose).to(dev)
    Rm, t = w2c[:3, :3], w2c[:3, 3]
    xyz = g.get_xyz
    p = xyz @ Rm.T + t
    vis = p[:, 2] > 0.2
    c = cfg["cam"]
    fx, fy, cx, cy = c["fx"], c["fy"], c["cx"], c["cy"]
    s = g.get_scaling
    if s.shape[1] == 1:
        s = s.repeat(1, 3)
    q = torch.nn.functional.normalize(g.get_rotation)
    r, x, y, z = q.unbind(1)
    Rq = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                      2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                      2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(-1, 3, 3)
    M = Rq * s[:, None, :]
    Sig = M @ M.transpose(1, 2)
    tz = p[:, 2]
    J = torch.zeros(p.shape[0], 2, 3, device=dev)
    J[:, 0, 0] = fx / tz; J[:, 0, 2] = -fx * p[:, 0] / tz ** 2
    J[:, 1, 1] = fy / tz; J[:, 1, 2] = -fy * p[:, 1] / tz ** 2
    A = J @ Rm
    cov = A @ Sig @ A.transpose(1, 2)
    a, b, cc = cov[:, 0, 0] + 0.3, cov[:, 0, 1], cov[:, 1, 1] + 0.3
    det = a * cc - b * b
    qa, qb, qc = cc / det, -b / det, a / det
    op = g.get_opacity[:, 0]
    tau = torch.log(255.0 * op)
    mx, my = fx * p[:, 0] / tz + cx - 0.5, fy * p[:, 1] / tz + cy - 0.5
    keep = vis & (tau > 0) & (det > 0) & (mx > -50) & (mx < W + 50) & (my > -50) & (my < H + 50)
    mid = 0.5 * (a + cc)
    lam1 = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0)); lam2 = det / lam1
    ratio = torch.sqrt(lam1 / lam2)[keep]
    print(f"Gaussians {xyz.shape[0]}, with a visible alpha >= 1/255 region {int(keep.sum())}")
    print("axis ratio of the projected splats p50/p90/p99:", [round(float(v), 2) for v in torch.quantile(ratio[:200000], torch.tensor([0.5, 0.9, 0.99], device=dev))])
    rad = torch.sqrt(2 * tau * lam1)[keep]
    print("alpha >= 1/255 radius (major axis, px) p50/p90/p99:", [round(float(v), 1) for v in torch.quantile(rad[:200000], torch.tensor([0.5, 0.9, 0.99], device=dev))])
    idx = torch.nonzero(keep)[:, 0]
    idx = idx[torch.randperm(idx.numel(), device=dev)[:40000]]
    mx, my, qa, qb, qc, tau, a, cc, lam1 = (v[idx] for v in (mx, my, qa, qb, qc, tau, a, cc, lam1))
    t2 = 2 * tau
    hx, hy, r2 = torch.sqrt(t2 * a), torch.sqrt(t2 * cc), t2 * lam1
    small = torch.sqrt(r2) < 40          # (the few huge splats would need a bigger candidate window; they are listed separately)
    print("sampled", idx.numel(), "of which radius < 40 px:", int(small.sum()))
    Rb = 11
    off = torch.arange(-Rb, Rb + 1, device=dev)
    bx0 = torch.floor(mx / 4).long()[:, None, None] + off[None, None, :]
    by0 = torch.floor(my / 4).long()[:, None, None] + off[None, :, None]
    bx0, by0 = bx0.expand(-1, 2 * Rb + 1, 2 * Rb + 1), by0.expand(-1, 2 * Rb + 1, 2 * Rb + 1)
    inimg = (bx0 >= 0) & (bx0 < W // 4) & (by0 >= 0) & (by0 < H // 4) & small[:, None, None]
    lox, loy = 4.0 * bx0, 4.0 * by0
    hix, hiy = lox + 3, loy + 3
    cxx, cyy = mx[:, None, None], my[:, None, None]
    # current rule: bounding box overlap AND distance to the rectangle within the disc of the major axis
    box = (cxx - hx[:, None, None] <= hix) & (cxx + hx[:, None, None] >= lox) & (cyy - hy[:, None, None] <= hiy) & (cyy + hy[:, None, None] >= loy)
    dx = torch.clamp(torch.maximum(lox - cxx, cxx - hix), min=0); dy = torch.clamp(torch.maximum(loy - cyy, cyy - hiy), min=0)
    cur = box & (dx * dx + dy * dy <= r2[:, None, None]) & inimg
    # (b) blocks with at least one pixel centre inside the ellipse
    A_, B_, C_ = qa[:, None, None], qb[:, None, None], qc[:, None, None]
    anypx = torch.zeros_like(cur)
    for py in range(4):
        for px in range(4):
            ddx, ddy = lox + px - cxx, loy + py - cyy
            anypx |= (A_ * ddx * ddx + 2 * B_ * ddx * ddy + C_ * ddy * ddy) <= t2[:, None, None]
    anypx &= inimg
    # (a) exact continuous test: minimum of the quadratic form over the rectangle (interior, else the four edges)
    inside = (cxx >= lox) & (cxx <= hix) & (cyy >= loy) & (cyy <= hiy)
    best = torch.full_like(lox, float("inf"))
    for ex in (lox, hix):          # vertical edges: x fixed, minimise over y in [loy, hiy]
        ddx = ex - cxx
        yopt = torch.clamp(cyy - B_ * ddx / C_, loy, hiy)
        ddy = yopt - cyy
        best = torch.minimum(best, A_ * ddx * ddx + 2 * B_ * ddx * ddy + C_ * ddy * ddy)
    for ey in (loy, hiy):
        ddy = ey - cyy
        xopt = torch.clamp(cxx - B_ * ddy / A_, lox, hix)
        ddx = xopt - cxx
        best = torch.minimum(best, A_ * ddx * ddx + 2 * B_ * ddx * ddy + C_ * ddy * ddy)
    exact = (inside | (best <= t2[:, None, None])) & inimg
    n_cur, n_ex, n_px = int(cur.sum()), int(exact.sum()), int(anypx.sum())
    assert int((exact & ~cur).sum()) == 0 and int((anypx & ~exact).sum()) == 0
    print(f"listed (block, splat) pairs per splat: current rule {n_cur / int(small.sum()):.2f}, exact ellipse-vs-block {n_ex / int(small.sum()):.2f} "
          f"({n_ex / n_cur:.3f} x), blocks with a pixel inside {n_px / int(small.sum()):.2f} ({n_px / n_cur:.3f} x)")
    # useful lanes: pixels with alpha >= 1/255 per listed block
    npx = torch.zeros_like(lox)
    for py in range(4):
        for px in range(4):
            ddx, ddy = lox + px - cxx, loy + py - cyy
            npx += ((A_ * ddx * ddx + 2 * B_ * ddx * ddy + C_ * ddy * ddy) <= t2[:, None, None]).float()
    print(f"pixels inside per listed block: current {float((npx * cur).sum()) / n_cur:.2f} / 16, exact {float((npx * exact).sum()) / n_ex:.2f} / 16")
    # ---- finer cells (next round's plan: 4x2-pixel cells walked by 8-lane rows): cells with a pixel inside, per splat, and their occupancy
    for cw, ch in ((4, 4), (4, 2), (2, 4), (8, 1), (2, 2), (8, 2)):
        Rc = (48 // cw + 1, 48 // ch + 1)
        ox = torch.arange(-Rc[0], Rc[0] + 1, device=dev); oy = torch.arange(-Rc[1], Rc[1] + 1, device=dev)
        cx0 = (torch.floor(mx / cw).long()[:, None, None] + ox[None, None, :]).expand(-1, oy.numel(), ox.numel())
        cy0 = (torch.floor(my / ch).long()[:, None, None] + oy[None, :, None]).expand(-1, oy.numel(), ox.numel())
        ok = (cx0 >= 0) & (cx0 < W // cw) & (cy0 >= 0) & (cy0 < H // ch) & small[:, None, None]
        cnt = torch.zeros(cx0.shape, device=dev)
        for py in range(ch):
            for px in range(cw):
                ddx, ddy = cw * cx0 + px - cxx, ch * cy0 + py - cyy
                cnt += ((A_ * ddx * ddx + 2 * B_ * ddx * ddy + C_ * ddy * ddy) <= t2[:, None, None]).float()
        cnt = cnt * ok
        ncell = int((cnt > 0).sum())
        lanes = 64 // (cw * ch)
        print(f"cells {cw}x{ch}: {ncell / int(small.sum()):6.2f} cells per splat with a pixel inside, {float(cnt.sum()) / ncell:5.2f} / {cw * ch} pixels inside; "
              f"wave steps per splat (64 lanes = {lanes} cells) {ncell / int(small.sum()) / lanes:.3f}")
