"""Developer diagnostic (GPU box): how much of a view's binning / sorting survives from one optimiser iteration to the next -- the measurement
behind VERDICT round 5's item 1 ("stop rebuilding per iteration what did not change": validated, bit-identical reuse of a view's bins, (depth, id)
order and 4x4 block lists across the 100 tracking + 150 mapping iterations of a frame).

Runs the benchmark's SLAM (640x480, ~150 k Gaussians, full 100 + 150 budget) with the native loops issued ONE iteration per C call, and after
every iteration reads the projection stage's state of the view that was rendered (tile rectangle, view depth, pixel centre, radius of every
Gaussian: geom_state).  Two renders of the SAME view are compared -- consecutive tracking iterations of a frame (the pose moved, the map did not),
consecutive visits of a keyframe inside a mapping loop (the pose did not move, the map took `gap` Adam steps) -- for what a reuse scheme
would have to re-validate:

  rect      share of the visible Gaussians whose integer tile rectangle changed (membership of a (tile, splat) pair = the CURRENT rectangle:
            SURVEY.md Appendix A step 5, a 3-sigma rectangle cuts alpha up to 0.011 o -- it cannot be approximated)
  member    share of the tiles whose pair SET changed
  order     share of the tiles whose (depth, id) order of the pairs present in both renders changed, and how far an entry moves in its tile's list
            when it does (max / mean rank displacement over the dirty tiles)
  margin m  lists built from rectangles grown by m pixels (supersets): how many Gaussians leave their grown rectangle within the loop (a violated
            superset = rebuild), and what the superset costs (pairs / pairs of the exact rectangles)

python tools/reuse_hit_rate.py [frames=8] [motion=bounded|desk]"""
import ctypes as C
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import FusedEngine
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
motion = sys.argv[2] if len(sys.argv) > 2 else "bounded"
H, W = 480, 640
GX, GY = (W + 15) // 16, (H + 15) // 16
T = GX * GY
MARGINS = (1.0, 2.0, 4.0)
up = lambda x: (x + 255) // 256 * 256


def capture(eng):
    """(rect0, rect1, depth, px, py, radii) of the view whose projection geom_state holds (layout: csrc/mm3dgs_common.h geom_view)."""
    P = eng.P
    g = eng.geom
    splat = g[:P * 48].view(torch.float32).reshape(P, 12)
    o = up(P * 48)
    depth = g[o:o + 4 * P].view(torch.float32).clone()
    o += up(P * 4)
    rect = g[o:o + 8 * P].view(torch.int32).reshape(P, 2).clone()
    return dict(r0=rect[:, 0].long(), r1=rect[:, 1].long(), depth=depth, px=splat[:, 0].clone(), py=splat[:, 1].clone(), rad=eng.radii.clone().long(), P=P)


def rect_fields(c):
    return c["r0"] & 0xffff, c["r0"] >> 16, c["r1"] & 0xffff, c["r1"] >> 16


def pairs_of(minx, miny, maxx, maxy):
    """(gid, tile) of every pair of the rectangles, Gaussian-major."""
    w = (maxx - minx).clamp_min(0)
    area = w * (maxy - miny).clamp_min(0)
    gid = torch.repeat_interleave(torch.arange(area.numel(), device=area.device), area)
    first = torch.cumsum(area, 0) - area
    k = torch.arange(gid.numel(), device=area.device) - first[gid]
    wg = w[gid].clamp_min(1)
    tile = (miny[gid] + k // wg) * GX + minx[gid] + k % wg
    return gid, tile


def grown_rect(c, m):
    """tile rectangle of radius + m (the rule of fused.hip slam_project_vals / Appendix A step 5)."""
    rf = (c["rad"].float() + m)
    vis = c["rad"] > 0
    f = lambda v, hi: torch.clamp(torch.trunc(v), 0, hi).long()
    minx, miny = f((c["px"] - rf) / 16, GX), f((c["py"] - rf) / 16, GY)
    maxx, maxy = f((c["px"] + rf + 15) / 16, GX), f((c["py"] + rf + 15) / 16, GY)
    z = torch.zeros_like(minx)
    return torch.where(vis, minx, z), torch.where(vis, miny, z), torch.where(vis, maxx, z), torch.where(vis, maxy, z)


def compare(a, b):
    """a: earlier render, b: later render of the same view (same map size)."""
    if a["P"] != b["P"]:
        return None
    P = a["P"]
    out = {}
    vis = (a["rad"] > 0) | (b["rad"] > 0)
    changed = ((a["r0"] != b["r0"]) | (a["r1"] != b["r1"])) & vis
    out["rect"] = float(changed.sum()) / max(int(vis.sum()), 1)
    ga, ta = pairs_of(*rect_fields(a))
    gb, tb = pairs_of(*rect_fields(b))
    ka, kb = ta * P + ga, tb * P + gb
    only_a = ka[~torch.isin(ka, kb)]
    only_b = kb[~torch.isin(kb, ka)]
    dirty_m = torch.zeros(T, dtype=torch.bool, device=ka.device)
    dirty_m[only_a // P] = True
    dirty_m[only_b // P] = True
    out["member"] = float(dirty_m.sum()) / T
    out["pairs"] = int(ka.numel())
    out["pairs_flipped"] = (int(only_a.numel()) + int(only_b.numel())) / max(int(ka.numel()), 1)
    # order of the common pairs under the two depth sets: stable sorts by depth bits, then by tile (ties keep the Gaussian index: pairs are Gaussian-major)
    common = torch.isin(ka, kb)
    g, t = ga[common], ta[common]

    def order(depth):
        o1 = torch.sort(depth[g].view(torch.int32).long(), stable=True).indices
        o2 = torch.sort(t[o1], stable=True).indices
        return o1[o2]
    oa, ob = order(a["depth"]), order(b["depth"])
    diff = g[oa] != g[ob]
    dirty_o = torch.zeros(T, dtype=torch.bool, device=ka.device)
    dirty_o[t[oa][diff]] = True
    out["order"] = float(dirty_o.sum()) / T
    out["order_entries"] = float(diff.sum()) / max(int(diff.numel()), 1)
    # rank displacement: position of every common pair in the two orders
    pos_a = torch.empty_like(oa); pos_a[oa] = torch.arange(oa.numel(), device=oa.device)
    pos_b = torch.empty_like(ob); pos_b[ob] = torch.arange(ob.numel(), device=ob.device)
    disp = (pos_a - pos_b).abs()
    out["disp_max"] = int(disp.max()) if disp.numel() else 0
    out["disp_mean_moved"] = float(disp[disp > 0].float().mean()) if bool((disp > 0).any()) else 0.0
    out["dirty_any"] = float((dirty_o | dirty_m).sum()) / T
    return out


def margin_violations(anchor, later, m):
    """Gaussians whose current rectangle is not inside the rectangle grown by m pixels at `anchor`, and the superset's relative size."""
    if anchor["P"] != later["P"]:
        return None
    ax0, ay0, ax1, ay1 = grown_rect(anchor, m)
    bx0, by0, bx1, by1 = rect_fields(later)
    live = later["rad"] > 0
    viol = live & ((bx0 < ax0) | (by0 < ay0) | (bx1 > ax1) | (by1 > ay1))
    sup = ((ax1 - ax0) * (ay1 - ay0)).sum()
    ex0, ey0, ex1, ey1 = rect_fields(anchor)
    exact = ((ex1 - ex0) * (ey1 - ey0)).sum()
    tiles = torch.zeros(T, dtype=torch.bool, device=viol.device)
    if bool(viol.any()):
        gid, tile = pairs_of(torch.where(viol, bx0, 0 * bx0), torch.where(viol, by0, 0 * by0), torch.where(viol, bx1, 0 * bx1), torch.where(viol, by1, 0 * by1))
        tiles[tile] = True
    return int(viol.sum()), float(tiles.sum()) / T, float(sup) / max(float(exact), 1.0)


class Acc:
    def __init__(self):
        self.rows = []

    def add(self, d):
        if d is not None:
            self.rows.append(d)

    def mean(self, k):
        v = [r[k] for r in self.rows]
        return sum(v) / max(len(v), 1)

    def maxv(self, k):
        return max((r[k] for r in self.rows), default=0)


track_pairs, map_pairs = Acc(), Acc()
track_margin = {m: [] for m in MARGINS}     # (iterations since the anchor, violating Gaussians, dirty tile share, superset size)
map_margin = {m: [] for m in MARGINS}
map_gaps = []
state = {"record": False, "track_prev": None, "track_anchor": None, "track_it": 0, "views": {}, "map_it": 0}

orig_track, orig_map = FusedEngine.track_loop, FusedEngine.map_loop


def track_loop(self, n_iter, pose, g, lcfg, gt_color, ref, pose_adam):
    if not state["record"]:
        return orig_track(self, n_iter, pose, g, lcfg, gt_color, ref, pose_adam)
    state["track_prev"] = state["track_anchor"] = None
    for it in range(n_iter):
        orig_track(self, 1, pose, g, lcfg, gt_color, ref, pose_adam)
        cur = capture(self)
        if state["track_prev"] is not None:
            track_pairs.add(compare(state["track_prev"], cur))
        if state["track_anchor"] is None:
            state["track_anchor"] = cur
        elif it in (1, 2, 5, 10, 20, 50, 99):
            for m in MARGINS:
                r = margin_violations(state["track_anchor"], cur, m)
                if r:
                    track_margin[m].append((it,) + r)
        state["track_prev"] = cur


def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, **kw):
    if not state["record"] or map_adam is None:
        return orig_map(self, views, g, lcfg, stats, map_adam, grads, **kw)
    for i, v in enumerate(views):
        ma = type(map_adam).from_buffer_copy(map_adam)
        ma.step = map_adam.step + i
        orig_map(self, [v], g, lcfg, stats, ma, grads, **kw)
        cur = capture(self)
        key = tuple(v[0].detach().cpu().tolist())
        state["map_it"] += 1
        prev = state["views"].get(key)
        if prev is not None and prev[0]["P"] == cur["P"]:
            d = compare(prev[0], cur)
            if d is not None:
                d["gap"] = state["map_it"] - prev[1]
                map_pairs.add(d)
                map_gaps.append(d["gap"])
            anchor, a_it = prev[2], prev[3]
            if anchor["P"] == cur["P"]:
                for m in MARGINS:
                    r = margin_violations(anchor, cur, m)
                    if r:
                        map_margin[m].append((state["map_it"] - a_it,) + r)
            state["views"][key] = (cur, state["map_it"], anchor, a_it)
        else:
            state["views"][key] = (cur, state["map_it"], cur, state["map_it"])      # (first visit, or the map changed size: a pruning step re-anchors)


FusedEngine.track_loop, FusedEngine.map_loop = track_loop, map_loop

cfg = default_config(device="cuda", height=H, width=W, mapping={"seed_fraction": 0.51})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0, motion=motion)
slam = SLAM(cfg, seq)
for i in range(frames):
    state["record"] = i >= max(frames - 3, 1)       # the last three frames, one iteration per call
    state["views"].clear()
    slam.step(i)
torch.cuda.synchronize()
print(f"benchmark SLAM run, {motion} trajectory, {frames} frames, {slam.gaussians.get_xyz.shape[0]} Gaussians, {len(slam.mapper.keyframes)} keyframes; "
      f"last 3 frames recorded ({len(track_pairs.rows)} tracking pairs of renders, {len(map_pairs.rows)} mapping pairs)")
for name, acc in (("tracking: iteration i -> i + 1 of a frame (pose step, frozen map)", track_pairs), ("mapping: visit -> next visit of the same keyframe (map stepped `gap` times)", map_pairs)):
    if not acc.rows:
        continue
    print(f"{name}: pairs per render {acc.mean('pairs'):.0f}")
    if name.startswith("mapping"):
        print(f"  gap between two visits: mean {np.mean(map_gaps):.1f} iterations, median {np.median(map_gaps):.0f}, max {np.max(map_gaps)}")
    print(f"  rect    {100 * acc.mean('rect'):.3f} % of the visible Gaussians change their tile rectangle; {100 * acc.mean('pairs_flipped'):.3f} % of the pairs appear / disappear")
    print(f"  member  {100 * acc.mean('member'):.1f} % of the tiles change their pair set")
    print(f"  order   {100 * acc.mean('order'):.1f} % of the tiles change the (depth, id) order of their common pairs ({100 * acc.mean('order_entries'):.2f} % of the list positions hold another splat); "
          f"rank displacement of a moved entry: mean {acc.mean('disp_mean_moved'):.1f}, max {acc.maxv('disp_max')}")
    print(f"  either  {100 * acc.mean('dirty_any'):.1f} % of the tiles are dirty (set or order)")
for name, tab in (("tracking (iterations since the loop's first render)", track_margin), ("mapping (iterations since the view's first render after a pruning step)", map_margin)):
    print(f"superset lists, {name}:")
    for m in MARGINS:
        rows = tab[m]
        if not rows:
            continue
        its = sorted(set(r[0] for r in rows))
        buckets = its if len(its) <= 8 else [q for q in (1, 2, 5, 10, 20, 50, 100, 150) if q <= its[-1]]
        line = []
        for b in buckets:
            sel = [r for r in rows if (r[0] == b if len(its) <= 8 else (b / 2 < r[0] <= b))]
            if sel:
                line.append(f"<= {b}: {np.mean([r[1] for r in sel]):.1f} Gaussians out ({100 * np.mean([r[2] for r in sel]):.2f} % tiles)")
        print(f"  margin {m:.0f} px (superset = {np.mean([r[3] for r in rows]):.3f} x the exact pairs): " + "; ".join(line))
