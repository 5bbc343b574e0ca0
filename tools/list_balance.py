"""How evenly is the compositors' work spread?  Runs the bench's SLAM scene for a few frames, reads the per-block list lengths of the last
render (image_state.subcount: sixteen 4x4-block lists per tile) and prints
  * row steps (sum of list lengths) against wave steps (a wave's four rows advance together: max of its four lists): the lanes' lockstep loss;
  * the same if every wave got four lists of similar length (sorted grouping) -- what a length-aware wave assembly could win;
  * the spread of work per tile, per wave and per SIMD under a round-robin placement (wave w of workgroup g on SIMD w of CU g % 256).
GPU box:  python tools/list_balance.py [frames]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import _engine
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
H, W = 480, 640
cfg = default_config(device="cuda", height=H, width=W, tracking={"iters": 100}, mapping={"iters": 150, "seed_fraction": 0.51})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0)
slam = SLAM(cfg, seq)
for i in range(frames):
    slam.step(i)
eng = _engine(slam.renderer)
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
up = lambda x: (x + 255) // 256 * 256
off = 256 + up(T * 4) + up((T + 1) * 4) + up(T * 4)
sub = eng.img_state[off:off + T * 16 * 4].view(torch.int32).cpu().numpy().reshape(T, 4, 4).astype(np.int64)      # [tile][wave = 8x8 sub-tile][block]
rows = sub.sum()
wave = sub.max(axis=2)                       # [T, 4] steps of each wave
print(f"P {slam.gaussians._xyz.shape[0]}  tiles {T}  row steps {rows}  mean list {sub.mean():.1f}  max list {sub.max()}")
print(f"wave steps (max of 4 rows) {wave.sum()}  = {wave.sum() * 4 / rows:.3f} x the row steps / 4")
flat = np.sort(sub.reshape(-1))
grouped = flat.reshape(-1, 4).max(axis=1).sum()
print(f"  with four lists of similar length per wave (global sort): {grouped * 4 / rows:.3f} x")
tile_sorted = np.sort(sub.reshape(T, 16), axis=1).reshape(T, 4, 4).max(axis=2).sum()
print(f"  regrouping only inside a tile: {tile_sorted * 4 / rows:.3f} x")
tw = wave.sum(axis=1)
print(f"per-tile wave steps: mean {tw.mean():.0f}  p95 {np.percentile(tw, 95):.0f}  max {tw.max()}   per-wave: mean {wave.mean():.0f} p95 {np.percentile(wave, 95):.0f} max {wave.max()}")
# round-robin placement model: workgroup g -> CU g % 256 (five slots), wave w -> SIMD w
for order_name, order in (("launch order", np.arange(T)), ("heaviest first", np.argsort(-tw))):
    simd = np.zeros((256, 4))
    for slot, t in enumerate(order):
        simd[slot % 256] += wave[t]
    print(f"SIMD load, {order_name}: mean {simd.mean():.0f}  max {simd.max():.0f}  max/mean {simd.max() / simd.mean():.3f}")
# greedy (LPT) onto the CU with the least load so far, five workgroups per CU at most
load, cnt = np.zeros(256), np.zeros(256, dtype=int)
for t in np.argsort(-tw):
    free = np.where(cnt < 5)[0]
    c = free[np.argmin(load[free])]
    load[c] += tw[t]; cnt[c] += 1
print(f"CU load (sum of its tiles' wave steps), greedy heaviest-first: max/mean {load.max() / load.mean():.3f};  launch order: "
      f"{np.array([tw[np.arange(T)[c::256]].sum() for c in range(256)]).max() / (tw.sum() / 256):.3f}")

# ---- round 5: the SIMD level.  A launch ends when its busiest SIMD ends; SIMD s of a CU runs wave s of each of the CU's workgroups (if the
# placement is wave w -> SIMD (w + c) % 4: tools/ubench/simd_placement.hip), i.e. ONE 8x8 sub-tile of each of its ~5 tiles.  Model the
# product's placement (workgroup b -> XCD b % 8, the k-th workgroup of an XCD -> CU k % 32; tile -> workgroup through the load-balanced table,
# modelled here as heaviest-first dealing) and ask what a per-tile ROTATION of the sub-tiles over the waves (wave w composites sub-tile
# (w + r) % 4: two bits per table entry) could win: greedy, tile by tile, the rotation that minimises the CU's busiest SIMD so far.
def simd_model(order_of_cu):
    ident, rot = [], []
    for tiles in order_of_cu:
        s0 = np.zeros(4); s1 = np.zeros(4)
        for t in tiles:
            s0 += wave[t]
            best = min(range(4), key=lambda r: (s1 + np.roll(wave[t], r)).max())
            s1 += np.roll(wave[t], best)
        ident.append(s0); rot.append(s1)
    return np.array(ident), np.array(rot)

per = (T + 7) // 8
cus = []
for x in range(8):
    span = list(range(x * per, min(T, (x + 1) * per)))
    span.sort(key=lambda t: -tw[t])
    # serpentine dealing of the heaviest-first list over the XCD's 32 CUs
    buckets = [[] for _ in range(32)]
    for i, t in enumerate(span):
        r, c = divmod(i, 32)
        buckets[c if r % 2 == 0 else 31 - c].append(t)
    cus += buckets
ident, rot = simd_model(cus)
mean = wave.sum() / 1024.0
print(f"busiest SIMD / mean SIMD (wave steps; mean {mean:.0f}): sub-tile = wave {ident.max() / mean:.3f};  with a per-tile rotation {rot.max() / mean:.3f};  "
      f"busiest CU / mean CU {ident.sum(axis=1).max() / (4 * mean):.3f}")
print(f"  a tile's four waves: mean of (max / mean over its sub-tiles) {np.mean(wave.max(axis=1) / np.maximum(wave.mean(axis=1), 1)):.3f}")
