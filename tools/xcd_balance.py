"""Developer diagnostic (GPU box): how evenly the compositors' work falls on the eight XCDs.  Runs a SLAM scenario for some frames, then renders
every view of the current mapping window and, from the per-block list lengths of each render (image_state.subcount), prints
  * the wave steps per XCD under the product's map (XCD x owns the contiguous tile span [x T/8, (x+1) T/8)): max / mean over the XCDs;
  * the same with the eight spans cut by LOAD (contiguous, equal wave steps) -- what a load-aware span table would give;
  * inside the worst XCD: max / mean over its 32 CUs with the heaviest-first dealing of binning.hip's tile_order_kernel.
python tools/xcd_balance.py [frames] [motion]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import _engine
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
motion = sys.argv[2] if len(sys.argv) > 2 else "desk"
H, W = 480, 640
cfg = default_config(device="cuda", height=H, width=W, mapping={"seed_fraction": 0.51})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0, motion=motion)
slam = SLAM(cfg, seq)
for i in range(frames):
    slam.step(i)
eng = _engine(slam.renderer)
T = ((W + 15) // 16) * ((H + 15) // 16)
up = lambda x: (x + 255) // 256 * 256
off = 256 + up(T * 4) + up((T + 1) * 4) + up(T * 4)


def loads(pose):
    with torch.no_grad():
        eng.forward(pose.detach().float().contiguous(), slam.gaussians)
    torch.cuda.synchronize()
    sub = eng.img_state[off:off + T * 16 * 4].view(torch.int32).cpu().numpy().reshape(T, 4, 4).astype(np.int64)
    return sub.max(axis=2).sum(axis=1), sub.sum()          # wave steps per tile, row steps of the render


def per_cu(span_loads):
    """heaviest-first dealing of one XCD's tiles to its 32 CU slots (serpentine), as tile_order_kernel does"""
    order = np.sort(span_loads)[::-1]
    cu = np.zeros(32)
    for r in range(0, len(order), 32):
        chunk = order[r:r + 32]
        idx = np.arange(len(chunk)) if (r // 32) % 2 == 0 else 31 - np.arange(len(chunk))
        cu[idx] += chunk
    return cu


views = [("current", slam.estimate_pose_list[frames - 1])] + [(f"kf{kf.idx}", kf.pose) for kf in slam.mapper.keyframes[-6:]]
print(f"P {slam.gaussians.get_xyz.shape[0]}  keyframes {len(slam.mapper.keyframes)}  tiles {T}")
per = T // 8
for name, pose in views:
    tw, rows = loads(pose)
    x_now = np.array([tw[x * per:(x + 1) * per].sum() for x in range(8)])
    # contiguous spans of equal load
    cs = np.cumsum(tw)
    cuts = [0] + [int(np.searchsorted(cs, cs[-1] * k / 8)) for k in range(1, 8)] + [T]
    x_bal = np.array([tw[cuts[k]:cuts[k + 1]].sum() for k in range(8)])
    worst = int(np.argmax(x_now))
    cu_now = per_cu(tw[worst * per:(worst + 1) * per])
    wb = int(np.argmax(x_bal))
    cu_bal = per_cu(tw[cuts[wb]:cuts[wb + 1]])
    mean_cu = tw.sum() / 256
    print(f"{name:8s} wave steps {tw.sum():7d} (row steps {rows})  XCD max/mean: fixed spans {x_now.max() / x_now.mean():.3f}  load-cut spans {x_bal.max() / x_bal.mean():.3f} "
          f"(span sizes {min(np.diff(cuts))}-{max(np.diff(cuts))})   busiest CU / mean CU: fixed {cu_now.max() / mean_cu:.3f}  load-cut {cu_bal.max() / mean_cu:.3f}   "
          f"heaviest tile {tw.max()} = {tw.max() / mean_cu:.2f} of a CU's mean load")
