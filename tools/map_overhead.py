"""Developer diagnostic: where the host-side milliseconds of FusedMapper.optimize_map go (GPU box)."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
dev = "cuda:0"
torch.manual_seed(0); random.seed(0); np.random.seed(0)
cfg = default_config(device=dev, mapping={"seed_fraction": 0.51})
seq = SyntheticSequence(cfg, 8, 150000, seed=0)
slam = SLAM(cfg, seq)
for i in range(4):
    slam.step(i)
torch.cuda.synchronize()
g, mp = slam.gaussians, slam.mapper
def timed(label, fn, n=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); print(f"{label:40s} {(time.perf_counter() - t) / n * 1e3:8.3f} ms"); return r
snap = timed("snapshot", g.snapshot)
timed("prune (nothing pruned)", lambda: g.prune(0.005, mp.camera_extent, 100))
timed("_inline_adam", lambda: mp._inline_adam(1))
from mm3dgs_slam_amd.fused import _engine
eng = _engine(slam.renderer)
timed("check_capacity", eng.check_capacity)
color, depth, pose = seq[4]
timed("need_new_keyframe", lambda: mp.need_new_keyframe(4, slam.estimate_pose_list[3], color, depth, depth))
timed("predict_pose", lambda: slam.tracker.predict_pose(4))
import cProfile, pstats
slam.estimate_pose_list[4] = slam.estimate_pose_list[3].clone()
pr = cProfile.Profile(); pr.enable()
mp.run_frame(4, color, depth, depth); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
