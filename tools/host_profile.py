"""Developer diagnostic (GPU box): where the HOST spends its time in a SLAM frame (cProfile over 10 frames after warm-up)."""
import cProfile, os, pstats, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd import _lib, rasterizer
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
_lib.load(); rasterizer.set_binning_policy("async")
torch.manual_seed(0); random.seed(0); np.random.seed(0)
cfg = default_config(device="cuda:0", height=480, width=640, mapping={"seed_fraction": 0.51})
seq = SyntheticSequence(cfg, 16, 150000, seed=0)
slam = SLAM(cfg, seq)
for i in range(4):
    slam.step(i)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(4, 14):
    slam.step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
