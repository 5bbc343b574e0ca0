"""Developer diagnostic (GPU box): how the projection + binning kernel's emission work is spread over its lanes / waves / workgroups.
Every lane emits the first four (tile, splat) pairs of its own splat; the pairs beyond the fourth of a wave's 64 splats form the wave's
flat work list, cut into 64 equal shares (csrc/fused.hip slam_bin_pairs).    python tools/bin_balance.py [frames] [motion]"""
import os, sys, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import _engine
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 8
motion = sys.argv[2] if len(sys.argv) > 2 else "bounded"
H, W = 480, 640
cfg = default_config(device="cuda", height=H, width=W, mapping={"seed_fraction": 0.51})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0, motion=motion)
slam = SLAM(cfg, seq)
for i in range(frames):
    slam.step(i)
eng = _engine(slam.renderer)
with torch.no_grad():
    eng.forward(slam.estimate_pose_list[frames - 1].detach().float().contiguous(), slam.gaussians)
torch.cuda.synchronize()
P = int(slam.gaussians.get_xyz.shape[0])
up = lambda x: (x + 255) // 256 * 256
off = up(P * 48) + up(P * 4)
rect = eng.geom[off:off + 8 * P].view(torch.int32).cpu().numpy().astype(np.int64).reshape(P, 2)
radii = eng.radii[:P].cpu().numpy()
w = (rect[:, 1] & 0xffff) - (rect[:, 0] & 0xffff)
h = (rect[:, 1] >> 16) - (rect[:, 0] >> 16)
area = np.where(radii > 0, np.maximum(w, 0) * np.maximum(h, 0), 0)
pad = (-P) % 256
a = np.concatenate([area, np.zeros(pad, dtype=np.int64)])
own, extra = np.minimum(a, 4), np.maximum(a - 4, 0)
wave_extra = extra.reshape(-1, 64).sum(1)
wave_steps = own.reshape(-1, 64).max(1) + (wave_extra + 63) // 64        # emission steps of a wave: its lanes' own pairs, then the shared list
wg_steps = wave_steps.reshape(-1, 4).max(1)
wg_pairs = a.reshape(-1, 256).sum(1)
print(f"P {P}  visible {int((area > 0).sum())}  pairs {int(a.sum())}  pairs beyond a splat's fourth {int(extra.sum())} ({extra.sum() / max(a.sum(), 1) * 100:.1f} %)")
print(f"tiles per visible splat: mean {area[area > 0].mean():.2f} p90 {np.percentile(area[area > 0], 90):.0f} p99 {np.percentile(area[area > 0], 99):.0f} max {area.max()};  splats of > 4 tiles {np.mean(area[area > 0] > 4) * 100:.1f} %, > 16 tiles {np.mean(area[area > 0] > 16) * 100:.2f} %")
for name, v in (("extra pairs per wave", wave_extra), ("emission steps per wave (model)", wave_steps), ("emission steps per workgroup (max of its waves)", wg_steps), ("pairs per workgroup", wg_pairs)):
    print(f"{name}: mean {v.mean():.1f} p50 {np.percentile(v, 50):.0f} p90 {np.percentile(v, 90):.0f} p99 {np.percentile(v, 99):.0f} max {v.max()}")
