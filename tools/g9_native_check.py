"""Next-round check (GPU box; written when no GPU time was left, so it is NOT yet part of the test suite): the NATIVE loops
(FusedTracker / FusedMapper over the HIP kernels) on the frames of the G9 fixtures, against the trajectories the reference's own
Tracker / Mapper classes produced with the CPU oracle as their rasterizer (tests/golden/make_golden_slam.py).  Expected, from what
the torch-graph loops show on CPU (tests/test_golden_slam.py): identical keyframes and RNG state; map size within ~0.5 %; poses to
~1e-4 while no threshold decision has flipped, ~1e-3 afterwards.  Promote to tests/test_gpu_golden_slam.py once it has run.

    python tools/g9_native_check.py [variant ...]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor as M
from mm3dgs_slam_amd.slam import SLAM

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"


class Frames:
    def __init__(self, F):
        self.frames = [(torch.from_numpy(c).to(DEV), torch.from_numpy(d).to(DEV)) for c, d in zip(F["color"], F["depth"])]
        self.poses = [torch.from_numpy(p).to(DEV) for p in F["gt_poses"]]
        self.imu_rows = torch.from_numpy(F["imu"])
        self.tstamps = [float(t) for t in F["tstamps"]]
        self.tf = {"c2i": torch.eye(4)}

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i][0], self.frames[i][1], self.poses[i]


def main():
    F = np.load(os.path.join(ROOT, "tests", "golden", "g9_frames.npz"))
    for variant in (sys.argv[1:] or ["vigs", "splatam", "ba", "imu", "estdepth"]):
        G = np.load(os.path.join(ROOT, "tests", "golden", f"g9_{variant}.npz"))
        overrides = eval(str(G["overrides"]), {"__builtins__": {}})
        cfg = default_config(device=DEV, height=int(F["H"]), width=int(F["W"]), **overrides)
        seq = Frames(F)
        use_imu = cfg["tracking"]["dynamics_model"].lower() == "imu"
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        slam = SLAM(cfg, seq)                               # native loops on the HIP rasterizer
        print(f"== {variant}: native tracker {type(slam.tracker).__name__}, mapper {type(slam.mapper).__name__}")
        for idx in range(len(seq)):
            color, depth, gt_pose = seq[idx]
            e_raw, e_scaled = (None, None) if cfg["use_gt_depth"] else (torch.from_numpy(F["est"][idx]).to(DEV), torch.from_numpy(F["est_scaled"][idx]).to(DEV))
            if idx == 0:
                slam.estimate_pose_list[idx] = gt_pose.clone()
            else:
                slam.tracker.run_frame(idx, color, depth, e_raw, imu_meas=seq.imu_rows[idx].clone() if use_imu else None)
            if idx == 0:
                slam.mapper.camera_extent = float((depth if cfg["use_gt_depth"] else e_scaled).max()) / cfg["scene_radius_depth_ratio"]
            slam.mapper.run_frame(idx, color, depth, e_scaled)
            g = slam.gaussians
            dM = float((M(slam.estimate_pose_list[idx].cpu()) - M(torch.from_numpy(G["est_poses"][idx]))).abs().max())
            op = torch.sigmoid(g._opacity.detach())
            got = np.array([float(g._xyz.mean()), float(g._xyz.std()), float(op.mean()), float(op.std()), float(g._scaling.mean()),
                            float(g._scaling.std()), float(g._features_dc.mean()), float(g._rotation[:, 0].mean())])
            print(f"  frame {idx}: keyframes {[kf.idx for kf in slam.mapper.keyframes]} (reference {G['keyframes'][idx]})  P {g._xyz.shape[0]} "
                  f"(reference {int(G['per_frame'][idx, 0])})  pose diff {dM:.2e}  max moment diff {np.abs(got - G['per_frame'][idx, 1:]).max():.2e}")
        after = np.array([random.random(), float(np.random.rand()), float(torch.rand(1))])
        print("  RNG streams end equal:", bool(np.allclose(after, G["rng_after"])), " overflows re-run:", getattr(slam.mapper, "loop_reruns", 0))


if __name__ == "__main__":
    main()
