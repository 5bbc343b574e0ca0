#!/usr/bin/env python
"""Entry point with the reference's command line (`python slam_top.py --config X.yml`, reference slam_top.py:30-42).

The YAML schema is the reference's (configs/TUM.yml, configs/UTMM.yml).  No dataset loader is in scope of this build
(SURVEY.md section 2), so the frames come from the in-memory synthetic RGB-D sequence (`dataset: synthetic`, the
default of `mm3dgs_slam_amd.config.default_config`); a reference config whose `dataset` is tum / utmm / replica is
accepted for its hot-path settings (iteration budgets, learning rates, pipeline flags, intrinsics) and run on the
synthetic sequence as well.  `use_gt_depth: false` (what configs/TUM.yml ships) and `tracking.dynamics_model: imu` are honoured: the
sequence provides a stand-in for the monocular estimate (`SyntheticSequence.est`, aligned per frame like slam/SLAM.py:411-448) and
synthetic IMU rows (`SyntheticSequence.imu`); only `niqe_kf` is forced off (it needs a downloaded network).

Outputs in `outputdir`, in the reference's formats (slam/SLAM.py:286-373,488-500): `point_cloud/iteration_<n>/point_cloud.ply` for
every frame index in `save_iterations` and for the final map (attribute layout of slam/gaussian_model.py:205-257), `results.npz`
with the reference's keys (pose_est, pose_gt, keyframes, ate_rmse, psnr_list, ssim_list, lpips_list [empty: LPIPS needs a downloaded
network], avg_tracking_it_time / avg_mapping_it_time with debug.get_runtime_stats).  A config that carries `iteration: <n>` resumes
from that checkpoint (map, poses, keyframes, covisibility graph), like the reference.
"""
import argparse
import os
import sys
import random
import time

import numpy as np
import torch


def seed_everything(seed=0):
    """Same sources of randomness as the reference seeds (slam_top.py:13-27)."""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=str, default=None, help="YAML config in the reference's schema")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--gaussians", type=int, default=150000, help="size of the synthetic ground-truth scene")
    args = ap.parse_args()
    from mm3dgs_slam_amd.config import default_config, load_config
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    seed_everything(0)
    cfg = default_config()
    if args.config:
        user = load_config(args.config)
        for k, v in user.items():
            if isinstance(v, dict) and isinstance(cfg.get(k), dict):
                cfg[k].update(v)
            else:
                cfg[k] = v
    cfg["mapping"]["niqe_kf"] = False
    dbg = dict(cfg.get("debug") or {})
    cfg["debug"] = {"get_runtime_stats": bool(dbg.get("get_runtime_stats", False)), "create_video": False, "save_keyframes": False}
    cfg.setdefault("outputdir", "output/synthetic")
    outdir = cfg["outputdir"]
    os.makedirs(outdir, exist_ok=True)
    seq = SyntheticSequence(cfg, args.frames, args.gaussians, seed=0)
    slam = SLAM(cfg, seq)
    times, t_last = [], [time.perf_counter()]

    def progress(i):
        torch.cuda.synchronize()
        now = time.perf_counter()
        times.append(now - t_last[0]); t_last[0] = now
        err = slam.pose_errors()[-1]
        print(f"frame {i:4d}  {times[-1] * 1e3:8.1f} ms  gaussians {slam.gaussians.get_xyz.shape[0]:7d}  pose error {err * 100:.2f} cm")

    slam.run(progress, reraise=False)          # (the reference's behaviour on a failed frame: print, save, carry on -- reported by the exit code below); checkpoints, the final map and results.npz are written inside (reference formats)
    res = np.load(os.path.join(outdir, "results.npz"), allow_pickle=True)
    print(f"Average Trajectory Error RMSE: {float(res['ate_rmse'])} m; {1.0 / np.mean(times[1:]):.2f} frames/s after frame 0; outputs in {outdir}")
    if slam.failure is not None:      # (the reference prints the exception and saves what it has, slam/SLAM.py:494-503; the exit code says so too)
        sys.exit(1)


if __name__ == "__main__":
    main()
