"""Developer diagnostic (GPU box): the bench's `moving` scenario on its own -- per-frame wall time, map size, keyframes -- optionally with
the host phases timed (adds device syncs).   python tools/moving_run.py [--motion desk|moving|bounded] [--frames 60] [--phases]
Run it under tools/kstats_cmd.sh for the per-kernel totals of the same frames."""
import argparse, os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd import _lib, rasterizer
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence

ap = argparse.ArgumentParser()
ap.add_argument("--motion", default="desk")
ap.add_argument("--frames", type=int, default=60)
ap.add_argument("--gaussians", type=int, default=150000)
ap.add_argument("--phases", action="store_true")
ap.add_argument("--every", type=int, default=1, help="print every n-th frame")
a = ap.parse_args()
_lib.load(); rasterizer.set_binning_policy("async")
H, W = 480, 640
frac = min(1.0, a.gaussians / (0.95 * H * W))
cfg = default_config(device="cuda:0", height=H, width=W, mapping={"seed_fraction": frac})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, 3 + a.frames, a.gaussians, seed=0, motion=a.motion)
slam = SLAM(cfg, seq)
slam.step(0)
torch.manual_seed(0); random.seed(0); np.random.seed(0)
slam.step(1); slam.step(2)
acc = {}
if a.phases:
    def wrap(obj, name, label):
        fn = getattr(obj, name)
        def timed(*x, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = fn(*x, **k)
            torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
            return r
        setattr(obj, name, timed)
    for o, n in ((slam.tracker, "optimize_cam"), (slam.tracker, "run_frame"), (slam.mapper, "get_covisible_set"), (slam.mapper, "need_new_keyframe"),
                 (slam.mapper, "initialize_new_gaussians"), (slam.mapper, "add_keyframe"), (slam.mapper, "optimize_map"), (slam.mapper, "run_frame")):
        wrap(o, n, f"{type(o).__name__}.{n}")
torch.cuda.synchronize()
t_all = time.perf_counter()
kf = len(slam.mapper.keyframes)
for i in range(3, 3 + a.frames):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    slam.step(i)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    nk = len(slam.mapper.keyframes)
    if (i - 3) % a.every == 0 or nk != kf:
        eng = slam.renderer._fused_engine
        print(f"frame {i:3d} {dt * 1e3:7.2f} ms  P={slam.gaussians.get_xyz.shape[0]:7d} keyframes={nk}{' (+KF)' if nk != kf else ''}  window={len(getattr(slam.mapper, 'last_window', []) or [])} "
              f"max_tile_len={eng.max_tile_len} overflows={getattr(eng, 'overflows', 0)} reruns={getattr(slam.mapper, 'loop_reruns', 0)}", flush=True)
    kf = nk
el = time.perf_counter() - t_all
errs = slam.pose_errors()
print(f"{a.motion}: {a.frames} frames, {a.frames / el:.2f} frames/s ({el / a.frames * 1e3:.2f} ms/frame), P {slam.gaussians.get_xyz.shape[0]}, keyframes {len(slam.mapper.keyframes)}, "
      f"translation error rmse {float(np.sqrt(np.mean(np.square(errs)))) * 100:.2f} cm, final {errs[-1] * 100:.2f} cm")
for k, v in acc.items():
    print(f"  phase {k:40s} {v / a.frames * 1e3:8.2f} ms/frame")
