"""How far does a SECOND float32 program drift from the G9 reference trajectories?  This repository's torch-graph Tracker / Mapper with the
same CPU oracle rasterizer the fixtures were generated with (tests/test_golden_slam.py's setup) -- the reference's arithmetic up to the
order of a few sums -- printed per frame like tools/g9_native_check.py prints the HIP loops.  The numbers are the noise floor the bars of
tests/test_gpu_golden_slam.py are read against.     python tools/g9_cpu_check.py [--large | --shipped] [--threads N] [variant ...]
(--shipped: the g9D set, ~2750 iterations per variant: more than an hour; --threads: another OpenMP team size = other summation orders)"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle.raster_ref import RefRasterizer
from tests import g9_util
from tests.test_golden_slam import _Frames

prefix = "g9"
if "--large" in sys.argv:
    sys.argv.remove("--large"); prefix = "g9L"
if "--shipped" in sys.argv:
    sys.argv.remove("--shipped"); prefix = "g9D"
if "--threads" in sys.argv:
    i = sys.argv.index("--threads"); torch.set_num_threads(int(sys.argv[i + 1])); del sys.argv[i:i + 2]
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor as M
from mm3dgs_slam_amd.slam import SLAM
for variant in (sys.argv[1:] or ["vigs"]):
    F = g9_util.load_frames(prefix); G = g9_util.load_variant(prefix, variant)
    import ast
    overrides = ast.literal_eval(str(G["overrides"]))
    cfg = default_config(device="cpu", height=F["H"], width=F["W"], **overrides)
    n = G["est_poses"].shape[0]
    seq = _Frames(F["color"][:n], F["depth"][:n], F["gt_poses"][:n], F["imu"][:n], F["tstamps"][:n])
    use_imu = cfg["tracking"]["dynamics_model"].lower() == "imu"
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer, render_mode="reference", native_loops=False)
    for idx in range(len(seq)):
        color, depth, gt_pose = seq[idx]
        e_raw, e_scaled = (None, None) if cfg["use_gt_depth"] else (torch.from_numpy(F["est"][idx]), torch.from_numpy(F["est_scaled"][idx]))
        if idx == 0:
            slam.estimate_pose_list[idx] = gt_pose.clone()
        else:
            slam.tracker.run_frame(idx, color, depth, e_raw, imu_meas=seq.imu_rows[idx].clone() if use_imu else None)
        if idx == 0:
            slam.mapper.camera_extent = float((depth if cfg["use_gt_depth"] else e_scaled).max()) / cfg["scene_radius_depth_ratio"]
        slam.mapper.run_frame(idx, color, depth, e_scaled)
        g = slam.gaussians
        dM = float((M(slam.estimate_pose_list[idx].detach()) - M(torch.from_numpy(G["est_poses"][idx]))).abs().max())
        with torch.no_grad():
            op = torch.sigmoid(g._opacity)
            got = np.array([float(g._xyz.mean()), float(g._xyz.std()), float(op.mean()), float(op.std()), float(g._scaling.mean()),
                            float(g._scaling.std()), float(g._features_dc.mean()), float(g._rotation[:, 0].mean())])
        print(f"  cpu torch-graph {variant} frame {idx}: P {g._xyz.shape[0]} (reference {int(G['per_frame'][idx, 0])})  pose diff {dM:.2e}  "
              f"max moment diff {np.abs(got - G['per_frame'][idx, 1:]).max():.2e}", flush=True)
