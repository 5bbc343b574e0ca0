"""G9: this repository's torch-graph Tracker / Mapper / GaussianModel / Renderer against an end-to-end run of the REFERENCE's own
classes (tests/golden/make_golden_slam.py: slam/tracker.py, slam/mapper.py, slam/gaussian_model.py, slam/renderer.py driven like
slam/SLAM.py:375-493 on CPU, with the CPU oracle standing in for the absent CUDA extension on both sides), in four configurations:
the shipped method, `method: splatam`, bundle adjustment, the UTMM-style IMU configuration, a run without sensor depth, and three
renderer branches outside the shipped configs (world-frame means, Python SH with max_sh_degree 2, white background).  Pins the harness rows of
SURVEY.md 8f: RNG consumption order (keyframe picks, window subsets), keyframe decisions and covisibility graph, seeding masks and
order, densification statistics, the pruning schedule and its interplay with Adam, both optimisers, pose propagation."""
import ast
import os
import random

import numpy as np
import pytest
import torch

from oracle.raster_ref import RefRasterizer

HERE = os.path.dirname(os.path.abspath(__file__))


class _Frames:
    """The fixture's stored frames behind the sequence interface of mm3dgs_slam_amd.slam.SLAM."""

    def __init__(self, color, depth, poses, imu=None, tstamps=None):
        self.frames = [(torch.from_numpy(c), torch.from_numpy(d)) for c, d in zip(color, depth)]
        self.poses = [torch.from_numpy(p) for p in poses]
        self.imu_rows = None if imu is None else torch.from_numpy(imu)       # [frame][sample][30]: synthetic 100 Hz rows per interval
        self.tstamps = None if tstamps is None else [float(t) for t in tstamps]
        self.tf = {"c2i": torch.eye(4)}

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i][0], self.frames[i][1], self.poses[i]


@pytest.mark.parametrize("variant", ["vigs", "vigs_rotfrozen", "splatam", "ba", "imu", "estdepth", "no_transform", "sh2_python", "white_bg", "sh2_active"])
def test_torch_graph_loops_reproduce_the_reference_classes_end_to_end(variant):
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.slam import SLAM
    F = np.load(os.path.join(HERE, "golden", "g9_frames.npz"))
    G = np.load(os.path.join(HERE, "golden", f"g9_{variant}.npz"))
    overrides = ast.literal_eval(str(G["overrides"]))          # a dict literal written by the generator
    resumed_sh = bool(overrides.pop("_resumed_sh", False))     # (round 6, `sh2_active`: the reference run raised active_sh_degree like load_ply does, slam/gaussian_model.py:363)
    cfg = default_config(device="cpu", height=int(F["H"]), width=int(F["W"]), **overrides)
    n = G["est_poses"].shape[0]                                            # (the renderer-branch variants run 3 of the 5 frames)
    seq = _Frames(F["color"][:n], F["depth"][:n], F["gt_poses"][:n], F["imu"][:n], F["tstamps"][:n])
    use_imu = cfg["tracking"]["dynamics_model"].lower() == "imu"
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer, render_mode="reference", native_loops=False)
    if resumed_sh:
        slam.gaussians.active_sh_degree = slam.gaussians.max_sh_degree
    want_kf = [[int(v) for v in s.split(",")] for s in G["keyframes"]]
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
    aligned = True      # the two maps still have the same rows
    # (bundle adjustment: the window poses are Adam(eps=1e-15) parameters of the mapping loss too; the rotation about the optical axis
    #  is weakly constrained at 64x48 and its rounding noise grows ~10x per frame: 3e-7, 2e-5, 2e-4 -- hence the wider pose bar there;
    #  white background: the silhouette is 1 everywhere, Gaussians behind others get ~0 gradients whose sign Adam amplifies: 2e-5 at frame 1)
    for idx in range(len(seq)):
        color, depth, gt_pose = seq[idx]
        # the call sequence of slam/SLAM.py:375-493; with sensor depth no depth estimate is passed on, without it the tracker gets the raw
        # monocular estimate and the mapper its rescaled version (both stored in the fixture)
        e_raw, e_scaled = (None, None) if cfg["use_gt_depth"] else (torch.from_numpy(F["est"][idx]), torch.from_numpy(F["est_scaled"][idx]))
        if idx == 0:
            slam.estimate_pose_list[idx] = gt_pose.clone()
        else:
            slam.tracker.run_frame(idx, color, depth, e_raw, imu_meas=seq.imu_rows[idx].clone() if use_imu else None)
        if idx == 0:
            slam.mapper.camera_extent = float((depth if cfg["use_gt_depth"] else e_scaled).max()) / cfg["scene_radius_depth_ratio"]
        slam.mapper.run_frame(idx, color, depth, e_scaled)
        g = slam.gaussians
        # discrete decisions: identical keyframes; the map size up to the Gaussians (or seeded pixels) that sit within rounding of a
        # threshold -- two float32 programs order a few sums differently.  Measured: shipped method, frames 0-1 identical, then 1-4 of
        # ~1300 rows differ (one opacity within 1e-6 of the pruning threshold); splatam frame 1: one pixel whose rendered silhouette is
        # 0.49990 in one program and 0.50005 in the other (threshold 0.5).  While the sizes agree the state is compared tightly.
        assert [kf.idx for kf in slam.mapper.keyframes] == want_kf[idx], (idx, [kf.idx for kf in slam.mapper.keyframes], want_kf[idx])
        P_ref = int(G["per_frame"][idx, 0])
        assert abs(g._xyz.shape[0] - P_ref) <= max(2, 0.005 * P_ref), (idx, g._xyz.shape[0], P_ref)
        aligned = aligned and g._xyz.shape[0] == P_ref
        if idx == 0:
            assert aligned
        # continuous state: the camera (the norm of the raw quaternion is a direction without gradient, so compare the matrices)
        got_M, ref_M = get_camera_from_tensor(slam.estimate_pose_list[idx]), get_camera_from_tensor(torch.from_numpy(G["est_poses"][idx]))
        assert (got_M - ref_M).abs().max() < ((5e-4 if variant in ("ba", "white_bg") else 1e-4) if aligned else 1e-3), (idx, (got_M - ref_M).abs().max())
        op = torch.sigmoid(g._opacity.detach())
        with torch.no_grad():
            got = np.array([float(g._xyz.mean()), float(g._xyz.std()), float(op.mean()), float(op.std()), float(g._scaling.mean()),
                            float(g._scaling.std()), float(g._features_dc.mean()), float(g._rotation[:, 0].mean())])
        tol = 1e-4 if aligned else 5e-3
        assert np.allclose(got, G["per_frame"][idx, 1:], atol=tol, rtol=tol), (idx, got, G["per_frame"][idx, 1:])
    graph = [",".join(map(str, sorted(slam.mapper.covisibility_graph[k]))) for k in range(len(slam.mapper.keyframes))]
    assert graph == [str(s) for s in G["graph"]]
    # keyframe poses as stored at the end (bundle adjustment refines them -- or, as the reference has it, only the current one)
    for kf, ref in zip(slam.mapper.keyframes, G["keyframe_poses"]):
        d = (get_camera_from_tensor(kf.pose.detach()) - get_camera_from_tensor(torch.from_numpy(ref))).abs().max()
        assert d < 5e-4, (kf.idx, float(d))
    # the final map as a population (rows are no longer aligned once a single pruning decision differs)
    qs = torch.tensor([0.02, 0.1, 0.25, 0.5, 0.75, 0.9, 0.98])
    for name, t in (("xyz", g._xyz), ("opacity", g._opacity), ("scaling", g._scaling), ("rotation", g._rotation), ("f_dc", g._features_dc)):
        ref = torch.from_numpy(G[name])
        for col in range(t.reshape(t.shape[0], -1).shape[1]):
            a = torch.quantile(t.detach().reshape(t.shape[0], -1)[:, col], qs)
            b = torch.quantile(ref.reshape(ref.shape[0], -1)[:, col], qs)
            assert (a - b).abs().max() < 0.02 * max(1.0, float(b.abs().max())), (name, col, a, b)
    # the three RNG streams were consumed exactly as the reference consumes them
    after = np.array([random.random(), float(np.random.rand()), float(torch.rand(1))])
    assert np.allclose(after, G["rng_after"]), (after, G["rng_after"])


@pytest.mark.parametrize("variant", ["vigs", "vigs_rotfrozen", "splatam", "ba", "imu", "estdepth", "white_bg", "sh2_python", "no_transform"])
def test_native_loop_orchestration_reproduces_the_reference_classes(variant, monkeypatch):
    """The HOST side of the native loops (mm3dgs_slam_amd/fused.py: FusedTracker / FusedMapper) against the same reference
    trajectories, with tests/cpu_engine.py executing the documented semantics of the C-ABI loops on CPU (through the very structs,
    pointers and step counters fused.py builds for the library).  Same bars as the torch-graph loops above."""
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
    from mm3dgs_slam_amd.slam import SLAM
    from tests.cpu_engine import CpuEngine
    F = np.load(os.path.join(HERE, "golden", "g9_frames.npz"))
    G = np.load(os.path.join(HERE, "golden", f"g9_{variant}.npz"))
    overrides = ast.literal_eval(str(G["overrides"]))
    cfg = default_config(device="cpu", height=int(F["H"]), width=int(F["W"]), **overrides)
    n = G["est_poses"].shape[0]
    seq = _Frames(F["color"][:n], F["depth"][:n], F["gt_poses"][:n], F["imu"][:n], F["tstamps"][:n])
    use_imu = cfg["tracking"]["dynamics_model"].lower() == "imu"
    # the product's eligibility rule minus "the device is a GPU"; the engine behind the loops is the CPU stand-in
    real_eligible = fused.FusedEngine.eligible
    monkeypatch.setattr(fused.FusedEngine, "eligible", staticmethod(lambda c, g: real_eligible(dict(c, device="cuda:0"), g)))
    engines = {}
    monkeypatch.setattr(fused, "_engine", lambda renderer: engines.setdefault(id(renderer), CpuEngine(renderer)))
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer, render_mode="reference", native_loops=True)
    assert type(slam.tracker).__name__ == "FusedTracker" and type(slam.mapper).__name__ == "FusedMapper"
    want_kf = [[int(v) for v in s.split(",")] for s in G["keyframes"]]
    aligned = True
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
        assert [kf.idx for kf in slam.mapper.keyframes] == want_kf[idx], (idx, [kf.idx for kf in slam.mapper.keyframes], want_kf[idx])
        P_ref = int(G["per_frame"][idx, 0])
        assert abs(g._xyz.shape[0] - P_ref) <= max(2, 0.005 * P_ref), (idx, g._xyz.shape[0], P_ref)
        aligned = aligned and g._xyz.shape[0] == P_ref
        got_M, ref_M = get_camera_from_tensor(slam.estimate_pose_list[idx]), get_camera_from_tensor(torch.from_numpy(G["est_poses"][idx]))
        assert (got_M - ref_M).abs().max() < ((5e-4 if variant in ("ba", "white_bg") else 1e-4) if aligned else 1e-3), (idx, (got_M - ref_M).abs().max())
        with torch.no_grad():
            op = torch.sigmoid(g._opacity)
            got = np.array([float(g._xyz.mean()), float(g._xyz.std()), float(op.mean()), float(op.std()), float(g._scaling.mean()),
                            float(g._scaling.std()), float(g._features_dc.mean()), float(g._rotation[:, 0].mean())])
        tol = 1e-4 if aligned else 5e-3
        assert np.allclose(got, G["per_frame"][idx, 1:], atol=tol, rtol=tol), (idx, got, G["per_frame"][idx, 1:])
    eng = next(iter(engines.values()))
    assert any(c[0] == "track" for c in eng.calls) and any(c[0] == "map" for c in eng.calls)      # the native loops did run
    graph = [",".join(map(str, sorted(slam.mapper.covisibility_graph[k]))) for k in range(len(slam.mapper.keyframes))]
    assert graph == [str(s) for s in G["graph"]]
    for kf, ref in zip(slam.mapper.keyframes, G["keyframe_poses"]):
        assert (get_camera_from_tensor(kf.pose.detach()) - get_camera_from_tensor(torch.from_numpy(ref))).abs().max() < 5e-4, kf.idx
    after = np.array([random.random(), float(np.random.rand()), float(torch.rand(1))])
    assert np.allclose(after, G["rng_after"]), (after, G["rng_after"])
