"""G9 on the GPU: the NATIVE loops (FusedTracker / FusedMapper over mm3dgs_slam_track / mm3dgs_slam_map, i.e. the HIP kernels) on
the frames of the G9 fixtures, against the trajectories the reference's own Tracker / Mapper / GaussianModel / Renderer classes
produced (tests/golden/make_golden_slam.py: slam/tracker.py:94-177, slam/mapper.py:718-950, slam/renderer.py:80-83,207-214 driven
like slam/SLAM.py:375-493 with the CPU oracle standing in for the absent CUDA extension).  Every native-eligible variant: the
shipped method, `method: splatam`, bundle adjustment, the UTMM-style IMU configuration, no sensor depth, white background.

Bars (the same as the CPU runs of tests/test_golden_slam.py hold for the torch-graph loops): identical keyframe lists, covisibility
graph and RNG end state; map size within 0.5 %; while no threshold decision has flipped (the maps still have the same rows) camera
matrices to 1e-4 (5e-4 for bundle adjustment / white background, whose weakly constrained directions amplify rounding ~10x per
frame on the reference side as well) and every map moment to 1e-4; afterwards 1e-3 / 5e-3."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
DEV = "cuda:0"


class _Frames:
    def __init__(self, F, n):
        self.frames = [(torch.from_numpy(np.ascontiguousarray(c)).to(DEV), torch.from_numpy(np.ascontiguousarray(d)).to(DEV)) for c, d in zip(F["color"][:n], F["depth"][:n])]
        self.poses = [torch.from_numpy(p).to(DEV) for p in F["gt_poses"][:n]]
        self.imu_rows = torch.from_numpy(F["imu"][:n])
        self.tstamps = [float(t) for t in F["tstamps"][:n]]
        self.tf = {"c2i": torch.eye(4)}

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        return self.frames[i][0], self.frames[i][1], self.poses[i]


def run_variant(variant, verbose=False, prefix="g9"):
    """Drives the native loops over the fixture's frames; returns the per-frame measurements (also used by tools/g9_native_check.py).
    prefix "g9": the 64x48 fixtures; "g9L": the 160x120 ones (80 tiles, ~8.6 k Gaussians, 8 frames, keyframes 0 / 2 / 4 / 6); "g9D": 160x120 at
    the SHIPPED schedule (configs/TUM.yml: 100 tracking / 150 mapping iterations, pruning_interval 50, min_opacity 0.005, kf_every 5,
    min_covisibility 0.95) on the hand-held sweep of the bench's `moving` line: 11 frames, keyframes 0 / 5 / 10, 19.2 k Gaussians + what the
    keyframes seed."""
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor as M
    from mm3dgs_slam_amd.slam import SLAM
    from tests import g9_util
    F = g9_util.load_frames(prefix)
    G = g9_util.load_variant(prefix, variant)
    import ast
    overrides = ast.literal_eval(str(G["overrides"]))          # a dict literal written by the generator
    resumed_sh = bool(overrides.pop("_resumed_sh", False))      # (the fixture's own key: the reference run raised active_sh_degree like load_ply does)
    cfg = default_config(device=DEV, height=int(F["H"]), width=int(F["W"]), **overrides)
    n = G["est_poses"].shape[0]
    seq = _Frames(F, n)
    use_imu = cfg["tracking"]["dynamics_model"].lower() == "imu"
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    slam = SLAM(cfg, seq)                               # default: native loops on the HIP library
    if resumed_sh:
        slam.gaussians.active_sh_degree = slam.gaussians.max_sh_degree      # slam/gaussian_model.py:363
    assert type(slam.tracker).__name__ == "FusedTracker" and type(slam.mapper).__name__ == "FusedMapper"
    from mm3dgs_slam_amd.fused import FusedEngine
    assert FusedEngine.eligible(cfg, slam.gaussians), "this variant must run on the native loops"
    rows = []
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
        dM = float((M(slam.estimate_pose_list[idx].detach().cpu().float()) - M(torch.from_numpy(G["est_poses"][idx]))).abs().max())
        with torch.no_grad():
            op = torch.sigmoid(g._opacity)
            got = np.array([float(g._xyz.mean()), float(g._xyz.std()), float(op.mean()), float(op.std()), float(g._scaling.mean()),
                            float(g._scaling.std()), float(g._features_dc.mean()), float(g._rotation[:, 0].mean())])
        rows.append(dict(idx=idx, keyframes=[kf.idx for kf in slam.mapper.keyframes], P=int(g._xyz.shape[0]), P_ref=int(G["per_frame"][idx, 0]),
                         pose_diff=dM, moments=got, moments_ref=G["per_frame"][idx, 1:]))
        if verbose:
            print(f"  {variant} frame {idx}: keyframes {rows[-1]['keyframes']} (reference {G['keyframes'][idx]})  P {rows[-1]['P']} (reference {rows[-1]['P_ref']})  "
                  f"pose diff {dM:.2e}  max moment diff {np.abs(got - G['per_frame'][idx, 1:]).max():.2e}")
    return slam, G, rows


# What this test found (round 3).  The reference seeds exactly isotropic Gaussians with identity rotation (slam/mapper.py:644-668).  Their
# rotation gradient is 2 s^2 (dSigma[j][k] - dSigma[k][j]): zero in exact arithmetic, and exactly zero in torch, whose autograd of
# Sigma = L L^T forms (dSigma + dSigma^T) L.  The kernels computed the two triangles of dSigma separately; their float32 rounding
# difference is noise that Adam(eps=1e-15) (slam/gaussian_model.py:143-195) turns into full +-lr steps of the quaternions, a random walk
# the reference does not take: poses drifted to 1.4e-5 after the first tracked frame and 4e-4 by frame 4, the mean quaternion w by 3e-4
# within frame 0 (with the rotation learning rate at 0 -- the `vigs_rotfrozen` fixture -- the drift was gone: 2e-8).  With dSigma averaged
# over its triangles (fused.hip / preprocess.hip) the HIP loops follow the reference's own classes to 1e-8 .. 5e-7 in the camera matrices
# and 6e-6 in every map moment for as long as no threshold decision flips (measured: vigs 1.2e-8, 6.8e-8, 5.2e-7, 1.8e-5 over frames
# 1-4; imu 5e-9 .. 2e-5; splatam 9e-7 at frame 1).  Bundle adjustment stays loose from frame 3 on: at 64x48 the rotation about the
# optical axis is barely constrained, and the reference's own arithmetic re-run by the torch-graph loops on CPU drifts from it by 6e-3 there.


# Round 4: the same runs at 160x120 (g9L_*: 80 tiles, 6.4 - 17 k Gaussians, 8 frames, keyframes 0 / 2 / 4 / 6; every native variant incl. the
# two that became native this round, sh2_python and no_transform).  At this size threshold decisions flip from the first mapping loop on
# (18 k seeded Gaussians pruned at opacity 0.4625: a handful always sit within rounding of the threshold), so "the maps still have the
# same rows" rarely lasts.  What two float32 programs can agree on here was measured by re-running the REFERENCE-side arithmetic itself --
# this repository's torch-graph loops over the same CPU oracle, 7 instead of 6 / 8 OpenMP threads, i.e. other summation orders
# (tools/g9_cpu_check.py --large, profiles/r04_g9L_cpu_float32_floor.txt): map size off by 32 - 74 of 8.5 k (0.4 - 0.9 %) from frame 0 on,
# camera matrices 2e-4 .. 1.5e-3 (bundle adjustment 6e-3), map moments up to 8.5e-3.  The HIP loops stay INSIDE that floor: map size within
# 28 (0.3 %), cameras 3e-6 .. 7e-5 while the rows agree and <= 2.7e-3 afterwards (bundle adjustment 6e-3 at frame 7), moments <= 3e-3;
# keyframes, covisibility graph and the three RNG streams identical in all nine variants.
# Round 5 ("what the test would let through is 2 - 50x what was measured"): the g9L bars are now PER FRAME, 3x what
# profiles/r04_g9L_hip_loops.txt measured for that variant and frame (camera-matrix difference | largest moment difference; floors 1e-6 /
# 1e-5) -- the native loops are deterministic (no atomics decide a value), so a run either reproduces those numbers or something changed.
# Round 6 (bars at 3x ONE run flake the moment a kernel rounds differently -- Adam(eps 1e-15) carries a last-bit difference
# forward over ~2750 steps): the tracking compositor's pose chain and the two-phase backward reduction change summation orders, and single
# frames moved by up to 7x (g9L imu, camera) / 4.5x (g9D tum) against the round-4 / round-5 runs while the bulk stayed within 1.2x.  The
# tables now hold, per variant and frame, the MAXIMUM over an ensemble of float32 programs -- the round-4 / round-5 kernels
# (profiles/r04_g9L_hip_loops.txt, r05_g9D_hip_loops.txt), this round's kernels (profiles/r06_g9L_hip_loops.txt, r06_g9D_hip_loops.txt) and this
# round's kernels on the gradient-record path (MM3DGS_NO_POSE_CHAIN=1: *_record_path.txt), and a fourth member, the final kernels with the
# two-term covariance chain of fused.hip (*_two_term_chain.txt) -- and the bars stay 3x that, capped by the old
# global bars (camera 5e-3 / 1e-2 for bundle adjustment, moments 1e-2).  Every member sits inside the float32 floor measured on the
# reference's own arithmetic (profiles/r04_g9L_cpu_float32_floor.txt, r05_g9D_cpu_float32_floor.txt).
G9L_MEASURED = {
    "vigs": [(9.31e-10, 3.75e-06), (1.09e-05, 5.13e-06), (2.53e-05, 2.93e-04), (4.89e-05, 2.11e-04), (1.65e-04, 3.76e-04), (2.87e-04, 3.90e-04), (4.57e-04, 7.96e-04), (2.07e-04, 1.42e-03)],
    "vigs_rotfrozen": [(9.31e-10, 3.81e-06), (1.04e-05, 2.62e-06), (4.02e-05, 5.53e-04), (9.06e-05, 5.34e-04), (2.94e-04, 5.04e-04), (3.84e-04, 6.82e-04), (4.92e-04, 2.70e-03), (3.58e-04, 3.13e-03)],
    "splatam": [(9.31e-10, 1.68e-06), (3.70e-06, 4.69e-05), (1.01e-05, 1.53e-04), (3.42e-05, 2.55e-04), (5.45e-04, 2.59e-04), (1.27e-03, 5.50e-04), (2.03e-03, 4.45e-04), (1.74e-03, 2.08e-03)],
    "ba": [(9.31e-10, 3.75e-06), (7.39e-06, 2.26e-06), (1.31e-04, 1.69e-04), (4.92e-03, 2.70e-04), (8.02e-03, 1.77e-03), (5.95e-03, 2.78e-03), (5.07e-03, 3.51e-03), (6.71e-03, 2.11e-03)],
    "imu": [(9.31e-10, 6.25e-05), (5.73e-05, 6.31e-05), (1.67e-04, 2.58e-04), (1.50e-04, 2.83e-04), (2.36e-04, 8.12e-04), (4.37e-04, 7.45e-04), (8.47e-04, 5.87e-04), (1.73e-04, 6.43e-04)],
    "estdepth": [(9.31e-10, 7.03e-05), (6.65e-05, 7.10e-05), (2.37e-04, 4.70e-04), (9.86e-04, 4.25e-04), (1.39e-03, 1.11e-03), (2.71e-03, 7.67e-04), (7.15e-04, 1.47e-03), (1.49e-03, 1.48e-03)],
    "white_bg": [(9.31e-10, 6.32e-05), (2.25e-05, 6.39e-05), (3.85e-05, 3.40e-04), (6.46e-05, 3.42e-04)],
    "sh2_python": [(9.31e-10, 4.11e-06), (3.12e-06, 2.68e-06), (1.25e-05, 1.90e-04), (1.93e-05, 1.08e-04)],
    "no_transform": [(9.31e-10, 1.02e-04), (1.42e-05, 1.03e-04), (3.67e-05, 2.60e-04), (7.33e-05, 3.21e-04)],
    "sh2_active": [(9.31e-10, 7.79e-07), (1.63e-06, 2.29e-06), (1.44e-05, 2.39e-04), (3.87e-05, 1.91e-04)],      # round 6 (one member so far: floors 2e-5 / 2e-4 in _g9L_bars)
}


def _g9L_bars(variant, idx):
    pose, mom = G9L_MEASURED[variant][idx]
    lo_p, lo_m = (2e-5, 2e-4) if variant == "sh2_active" else (1e-6, 1e-5)      # (a single measurement so far: wider floors)
    return min(max(3.0 * pose, lo_p), 1e-2 if variant == "ba" else 5e-3), min(max(3.0 * mom, lo_m), 1e-2)


@pytest.mark.parametrize("prefix", ["g9", "g9L"])
@pytest.mark.parametrize("variant", ["vigs", "vigs_rotfrozen", "splatam", "ba", "imu", "estdepth", "white_bg", "sh2_python", "no_transform", "sh2_active"])
def test_native_hip_loops_reproduce_the_reference_classes_end_to_end(variant, prefix):
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
    from tests import g9_util
    slam, G, rows = run_variant(variant, prefix=prefix)
    want_kf = [[int(v) for v in s.split(",")] for s in G["keyframes"]]
    aligned = True
    for r in rows:
        idx = r["idx"]
        assert r["keyframes"] == want_kf[idx], (idx, r["keyframes"], want_kf[idx])
        assert abs(r["P"] - r["P_ref"]) <= max(2, 0.005 * r["P_ref"]), (idx, r["P"], r["P_ref"])
        aligned = aligned and r["P"] == r["P_ref"]
        large = prefix == "g9L"
        if idx == 0 and not large:
            assert aligned
        loose_ba = variant == "ba" and idx >= 3
        if large:
            bar = _g9L_bars(variant, idx)[0]
        else:
            # (no_transform: in that mode the pose gradient carries the covariance-rotation terms too, which largely cancel over the map; its
            #  float32 floor is ~10x the camera-frame mode's -- tests/test_gpu_fused.py, world-frame population -- and the runs separate
            #  sooner: 7.7e-6 after the first tracked frame, 1.1e-4 after the second, against 5e-9 / 2e-8 for sh2_python on the same frames)
            bar = 1e-2 if loose_ba else (((5e-4 if variant in ("ba", "white_bg", "no_transform") else 1e-4) if aligned else 1e-3))
        assert r["pose_diff"] < bar, (idx, r["pose_diff"], bar)
        if large:
            assert float(np.abs(r["moments"] - r["moments_ref"]).max()) <= _g9L_bars(variant, idx)[1], (idx, r["moments"], r["moments_ref"])
            continue
        tol = (5e-4 if variant == "no_transform" else 1e-4) if (aligned and not loose_ba) else 5e-3
        assert np.all(np.abs(r["moments"] - r["moments_ref"]) <= tol + tol * np.abs(r["moments_ref"])), (idx, r["moments"], r["moments_ref"])
    graph = [",".join(map(str, sorted(slam.mapper.covisibility_graph[k]))) for k in range(len(slam.mapper.keyframes))]
    assert graph == [str(s) for s in G["graph"]]
    for kf, ref in zip(slam.mapper.keyframes, G["keyframe_poses"]):
        d = (get_camera_from_tensor(kf.pose.detach().cpu().float()) - get_camera_from_tensor(torch.from_numpy(ref))).abs().max()
        assert d < (1e-2 if variant == "ba" else (5e-3 if prefix == "g9L" else 5e-4)), (kf.idx, float(d))
    # the final map as a population (rows are no longer aligned once a single pruning decision differs)
    g = slam.gaussians
    for name, t in (("xyz", g._xyz), ("opacity", g._opacity), ("scaling", g._scaling), ("rotation", g._rotation), ("f_dc", g._features_dc)):
        got, ref = g9_util.final_quantiles(G, name, t)
        for col in range(got.shape[1]):
            a, b = got[:, col], ref[:, col]
            assert (a - b).abs().max() < 0.02 * max(1.0, float(b.abs().max())), (name, col, a, b)
    # the three RNG streams were consumed exactly as the reference consumes them
    after = np.array([random.random(), float(np.random.rand()), float(torch.rand(1))])
    assert np.allclose(after, G["rng_after"]), (after, G["rng_after"])
    assert getattr(slam.mapper, "loop_reruns", 0) >= 0


# Round 5 (missing #2): the schedule the reference SHIPS and the benchmark times -- 100 tracking + 150 mapping iterations per
# frame, pruning_interval 50 with its no-op Adam steps at mapping iterations 0 and 50, min_opacity 0.005, kf_every 5, min_covisibility 0.95
# (/root/reference/configs/TUM.yml:32,44-50,73-75; `vigs`: those settings with sensor depth; `tum`: the file to the letter, `use_gt_depth: false` (:8) -- the mapper
# seeds from and regresses on the rescaled monocular estimate --; `imu`: configs/UTMM.yml's hot-path settings on the same schedule) -- held to
# slam/tracker.py:94-177 and slam/mapper.py:718-950 end to end: 11 frames at 160x120 (one Gaussian per pixel: 19.2 k), three keyframes,
# ~2750 optimiser iterations per variant (tests/golden/make_golden_slam.py --shipped-desk: more than an hour of the CPU oracle per variant).
# The camera follows mm3dgs_slam_amd.slam.trajectory_desk over a 1.8x wider scene: on the bounded trajectory of the other sets the view never
# loses 5 % of the last keyframe, and the shipped keyframe rule (slam/mapper.py:141-173: covisibility below min_covisibility AND kf_every frames
# since the last keyframe) would never spawn a second one.  With min_opacity 0.005 nothing sits near the pruning threshold, so the two maps
# keep the same rows much longer than in the g9L set.
# Measured (profiles/r05_g9D_hip_loops.txt; bars = 3 x, like the g9L set): identical keyframes (0 / 5 / 10), covisibility graph and RNG end state, map size
# within 23 of 20.8 k (0.1 %: a handful of seeding decisions -- silhouette against 0.5 -- at the two keyframes), camera matrices 2e-4 .. 9e-4, moments <= 1.3e-3 --
# ten times the agreement of the short fixtures' first tracked frames, because a frame here is 250 optimiser steps instead of ~24 and Adam carries every
# last-bit difference forward; the float32 floor of the same schedule (the reference's arithmetic re-run on CPU in another summation order,
# tools/g9_cpu_check.py --shipped --threads 7) is in DESIGN.md section 2.
G9D_MEASURED = {      # variant -> [(camera-matrix difference, largest moment difference) per frame]: ensemble maximum, see G9L_MEASURED
    "vigs": [(1.46e-11, 1.37e-04), (4.55e-04, 7.19e-04), (5.40e-04, 1.29e-03), (4.35e-04, 1.07e-03), (6.61e-04, 1.16e-03), (8.14e-04, 1.62e-03), (6.13e-04, 1.62e-03), (6.51e-04, 1.49e-03), (7.62e-04, 1.79e-03), (9.27e-04, 1.36e-03), (7.93e-04, 1.60e-03)],
    "imu": [(1.46e-11, 1.33e-04), (3.37e-04, 8.01e-04), (3.31e-04, 9.10e-04), (7.35e-04, 1.24e-03), (3.98e-04, 8.15e-04), (8.93e-04, 2.00e-03), (8.32e-04, 9.76e-04), (7.52e-04, 7.02e-04), (7.36e-04, 1.08e-03), (7.94e-04, 1.05e-03), (8.56e-04, 1.12e-03)],
    "tum": [(1.46e-11, 1.95e-04), (1.14e-04, 2.35e-04), (2.51e-04, 3.78e-04), (2.01e-04, 5.40e-04), (5.36e-04, 5.73e-04), (1.26e-03, 1.23e-03), (1.57e-03, 1.28e-03), (1.24e-03, 1.67e-03), (1.00e-03, 1.43e-03), (1.28e-03, 2.23e-03), (9.80e-04, 1.92e-03)],
}


@pytest.mark.parametrize("variant", ["vigs", "imu", "tum"])
def test_native_hip_loops_reproduce_the_reference_classes_at_the_shipped_schedule(variant):
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
    from tests import g9_util
    if not os.path.exists(os.path.join(HERE, "golden", f"g9D_{variant}.npz")):
        pytest.skip("fixture not generated (tests/golden/make_golden_slam.py --shipped-desk)")
    slam, G, rows = run_variant(variant, prefix="g9D")
    want_kf = [[int(v) for v in s.split(",")] for s in G["keyframes"]]
    assert want_kf[-1] == [0, 5, 10]
    for r in rows:
        idx = r["idx"]
        assert r["keyframes"] == want_kf[idx], (idx, r["keyframes"], want_kf[idx])
        assert abs(r["P"] - r["P_ref"]) <= max(2, 0.005 * r["P_ref"]), (idx, r["P"], r["P_ref"])
        pose_bar, mom_bar = (5e-3, 1e-2)
        if variant in G9D_MEASURED:
            pose_bar, mom_bar = min(max(3.0 * G9D_MEASURED[variant][idx][0], 1e-6), 5e-3), min(max(3.0 * G9D_MEASURED[variant][idx][1], 1e-5), 1e-2)
        assert r["pose_diff"] < pose_bar, (idx, r["pose_diff"], pose_bar)
        assert float(np.abs(r["moments"] - r["moments_ref"]).max()) <= mom_bar, (idx, r["moments"], r["moments_ref"])
    graph = [",".join(map(str, sorted(slam.mapper.covisibility_graph[k]))) for k in range(len(slam.mapper.keyframes))]
    assert graph == [str(s) for s in G["graph"]]
    for kf, ref in zip(slam.mapper.keyframes, G["keyframe_poses"]):
        d = (get_camera_from_tensor(kf.pose.detach().cpu().float()) - get_camera_from_tensor(torch.from_numpy(ref))).abs().max()
        assert d < 5e-3, (kf.idx, float(d))
    g = slam.gaussians
    for name, t in (("xyz", g._xyz), ("opacity", g._opacity), ("scaling", g._scaling), ("rotation", g._rotation), ("f_dc", g._features_dc)):
        got, ref = g9_util.final_quantiles(G, name, t)
        for col in range(got.shape[1]):
            a, b = got[:, col], ref[:, col]
            assert (a - b).abs().max() < 0.02 * max(1.0, float(b.abs().max())), (name, col, a, b)
    after = np.array([random.random(), float(np.random.rand()), float(torch.rand(1))])
    assert np.allclose(after, G["rng_after"]), (after, G["rng_after"])
