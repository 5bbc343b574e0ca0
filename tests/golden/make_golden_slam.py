"""G9: an end-to-end SLAM run of the REFERENCE's own classes as a golden trajectory (build container only).

    python tests/golden/make_golden_slam.py

`slam/tracker.py::Tracker`, `slam/mapper.py::Mapper`, `slam/gaussian_model.py::GaussianModel` and `slam/renderer.py::Renderer` are
imported from /root/reference and driven exactly like `slam/SLAM.py:375-493` drives them (frame 0 takes the ground-truth pose,
later frames are tracked; `camera_extent` from frame 0; the mapper runs on every frame) over a small in-memory RGB-D sequence.
The one thing the reference cannot bring along -- its CUDA rasterizer extension -- is replaced by this repository's CPU oracle
(`oracle/raster_ref.py`), injected under the two names `slam/renderer.py:15-18` imports; `device="cuda"` literals run on the CPU
through the same TorchFunctionMode the other fixtures use.  Both sides of the comparison (`tests/test_golden_slam.py` runs this
repository's torch-graph `Tracker` / `Mapper` with the same oracle rasterizer on the same frames) therefore share the rasterizer
arithmetic, and what the fixture pins is everything around it: the order in which the three RNGs are consumed (`random.randint`
keyframe picks, `np.random.permutation` window subsets), keyframe decisions and the covisibility graph, seeding masks and order,
the densification statistics, the pruning schedule and its interaction with Adam, both optimisers, pose propagation.

Stored: the input frames and ground-truth poses (g9_frames.npz) and, per variant (g9_<variant>.npz: the shipped method, `method:
splatam`, bundle adjustment, no sensor depth, the UTMM-style IMU configuration), the configuration overrides and per frame the estimated pose, the keyframe indices, the number of
Gaussians and a few moments of the parameters; the final parameters, keyframe poses and covisibility graph in full."""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(1, ROOT)

import make_golden as mg          # noqa: E402  (puts /root/reference first on sys.path, stubs the absent third-party modules)

H, W, N_FRAMES, N_SEED = 48, 64, 5, 2500
SHORT = {"no_transform": 3, "sh2_python": 3, "white_bg": 3, "sh2_active": 3}          # frames of the short variants
PREFIX = "g9"
# --large (round 4): the same runs at 160x120 -- 80 tiles instead of 12, ~18 k Gaussians (one per valid pixel of frame 0, as the reference
# seeds), 8 frames, keyframes 0 / 2 / 4 / 6 -- so that the native kernels' machinery (direct bins, XCD tile map, load-balanced tile table,
# tile lists of hundreds of splats) is held to the reference's own classes through whole loops, not only through single renders.  Written
# as g9L_*.npz.  The inputs are stored the way a dataset stores them -- 8-bit colour, 16-bit depth at TUM's png_depth_scale 5000
# (configs/TUM.yml:88), float16 monocular stand-ins -- and both sides read the dequantised values; of the final map only per-column
# quantiles are kept (the consumers compare populations, not rows).
LARGE = "--large" in sys.argv
if LARGE:
    sys.argv.remove("--large")
    H, W, N_FRAMES, N_SEED = 120, 160, 8, 16000
    SHORT = {"no_transform": 4, "sh2_python": 4, "white_bg": 4, "sh2_active": 4}
    PREFIX = "g9L"
# --shipped (round 5): 160x120 again, but with the schedule the reference SHIPS and the benchmark times (configs/TUM.yml:32,44-75: 100 tracking /
# 150 mapping iterations, pruning_interval 50 -- with its no-op Adam steps at mapping iterations 0 and 50 --, min_opacity 0.005, kf_every 5,
# min_covisibility 0.95, size_threshold 100) over 11 frames = three keyframes (0 / 5 / 10): ~2750 optimiser iterations per variant through
# slam/tracker.py:94-177 and slam/mapper.py:718-950.  Written as g9S_*.npz in the --large storage format.
# --shipped-desk: the same schedule on the hand-held sweep of the bench's `moving` line (mm3dgs_slam_amd.slam.trajectory_desk over a 1.8x wider
# scene).  On the bounded trajectory of the other sets the view keeps > 95 % of the last keyframe in sight, so with the shipped min_covisibility
# no second keyframe is ever spawned (slam/mapper.py:141-173 needs BOTH: covisibility below the threshold AND kf_every frames since the last one);
# on the sweep kf_every 5 spaces them: keyframes 0 / 5 / 10, seeding of newly seen surface at 5 and 10, a three-keyframe covisibility graph and
# window.  Written as g9D_*.npz.
SHIPPED = "--shipped" in sys.argv or "--shipped-desk" in sys.argv
MOTION = "bounded"
if SHIPPED:
    LARGE = True
    H, W, N_FRAMES, N_SEED = 120, 160, 11, 16000
    SHORT = {}
    PREFIX = "g9S"
    if "--shipped-desk" in sys.argv:
        sys.argv.remove("--shipped-desk")
        PREFIX, MOTION = "g9D", "desk"
    else:
        sys.argv.remove("--shipped")
QS = [0.02, 0.1, 0.25, 0.5, 0.75, 0.9, 0.98]
_MAP = {"iters": 14, "kf_every": 2, "min_covisibility": 0.999, "densify_until_iter": 9, "pruning_interval": 4, "min_opacity": 0.4625,
        "size_threshold": 20}
VARIANTS = {
    # the shipped method (configs/TUM.yml schema): covisibility-graph window, L1 + SSIM + Pearson mapping loss, pruning every 4 iterations
    "vigs": dict(tracking={"iters": 10}, mapping=dict(_MAP)),
    # the same with the Gaussians' rotation learning rate at 0.  Seeded Gaussians are exactly isotropic, so d(loss)/d(rotation) is
    # analytically zero and numerically rounding noise, which Adam(eps=1e-15) turns into full +-lr steps: a random walk of the
    # quaternions whose direction no two float32 implementations share (and which becomes a real effect once the scales have gone
    # anisotropic).  With it frozen, a different rasterizer arithmetic (the HIP kernels, tests/test_gpu_golden_slam.py) can be held to
    # the reference trajectory tightly over all five frames
    "vigs_rotfrozen": dict(tracking={"iters": 10}, mapping=dict(_MAP, rotation_lr=0.0)),
    # (bundle adjustment stays loose even so: at 64x48 the rotation about the optical axis is barely constrained, and the reference's
    #  own arithmetic re-run by the torch-graph loops drifts from it by 6e-3 at frame 3 once one pruning decision differs)
    # method: splatam -- keyframe every kf_every frames, overlap-ranked window (torch.randint samples), sum-L1 tracking loss,
    # depth-L1 mapping loss, pruning at mapping iterations 0 and 20 only, its own seeding rule
    "splatam": dict(method="splatam", tracking={"iters": 10}, mapping=dict(_MAP, iters=22)),
    # bundle adjustment: pose optimiser over the window (slam/mapper.py:742-760,812-825,944-948), covisible-Gaussian mask
    "ba": dict(tracking={"iters": 10}, mapping=dict(_MAP, do_BA=True)),
    # configs/UTMM.yml's hot-path settings: IMU dead-reckoning for the pose prediction (utils/pose_utils.py:148-200 inside
    # Tracker.run_frame), Pearson depth term in the tracking loss, isotropic Gaussians, 0.002 pose learning rates
    # no sensor depth (use_gt_depth: false): the monocular estimate (here: a synthetic inverse-depth map) feeds the tracking Pearson term
    # with its min-of-two-targets form (utils/loss_utils.py:43-61), its rescaled version (slam/SLAM.py:413-450 computes it; here
    # given) seeds new Gaussians and feeds the mapping Pearson term
    "estdepth": dict(use_gt_depth=False, tracking={"iters": 10, "use_depth_estimate_loss": True}, mapping=dict(_MAP)),
    # renderer branches outside the shipped configs (3 frames each): world-frame means with the full view matrix -- including the
    # transposed matrix slam/renderer.py:207-214 hands to get_depth_and_silhouette in that mode --, SH evaluated in Python with
    # max_sh_degree 2 (f_rest rows through seeding, pruning and the optimiser groups), white background (composited under the depth
    # bundle too, so the silhouette is 1 everywhere)
    "no_transform": dict(pipeline={"transform_means_python": False}, tracking={"iters": 8}, mapping=dict(_MAP, iters=12)),
    "sh2_python": dict(pipeline={"convert_SHs_python": True}, tracking={"iters": 8}, mapping=dict(_MAP, iters=12, sh_degree=2)),
    "white_bg": dict(white_background=True, tracking={"iters": 8}, mapping=dict(_MAP, iters=12)),
    # round 6: an ACTIVE SH degree above 0 with the rasterizer's own SH evaluation -- the state of a map resumed from a checkpoint
    # (slam/gaussian_model.py:363 load_ply: active_sh_degree = max_sh_degree; oneupSHdegree is never called on the SLAM path): slam/renderer.py:179-193
    # hands shs = cat(f_dc, f_rest) and sh_degree = 2 to the rasterizer, with the pre-transformed means and campos = 0 of the shipped mode -- f_rest rows
    # (seeded at zero, slam/mapper.py:644-668) receive gradients and are stepped at feature_lr / 20, and the viewing direction carries gradient into
    # the means and the tracked pose.  ("_resumed_sh" is the fixture's own key: stripped before the reference's configuration is built)
    "sh2_active": dict(tracking={"iters": 8}, mapping=dict(_MAP, iters=12, sh_degree=2), _resumed_sh=True),
    "imu": dict(pipeline={"force_isotropic": True},
                tracking={"iters": 10, "dynamics_model": "imu", "use_depth_estimate_loss": True, "pearson_weight": 0.001, "position_lr": 0.002,
                          "rotation_lr": 0.002},
                mapping=dict(_MAP, pearson_weight=0.001)),
}


if SHIPPED:
    VARIANTS = {
        # configs/TUM.yml's hot-path settings (default_config's values ARE that file's) with sensor depth: nothing overridden
        "vigs": dict(),
        # configs/TUM.yml to the letter: `use_gt_depth: false` (:8) -- the tracker gets the monocular estimate (unused: tracking.use_depth_estimate_loss
        # is false, :38), the mapper seeds from and regresses (Pearson, :54-55) on its rescaled version
        "tum": dict(use_gt_depth=False),
        # configs/UTMM.yml's hot-path settings on the shipped schedule: IMU pose prediction, Pearson term in tracking, isotropic Gaussians,
        # 0.002 pose learning rates, size_threshold 200 (configs/UTMM.yml:31-77)
        "imu": dict(pipeline={"force_isotropic": True},
                    tracking={"dynamics_model": "imu", "use_depth_estimate_loss": True, "pearson_weight": 0.001, "position_lr": 0.002, "rotation_lr": 0.002},
                    mapping={"pearson_weight": 0.001, "cam_t_lr": 0.002, "cam_q_lr": 0.002, "size_threshold": 200}),
    }


def summary(g):
    op = torch.sigmoid(g._opacity)
    return np.array([g._xyz.shape[0], float(g._xyz.mean()), float(g._xyz.std()), float(op.mean()), float(op.std()),
                     float(g._scaling.mean()), float(g._scaling.std()), float(g._features_dc.mean()), float(g._rotation[:, 0].mean())],
                    dtype=np.float64)


def make_frames():
    """Inputs: a small synthetic RGB-D sequence (this repository's generator; inputs only, stored in the fixture)."""
    from oracle.raster_ref import RefRasterizer
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.renderer import Renderer as OurRenderer
    from mm3dgs_slam_amd.slam import SyntheticSequence
    cfg = default_config(device="cpu", height=H, width=W)
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    seq = SyntheticSequence(cfg, N_FRAMES, N_SEED, seed=3, renderer=OurRenderer(cfg, rasterizer_cls=RefRasterizer, mode="reference"), motion=MOTION)
    frames = [(c.clone(), d.clone()) for c, d in seq.frames]
    if LARGE:
        frames = [((c.clamp(0, 1) * 255.0).round().to(torch.uint8).float() / 255.0, (d * 5000.0).round().clamp(0, 65535).to(torch.int32).float() / 5000.0)
                  for c, d in frames]
    gt_poses = torch.stack([p.clone() for p in seq.poses])
    imu = torch.stack([seq.imu(i) if i else torch.zeros_like(seq.imu(1)) for i in range(N_FRAMES)])     # synthetic 100 Hz rows per frame interval
    # stand-ins for the monocular network's output (inverse-depth-like, arbitrary scale) and for its rescaled version
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    est = [1000.0 / (d + 0.5 + 0.05 * torch.sin(xx / 5.0 + i)) for i, (_, d) in enumerate(frames)]
    est_scaled = [torch.where(d > 0, d * (1.0 + 0.03 * torch.sin(xx / 9.0 + 0.3 * i) * torch.cos(yy / 7.0)), torch.full_like(d, 2.0)) for i, (_, d) in enumerate(frames)]
    if LARGE:
        est = [e.half().float() for e in est]
        est_scaled = [e.half().float() for e in est_scaled]
        np.savez_compressed(os.path.join(HERE, f"{PREFIX}_frames.npz"), H=H, W=W,
                            color_u8=np.stack([(mg.t2n(c) * 255.0).round().astype(np.uint8) for c, _ in frames]),
                            depth_u16=np.stack([(mg.t2n(d) * 5000.0).round().astype(np.uint16) for _, d in frames]), gt_poses=mg.t2n(gt_poses), imu=mg.t2n(imu),
                            tstamps=np.array(seq.tstamps, dtype=np.float64), est_f16=np.stack([mg.t2n(e).astype(np.float16) for e in est]),
                            est_scaled_f16=np.stack([mg.t2n(e).astype(np.float16) for e in est_scaled]))
        return frames, gt_poses, imu, list(seq.tstamps), est, est_scaled
    np.savez_compressed(os.path.join(HERE, "g9_frames.npz"), H=H, W=W, color=np.stack([mg.t2n(c) for c, _ in frames]),
                        depth=np.stack([mg.t2n(d) for _, d in frames]), gt_poses=mg.t2n(gt_poses), imu=mg.t2n(imu),
                        tstamps=np.array(seq.tstamps, dtype=np.float64), est=np.stack([mg.t2n(e) for e in est]),
                        est_scaled=np.stack([mg.t2n(e) for e in est_scaled]))
    return frames, gt_poses, imu, list(seq.tstamps), est, est_scaled


def run_reference(name, overrides, frames, gt_poses, imu, tstamps, mono, mono_scaled):
    from oracle.raster_ref import RefRasterizer, RefSettings
    from mm3dgs_slam_amd.config import default_config
    with mg._CpuMode():
        from slam import gaussian_model as ref_gm, mapper as ref_mapper, renderer as ref_renderer, tracker as ref_tracker
        from utils import pose_utils
        ref_renderer.GaussianRasterizer = RefRasterizer                     # the names slam/renderer.py:15-18 binds
        ref_renderer.GaussianRasterizationSettings = RefSettings
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        resumed_sh = bool(overrides.get("_resumed_sh", False))
        rcfg = default_config(device="cpu", height=H, width=W, **{k: v for k, v in overrides.items() if not k.startswith("_")})   # (the TUM.yml schema as a dict: configuration, not code)
        n_frames = SHORT.get(name, N_FRAMES)
        est = torch.zeros(n_frames, 7)
        use_imu = rcfg["tracking"]["dynamics_model"].lower() == "imu"                      # (slam/SLAM.py:44)
        ns = types.SimpleNamespace(cfg=rcfg, gaussians=ref_gm.GaussianModel(rcfg), n_img=n_frames, estimate_pose_list=est,
                                   gt_pose_list=torch.zeros(n_frames, 7), use_imu=use_imu, tf={"c2i": torch.eye(4)}, tstamps=tstamps)
        ns.gaussians.training_setup()
        if resumed_sh:      # what load_ply leaves behind (slam/gaussian_model.py:363)
            ns.gaussians.active_sh_degree = ns.gaussians.max_sh_degree
        ns.renderer = ref_renderer.Renderer(rcfg)
        mapper, tracker = ref_mapper.Mapper(ns), ref_tracker.Tracker(ns)
        per_frame, kf_lists, kf_poses = [], [], None
        for idx in range(n_frames):
            gt_color, gt_depth = frames[idx]
            e_raw, e_scaled = (None, None) if rcfg["use_gt_depth"] else (mono[idx], mono_scaled[idx])       # slam/SLAM.py:390-394,413-450
            gt_w2c = pose_utils.get_camera_from_tensor(gt_poses[idx])
            if idx == 0:
                est[idx] = pose_utils.get_tensor_from_camera(gt_w2c)
            else:
                tracker.run_frame(idx, gt_color, gt_depth, e_raw, imu[idx].clone() if use_imu else None)
            if idx == 0:        # slam/SLAM.py:456-463
                mapper.camera_extent = torch.max(gt_depth if rcfg["use_gt_depth"] else e_scaled) / rcfg["scene_radius_depth_ratio"]
            mapper.run_frame(idx, gt_color, gt_depth, e_scaled, None)
            per_frame.append(summary(ns.gaussians))
            kf_lists.append([kf.idx for kf in mapper.keyframes])
            print(f"{name} frame {idx}: P={ns.gaussians._xyz.shape[0]} keyframes={kf_lists[-1]}", flush=True)
        g = ns.gaussians
        out = dict(est_poses=mg.t2n(est), per_frame=np.stack(per_frame), keyframes=np.array([",".join(map(str, k)) for k in kf_lists]),
                   keyframe_poses=np.stack([mg.t2n(kf.pose) for kf in mapper.keyframes]),
                   graph=np.array([",".join(map(str, sorted(mapper.covisibility_graph[k]))) for k in range(len(mapper.keyframes))]),
                   **({name_: mg.t2n(torch.quantile(t_.detach().reshape(t_.shape[0], -1).float(), torch.tensor(QS), dim=0)) for name_, t_ in
                       (("q_xyz", g._xyz), ("q_opacity", g._opacity), ("q_scaling", g._scaling), ("q_rotation", g._rotation), ("q_f_dc", g._features_dc))}
                      if LARGE else
                      dict(xyz=mg.t2n(g._xyz), opacity=mg.t2n(g._opacity), scaling=mg.t2n(g._scaling), rotation=mg.t2n(g._rotation), f_dc=mg.t2n(g._features_dc))),
                   rng_after=np.array([random.random(), float(np.random.rand()), float(torch.rand(1))]), overrides=np.array(repr(overrides)))
    path = os.path.join(HERE, f"{PREFIX}_{name}.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "bytes")


def main():
    mg.stub_modules()
    sys.modules["pyiqa"].create_metric = lambda *a, **k: None

    def pearson_corrcoef(preds, target):
        """torchmetrics is not installed: the textbook definition it computes (cov / (sd sd), on 1-D inputs), differentiable --
        the G8 fixture binds the name to scipy.stats.pearsonr for values; an optimisation loop needs the gradient too."""
        x, y = preds - preds.mean(), target - target.mean()
        return (x * y).sum() / torch.sqrt((x * x).sum() * (y * y).sum())
    sys.modules["torchmetrics.functional.regression"].pearson_corrcoef = pearson_corrcoef
    frames, gt_poses, imu, tstamps, est, est_scaled = make_frames()
    for name in (sys.argv[1:] or list(VARIANTS)):
        run_reference(name, VARIANTS[name], frames, gt_poses, imu, tstamps, est, est_scaled)


if __name__ == "__main__":
    main()
