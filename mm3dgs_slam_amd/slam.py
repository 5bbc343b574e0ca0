"""Frame loop: load -> track -> map (reference ``slam/SLAM.py:375-493``), over an in-memory RGB-D sequence.

The hot-path part of the reference's orchestrator: frame 0 takes the ground-truth pose (``SLAM.py:399-401``), later frames are
tracked, ``camera_extent = max(depth) / scene_radius_depth_ratio`` is fixed on frame 0 (``:456-463``), then the mapper runs.
Harness outputs in the reference's formats: ``save_map`` = ``outputdir/point_cloud/iteration_<n>/point_cloud.ply``
(``SLAM.py:286-292``), ``save_iterations`` checkpoints inside ``run`` (``:488-492``) and the final map (``:497-500``),
``save_results`` = ``outputdir/results.npz`` with the reference's key set (``:294-373``: pose_est, pose_gt, keyframes, ate_rmse,
psnr_list, ssim_list, lpips_list, avg_tracking_it_time, avg_mapping_it_time), and resuming from a checkpoint when the
configuration carries ``iteration`` (``:90-104`` map + poses, ``slam/mapper.py:65-71`` keyframes + covisibility graph).  Dataset
loaders, MiDaS depth alignment, debug videos and LPIPS (a downloaded network; its list stays empty) are out of scope
(SURVEY.md section 2)."""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import synthetic
from .gaussian_model import GaussianModel
from .mapper import Mapper
from .pose_utils import get_camera_from_tensor, get_tensor_from_camera
from .renderer import Renderer
from .tracker import Tracker


def trajectory(n, step_t=0.01, step_r=0.008):
    """Smooth 6-DoF sinusoid, <= ~2 cm / 1 degree per frame.  Returns n 4x4 world->camera matrices."""
    out = []
    for i in range(n):
        a = i / 7.0
        rx, ry, rz = step_r * 2 * math.sin(a), step_r * 3 * math.sin(0.7 * a + 0.3), step_r * math.sin(1.3 * a)
        M = torch.eye(4)
        cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
        Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
        Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
        M[:3, :3] = Rz @ Ry @ Rx
        M[:3, 3] = torch.tensor([step_t * 3 * math.sin(0.9 * a), step_t * 2 * math.sin(0.5 * a + 1.0), step_t * 2 * (1 - math.cos(0.6 * a))])
        out.append(M)
    return out


def trajectory_moving(n, step_t=0.014, step_r=0.014):
    """A camera that keeps MOVING (the bench's `moving` line): a lateral sweep of +-0.5 m past the scene at up to `step_t` m per frame
    (TUM fr1/desk: ~1.4 cm per frame) with the yaw that keeps the scene centre (3 m ahead) in view, a slow vertical bob and a roll /
    pitch wobble of up to `step_r` rad (0.8 degrees) per frame.  Unlike `trajectory` the displacement accumulates -- tens of
    centimetres within a few dozen frames --, so the keyframe test fires every handful of frames, new Gaussians are seeded, the map
    grows and the mapping window fills.  Returns n 4x4 world->camera matrices."""
    out = []
    A = 0.5
    w = step_t / A                                    # peak speed A w = step_t
    for i in range(n):
        x = A * math.sin(w * i)
        y = 0.08 * math.sin(0.5 * w * i + 0.7)
        z = 0.10 * (1.0 - math.cos(0.6 * w * i))
        yaw = math.atan2(x, 3.0)                      # look at the point 3 m in front of the start pose
        pitch = -math.atan2(y, 3.0) + (0.55 * step_r / 0.15) * math.sin(0.15 * i)      # wobble: amplitude x frequency = rate per frame
        roll = (0.5 * step_r / 0.2) * math.sin(0.2 * i + 0.4)
        cy_, sy_, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
        Ry = torch.tensor([[cy_, 0, -sy_], [0, 1, 0], [sy_, 0, cy_]])          # camera-to-world yaw about the camera's y axis (x right, y down, z forward)
        Rx = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = torch.tensor([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        c2w = torch.eye(4)
        c2w[:3, :3] = (Ry @ Rx @ Rz).float()
        c2w[:3, 3] = torch.tensor([x, y, z])
        out.append(torch.linalg.inv(c2w))
    return out


def trajectory_desk(n, step_t=0.014, step_r_deg=0.8, amp=1.0):
    """A hand-held sweep at TUM fr1/desk's pace (the bench's `moving` line since round 4): the camera pans back and forth over the scene --
    yaw +-10 degrees at up to `step_r_deg` per frame (0.8), pitch +-4 degrees at up to 0.4 per frame -- while it translates sideways
    +-0.25 m at up to `step_t` m per frame (1.4 cm) and bobs a few centimetres.  At that pace the view shares less than
    `mapping.min_covisibility` (0.95) of the last keyframe within two or three frames, so `mapping.kf_every: 5` is what spaces the
    keyframes (slam/mapper.py:141-173, configs/TUM.yml:45-50) except at the turning points of the sweep.  The sequence needs a scene
    wider than the first view: SyntheticSequence(motion="desk") seeds its ground-truth map from a 1.8x wider / taller virtual frame.
    `amp` scales the three amplitudes at the same pace per frame (motion="desk_wide": 1.6 -- pan +-16 degrees, the whole 1.8x scene comes
    into view over a sweep: the bench grows a map to a stated size with it).  Returns n 4x4 world->camera matrices."""
    out = []
    ay, ap, ax = math.radians(10.0) * amp, math.radians(4.0) * amp, 0.25 * amp
    wy, wp, wx = math.radians(step_r_deg) / ay, math.radians(0.5 * step_r_deg) / ap, step_t / ax
    for i in range(n):
        yaw = ay * math.sin(wy * i)
        pitch = ap * math.sin(wp * i + 0.5)
        roll = math.radians(1.5) * math.sin(0.11 * i + 0.4)
        x = ax * math.sin(wx * i)
        y = 0.04 * math.sin(0.6 * wx * i + 0.7)
        z = 0.05 * (1.0 - math.cos(0.5 * wx * i))
        cy_, sy_, cp, sp, cr, sr = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch), math.cos(roll), math.sin(roll)
        Ry = torch.tensor([[cy_, 0, sy_], [0, 1, 0], [-sy_, 0, cy_]])
        Rx = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
        Rz = torch.tensor([[cr, -sr, 0], [sr, cr, 0], [0, 0, 1]])
        c2w = torch.eye(4)
        c2w[:3, :3] = (Ry @ Rx @ Rz).float()
        c2w[:3, 3] = torch.tensor([x, y, z])
        out.append(torch.linalg.inv(c2w))
    return out


class SyntheticSequence:
    """RGB-D frames rendered from a fixed ground-truth Gaussian map along ``trajectory`` (built once, untimed).
    The ground-truth map is the seeded first frame of ``synthetic.rgbd_frame`` -- for ``motion="desk"`` a 1.8x wider and taller
    virtual first frame with the same focal length (the sweep looks past the edges of the first view), seeded at the same density."""

    def __init__(self, cfg, n_frames, n_gaussians, seed=0, renderer=None, motion="bounded"):
        dev = cfg["device"]
        H, W = int(cfg["desired_height"]), int(cfg["desired_width"])
        c = cfg["cam"]
        if motion in ("desk", "desk_wide"):
            Hg, Wg = int(round(1.8 * H)), int(round(1.8 * W))
            color, depth = synthetic.rgbd_frame(Hg, Wg, seed=seed, n_boxes=14)
            G = synthetic.seed_gaussians(color, depth, c["fx"], c["fy"], c["cx"] + 0.5 * (Wg - W), c["cy"] + 0.5 * (Hg - H),
                                         int(n_gaussians * (Hg * Wg) / (H * W)), seed=seed)
        else:
            color, depth = synthetic.rgbd_frame(H, W, seed=seed)
            G = synthetic.seed_gaussians(color, depth, c["fx"], c["fy"], c["cx"], c["cy"], n_gaussians, seed=seed)
        self.seed_params = {k: v.to(dev) for k, v in G.items()}
        traj = {"moving": trajectory_moving, "desk": trajectory_desk, "desk_wide": lambda n: trajectory_desk(n, amp=1.6)}.get(motion, trajectory)
        self.poses = [get_tensor_from_camera(M).to(dev) for M in traj(n_frames)]
        self.frames = []
        renderer = renderer or Renderer(cfg)
        gt = _FixedMap(self.seed_params, cfg, opaque=True)
        with torch.no_grad():
            for p in self.poses:
                r = renderer.render(gt, p)
                sil = r["depth"][1]
                d = torch.where(sil > 0.5, r["depth"][0] / sil.clamp_min(1e-6), torch.zeros_like(sil))
                self.frames.append((r["render"].clamp(0, 1).contiguous(), d.contiguous()))

        self.dt_cam = 0.04                       # 25 frames/s: four IMU samples at 100 Hz between two frames
        self.tstamps = [i * self.dt_cam for i in range(n_frames)]
        self.tf = {"c2i": torch.eye(4)}          # camera and IMU frames coincide in the synthetic rig
        self._imu = {}
        self._est = {}

    def __len__(self):
        return len(self.frames)

    def __getitem__(self, i):
        color, depth = self.frames[i]
        return color, depth, self.poses[i]

    def est(self, i):
        """A stand-in for the monocular network's output on frame i (slam/SLAM.py:392-396): inverse-depth-like, arbitrary scale and
        offset, a smooth multiplicative error of a few percent -- what the per-frame least-squares alignment has to undo."""
        if i not in self._est:
            _, d = self.frames[i]
            H, W = d.shape
            yy, xx = torch.meshgrid(torch.arange(H, device=d.device).float(), torch.arange(W, device=d.device).float(), indexing="ij")
            wobble = 1.0 + 0.03 * torch.sin(xx / 37.0 + 0.3 * i) * torch.cos(yy / 29.0)
            self._est[i] = (1000.0 / (torch.where(d > 0, d, torch.full_like(d, 3.0)) * wobble + 0.5)).contiguous()
        return self._est[i]

    def imu(self, i):
        """Synthetic IMU rows for the interval (frame i-1, frame i], in the layout utils/pose_utils.py:179-180 reads (angular
        velocity in columns 13:16, linear acceleration incl. gravity in columns 25:28; 100 Hz): derived from the ground-truth
        trajectory so that ``propagate_imu`` started at the true poses i-2, i-1 lands on the true pose i.
        A fresh tensor per call: ``propagate_imu`` subtracts gravity from the sample in place, as the reference does."""
        if i not in self._imu:
            from .pose_utils import GRAVITY
            n, dt = int(round(self.dt_cam / 0.01)), 0.01
            inv = lambda p: torch.linalg.inv(get_camera_from_tensor(p.cpu()).double())
            w1 = inv(self.poses[i - 1])
            w0 = inv(self.poses[i - 2]) if i >= 2 else w1
            w2 = inv(self.poses[i])
            D = torch.linalg.inv(w1) @ w2                                   # motion over the interval, in the frame of pose i-1
            # small-angle rotation vector of D spread over the n samples (euler 'sxyz' of small angles ~ rotation vector)
            rv = torch.stack([D[2, 1] - D[1, 2], D[0, 2] - D[2, 0], D[1, 0] - D[0, 1]]) * 0.5
            omega = rv / (n * dt)
            dt_prev = self.dt_cam if i >= 2 else 1.0
            v = (torch.linalg.inv(w0) @ w1)[:3, 3] / dt_prev              # what propagate_imu uses as the (constant) velocity
            acc = 2.0 * (D[:3, 3] / n - v * dt) / (dt * dt)
            g_imu = w1[:3, :3].T @ torch.tensor(GRAVITY, dtype=torch.float64)
            rows = torch.zeros(n, 30)
            rows[:, 13:16] = omega.float()
            rows[:, 25:28] = (acc + g_imu).float()
            self._imu[i] = rows
        return self._imu[i].clone()


class _FixedMap:
    """Minimal GaussianModel-shaped holder for rendering a parameter dict."""

    def __init__(self, G, cfg, opaque=False):
        self.active_sh_degree = 0
        self.max_sh_degree = cfg["mapping"]["sh_degree"]
        self._xyz, self._scaling, self._rotation = G["xyz"], G["scaling"], G["rotation"]
        self.get_xyz = G["xyz"]
        self.get_opacity = torch.full_like(G["opacity"], 0.98) if opaque else torch.sigmoid(G["opacity"])
        self.get_scaling = torch.exp(G["scaling"])
        self.get_rotation = torch.nn.functional.normalize(G["rotation"])
        n_rest = (self.max_sh_degree + 1) ** 2 - 1
        self.get_features = torch.cat([G["f_dc"], torch.zeros(G["f_dc"].shape[0], n_rest, 3, device=G["f_dc"].device)], 1)


class SLAM:
    def __init__(self, cfg, sequence, rasterizer_cls=None, settings_cls=None, render_mode="fused", window=None, native_loops=None):
        self.cfg = cfg
        self.seq = sequence
        self.gaussians = GaussianModel(cfg)
        resume = None
        if "iteration" in cfg:      # checkpoint: the map of save_map(iteration) and the poses / keyframes of results.npz (slam/SLAM.py:90-104)
            self.gaussians.load_ply(os.path.join(cfg["outputdir"], "point_cloud", "iteration_" + str(cfg["iteration"]), "point_cloud.ply"))
            resume = np.load(os.path.join(cfg["outputdir"], "results.npz"), allow_pickle=True)
        self.gaussians.training_setup()
        self.renderer = Renderer(cfg, rasterizer_cls=rasterizer_cls, settings_cls=settings_cls, mode=render_mode)
        n = len(sequence)
        self.estimate_pose_list = [None] * n
        # native (HIP) iteration loops by default on a GPU with the HIP rasterizer; the torch-graph loops otherwise
        if native_loops is None:
            native_loops = rasterizer_cls is None and str(cfg["device"]).startswith("cuda")
        if native_loops:
            from .fused import FusedMapper as MapperCls, FusedTracker as TrackerCls
        else:
            TrackerCls, MapperCls = Tracker, Mapper
        self.tracker = TrackerCls(cfg, self.gaussians, self.renderer, self.estimate_pose_list, tf=getattr(sequence, "tf", None),
                                  tstamps=getattr(sequence, "tstamps", None))
        self.mapper = MapperCls(cfg, self.gaussians, self.renderer, self.estimate_pose_list, n_img=n, window=window)
        self.gt_pose_list = [None] * n
        if resume is not None:
            dev = cfg["device"]
            for i, p in enumerate(resume["pose_est"][:n]):
                self.estimate_pose_list[i] = torch.tensor(p, device=dev)
            # slam/mapper.py:65-71: KeyFrame(**kf) for every stored keyframe, then the covisibility graph is rebuilt edge by edge
            from .mapper import KeyFrame
            to_dev = lambda v: None if v is None else torch.as_tensor(v).to(dev)
            for kf in resume["keyframes"]:
                self.mapper.keyframes.append(KeyFrame(int(kf["idx"]), to_dev(kf["gt_color"]), to_dev(kf["est_pose"]), to_dev(kf["gt_depth"]),
                                                      to_dev(kf["est_depth"])))
            for k in range(len(self.mapper.keyframes)):
                self.mapper.update_covisibility_graph(k)

    def step(self, idx):
        """Track + map one frame (the unit the headline metric counts).  Without sensor depth (`use_gt_depth: false`, configs/TUM.yml:8)
        and with a sequence that provides a monocular estimate (`sequence.est(idx)`: the network is out of scope, its output is an input),
        the frame follows slam/SLAM.py:392-463: the tracker gets the raw estimate, then the map is rendered once at the tracked pose and
        the estimate is fitted to it by least squares (depth_utils.scale_depth_estimate); the mapper seeds from / regresses on that."""
        color, depth, gt_pose = self.seq[idx]
        mono = (not self.cfg["use_gt_depth"]) and hasattr(self.seq, "est")
        est = self.seq.est(idx) if mono else depth
        if idx == 0 or self.cfg["tracking"]["use_gt_pose"]:
            self.estimate_pose_list[idx] = gt_pose.clone()
        else:
            imu = self.seq.imu(idx) if (self.cfg["tracking"].get("dynamics_model") or "").lower() == "imu" else None
            self.tracker.run_frame(idx, color, depth, est, imu_meas=imu)
        est_scaled = depth
        if mono:
            from .depth_utils import scale_depth_estimate
            est_scaled = scale_depth_estimate(self.cfg, idx, est, depth, lambda: self.mapper._render_depth_sil(self.estimate_pose_list[idx]),
                                              resumed="iteration" in self.cfg)
        if idx == 0:
            self.mapper.camera_extent = float((est_scaled if mono else depth).max()) / self.cfg["scene_radius_depth_ratio"]
        self.mapper.run_frame(idx, color, depth, est_scaled)
        self.gt_pose_list[idx] = gt_pose.detach().clone()

    def run(self, progress=None, reraise=True):
        """slam/SLAM.py:375-503: every frame; `save_iterations` checkpoints on the way; with an `outputdir` the final map (as
        iteration <last_idx>) and results.npz at the end -- also when a frame raised: the reference catches the exception, prints it
        ("SLAM failed. Saving map and results.") and goes on to its `finally` (slam/SLAM.py:494-503).  Same here; the exception is kept
        in `self.failure`, and -- the default for library callers (a failed run must not look like a success) -- raised again
        AFTER the outputs were written (a failure while writing them never masks it: it is printed and the original one is raised).
        `reraise=False` is the reference's literal print-and-carry-on, which `slam_top.py` opts into (and turns into the exit code)."""
        last_idx = 0
        self.failure = None
        try:
            for idx in range(len(self.seq)):
                self.step(idx)
                if progress is not None:
                    progress(idx)
                if "outputdir" in self.cfg and idx in self.cfg.get("save_iterations", ()):
                    self.save_map(idx)
                last_idx += 1
        except Exception as e:      # noqa: BLE001 -- the reference's behaviour
            self.failure = e
            print(e)
            print("\nSLAM failed. Saving map and results.\n")
        finally:
            if "outputdir" in self.cfg and last_idx > 0:
                try:
                    with torch.no_grad():
                        self.save_map(last_idx)
                        self.save_results(last_idx)
                except Exception as e2:      # noqa: BLE001
                    if self.failure is None:
                        raise
                    print(f"(while saving after the failure above: {e2!r})")
        if self.failure is not None and reraise:
            raise self.failure

    # ---- outputs in the reference's formats ------------------------------------------------------------------------------------
    def save_map(self, iteration):
        path = os.path.join(self.cfg["outputdir"], "point_cloud", "iteration_{}".format(iteration))
        os.makedirs(path, exist_ok=True)
        self.gaussians.save_ply(os.path.join(path, "point_cloud.ply"))
        return os.path.join(path, "point_cloud.ply")

    def evaluate_images(self, last_idx):
        """PSNR / SSIM of the map rendered from the estimated pose of frame 0 and every `eval_every`-th frame (slam/SLAM.py:197-231;
        LPIPS needs a downloaded network and is left out: its list stays empty)."""
        from .eval_utils import psnr
        from .loss_utils import ssim
        every = int(self.cfg.get("eval_every", 1))
        psnr_list, ssim_list = [], []
        with torch.no_grad():
            for idx in range(last_idx):
                if idx != 0 and (idx + 1) % every != 0:
                    continue
                color = self.seq[idx][0]
                image = self.renderer.render(self.gaussians, camera_pose=self.estimate_pose_list[idx])["render"]
                psnr_list.append(psnr(image, color).mean().detach().cpu().numpy())
                ssim_list.append(ssim(image, color).detach().cpu().numpy())
        return psnr_list, ssim_list, []

    def save_results(self, last_idx):
        """results.npz with the key set, shapes and types of slam/SLAM.py:294-373 (np.load(..., allow_pickle=True) reads the keyframe
        dicts back, as slam/mapper.py:65-71 does)."""
        from .eval_utils import evaluate_ate_rmse
        est = torch.stack([p.detach() for p in self.estimate_pose_list[:last_idx]])
        gt = torch.stack([p.detach() for p in self.gt_pose_list[:last_idx]])
        results = {"pose_est": est.cpu().numpy(), "pose_gt": gt.cpu().numpy()}
        results["keyframes"] = [{"idx": kf.idx, "gt_color": kf.gt_color, "est_pose": kf.pose, "gt_depth": kf.gt_depth, "est_depth": kf.est_depth}
                                for kf in self.mapper.keyframes]
        _, results["ate_rmse"] = evaluate_ate_rmse(est, gt, method="umeyama")          # on the raw (world->camera) vectors, as the reference does
        psnr_list, ssim_list, lpips_list = self.evaluate_images(last_idx)
        results["psnr_list"], results["ssim_list"], results["lpips_list"] = psnr_list, ssim_list, lpips_list
        if self.cfg["debug"]["get_runtime_stats"]:
            results["avg_tracking_it_time"] = self.tracker.tracking_time_sum / max(self.tracker.tracking_iter_count, 1) * 1000
            results["avg_mapping_it_time"] = self.mapper.mapping_time_sum / max(self.mapper.mapping_iter_count, 1) * 1000
        os.makedirs(self.cfg["outputdir"], exist_ok=True)
        np.savez(os.path.join(self.cfg["outputdir"], "results"), **results)
        return results

    def pose_errors(self):
        """Translation error (m) of every estimated pose against the sequence's ground truth."""
        errs = []
        for est, gt in zip(self.estimate_pose_list, self.seq.poses):
            if est is None:
                continue
            a = torch.linalg.inv(get_camera_from_tensor(est))[:3, 3]
            b = torch.linalg.inv(get_camera_from_tensor(gt))[:3, 3]
            errs.append(float((a - b).norm()))
        return errs
