"""Per-frame camera pose optimisation (reference ``slam/tracker.py:12-266``).

``Tracker.run_frame(idx, gt_color, gt_depth, est_depth, imu_meas)`` keeps the reference's behaviour: pose initialised
from the previous estimate, the constant-velocity model (``tracker.py:203-206``) or IMU propagation (``:208-228``);
Adam on translation (``tracking.position_lr``) and quaternion (``tracking.rotation_lr``) (``:233-246``);
``tracking.iters`` x {render -> loss -> backward -> step}; loss = masked mean-L1 over ``silhouette > 0.99`` plus the
optional Pearson depth term and IMU relative-pose term (``:104-155``), or the "splatam" sum-L1 variant (``:110-126``).

Reference quirks kept by default: the best-loss candidate is tracked but the LAST iterate is what gets stored
(``tracker.py:180-181,262-264``; ``keep_best_candidate=True`` switches to the intended behaviour).  One
result-neutral optimisation: the map tensors enter the render detached, so the backward pass skips the Gaussian-side
gradients -- in the reference they are computed and then discarded by the mapper's prune.
"""
from __future__ import annotations

import time

import torch

from .loss_utils import pearson_loss, rel_pose_loss
from .pose_utils import propagate_const_vel, propagate_const_vel_np, propagate_imu


class _FrozenMap:
    """Read-only view of a GaussianModel whose tensors are detached (tracking never updates the map)."""

    def __init__(self, pc):
        self._pc = pc
        self.active_sh_degree = pc.active_sh_degree
        self.max_sh_degree = pc.max_sh_degree
        self._xyz = pc._xyz.detach()
        self._scaling = pc._scaling.detach()
        self._rotation = pc._rotation.detach()
        with torch.no_grad():
            self.get_xyz = self._xyz
            self.get_opacity = pc.get_opacity.detach()
            self.get_scaling = pc.get_scaling.detach()
            self.get_rotation = pc.get_rotation.detach()
            self.get_features = pc.get_features.detach()

    def get_covariance(self, scaling_modifier=1):
        with torch.no_grad():
            return self._pc.get_covariance(scaling_modifier)


class Tracker:
    def __init__(self, cfg, gaussians, renderer, estimate_pose_list, tf=None, tstamps=None, keep_best_candidate=False):
        self.cfg = cfg
        self.gaussians = gaussians
        self.renderer = renderer
        self.estimate_pose_list = estimate_pose_list
        self.tf = tf
        self.tstamps = tstamps
        self.num_iter = cfg["tracking"]["iters"]
        self.dyn_model = cfg["tracking"].get("dynamics_model")
        self.keep_best_candidate = keep_best_candidate
        self.tracking_time_sum = 0.0
        self.tracking_iter_count = 0

    def _loss(self, result, q, T, initial_pose, gt_color, gt_depth, est_depth):
        cfg, trk = self.cfg, self.cfg["tracking"]
        image = result["render"]
        depth, silhouette = result["depth"][0], result["depth"][1]
        presence = silhouette > 0.99
        if cfg["method"].lower() == "splatam":
            unc = (result["depth"][2] - depth ** 2).detach()
            mask = ((gt_depth > 0) & ~torch.isnan(depth) & ~torch.isnan(unc) & presence).detach()
            return (gt_depth - depth).abs()[mask].sum() + 0.5 * (gt_color - image).abs()[:, mask].sum()
        loss = (image - gt_color).abs()[:, presence].mean()
        if trk["use_depth_estimate_loss"]:
            if not cfg["use_gt_depth"]:
                loss = loss + trk["pearson_weight"] * pearson_loss(depth, est_depth, mask=presence, invert_estimate=True)
            else:
                loss = loss + trk["pearson_weight"] * pearson_loss(depth, gt_depth, mask=presence & (gt_depth > 0),
                                                                   invert_estimate=True)
        if trk["use_imu_loss"]:
            # (the reference's literal rel_pose_loss is NaN at its own starting point; tracking.imu_loss_literal restores that)
            t_l, q_l = rel_pose_loss(torch.cat([q, T]), initial_pose, safe=not trk.get("imu_loss_literal", False))
            loss = loss + trk["imu_T_weight"] * t_l + trk["imu_q_weight"] * q_l
        return loss

    def optimize_cam(self, idx, num_iter, optimizer, camera_tensor_q, camera_tensor_T, gt_color, gt_depth=None,
                     est_depth=None):
        frozen = _FrozenMap(self.gaussians)
        if num_iter == 0:
            with torch.no_grad():
                return 0, self.renderer.render(frozen, torch.cat([camera_tensor_q, camera_tensor_T]))["render"]
        initial_pose = torch.cat([camera_tensor_q, camera_tensor_T]).clone().detach()
        best_q, best_T, best_loss = camera_tensor_q.detach().clone(), camera_tensor_T.detach().clone(), None
        stats = self.cfg["debug"]["get_runtime_stats"]
        loss = image = None
        for _ in range(num_iter):
            t0 = time.perf_counter() if stats else 0.0
            result = self.renderer.render(frozen, torch.cat([camera_tensor_q, camera_tensor_T]))
            image = result["render"]
            loss = self._loss(result, camera_tensor_q, camera_tensor_T, initial_pose, gt_color, gt_depth, est_depth)
            loss.backward()
            with torch.no_grad():
                optimizer.step()
                optimizer.zero_grad(set_to_none=True)
                if self.keep_best_candidate:      # costs a host sync per iteration; the reference pays it always
                    if best_loss is None or loss < best_loss:
                        best_loss = loss.detach()
                        best_q, best_T = camera_tensor_q.detach().clone(), camera_tensor_T.detach().clone()
            if stats:
                self.tracking_time_sum += time.perf_counter() - t0
                self.tracking_iter_count += 1
        if self.keep_best_candidate:
            with torch.no_grad():
                camera_tensor_q.copy_(best_q)
                camera_tensor_T.copy_(best_T)
        return loss, image

    def predict_pose(self, idx, imu_meas=None):
        poses = self.estimate_pose_list
        cam = poses[idx - 1].clone().detach()
        model = (self.dyn_model or "").lower()
        if model == "const_velocity":
            if idx - 2 >= 0:
                if poses[idx - 1].is_cuda:
                    # one single-lane launch (double precision): no pose on the host, so the frame's tracking loop is enqueued while
                    # the GPU is still busy with the previous frame's mapping (the read-back drained the device once per frame)
                    cam = self._predict_const_vel_device(poses[idx - 1], poses[idx - 2])
                else:
                    # 7-float algebra on the host instead of ~100 one-element torch kernels
                    both = torch.stack([poses[idx - 1].detach(), poses[idx - 2].detach()]).cpu().numpy()
                    cam = torch.from_numpy(propagate_const_vel_np(both[0], both[1])).float()
        elif model == "imu":
            assert imu_meas is not None, "IMU measurements must be provided"
            # 4x4 algebra over a handful of samples: on the host (one 7-float copy; ~100 one-element device kernels otherwise)
            p1 = poses[idx - 1].detach().cpu()
            p2 = poses[idx - 2].detach().cpu() if idx - 2 >= 0 else p1
            dt_cam = (self.tstamps[idx - 1] - self.tstamps[idx - 2]) if idx - 2 >= 0 else 1.0      # (zero velocity at the start)
            cam = propagate_imu(p1, p2, imu_meas.cpu(), self.tf["c2i"].cpu(), dt_cam, 1 / 100.0)
        elif model:
            raise ValueError(f"Unknown dynamics model {self.dyn_model}")
        return cam

    @staticmethod
    def _predict_const_vel_device(pm1, pm2):
        """mm3dgs_propagate_const_vel: utils/pose_utils.py:203-216 on the device (same algebra as propagate_const_vel_np, float64)."""
        import ctypes as C
        from . import _lib
        from .rasterizer import _stream
        a, b = pm1.detach().float().contiguous(), pm2.detach().float().contiguous()
        out = torch.empty(7, dtype=torch.float32, device=a.device)
        _lib.check(_lib.load().mm3dgs_propagate_const_vel(C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), _stream()))
        return out

    def run_frame(self, idx, gt_color, gt_depth=None, est_depth=None, imu_meas=None):
        cam = self.predict_pose(idx, imu_meas).to(self.cfg["device"])
        camera_tensor_T = cam[-3:].clone().requires_grad_()
        camera_tensor_q = cam[:4].clone().requires_grad_()
        trk = self.cfg["tracking"]
        opt = torch.optim.Adam([{"params": [camera_tensor_T], "lr": trk["position_lr"]},
                                {"params": [camera_tensor_q], "lr": trk["rotation_lr"]}])
        _, image = self.optimize_cam(idx, self.num_iter, opt, camera_tensor_q, camera_tensor_T, gt_color, gt_depth,
                                     est_depth)
        with torch.no_grad():
            self.estimate_pose_list[idx] = torch.cat([camera_tensor_q, camera_tensor_T]).clone().detach()
        return image
