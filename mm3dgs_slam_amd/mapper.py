"""Map optimisation over the keyframe window (reference ``slam/mapper.py:36-1014``).

Kept from the reference (file:line into ``slam/mapper.py``): keyframe record; ``need_new_keyframe`` = covisibility with
the last keyframe + ``kf_every`` (:104-173; the NIQE sliding window :116-134 needs the ``pyiqa`` network and is out
of scope -- ``mapping.niqe_kf`` must be false here); covisibility graph from back-projected rendered depth
(:175-281); ``get_covisible_set`` breadth-first over ``covisibility_level`` with a random cap at ``kf_window_size``
(:283-407); new-Gaussian seeding from the non-presence mask (:409-493,600-688: one Gaussian per selected pixel,
isotropic log-scale ``log(z / ((fx+fy)/2))``, opacity logit 0, identity quaternion, ``f_dc = RGB2SH(rgb)``);
``optimize_map`` (:718-950): per iteration one keyframe popped from a refillable stack, render, loss
``(1-l)*L1 + l*(1-SSIM)`` (+ Pearson depth term, or the splatam depth-L1 variant), backward, densification statistics
and prune on the reference's schedule, Adam step; optional bundle adjustment of keyframe poses (``do_BA``).

Reference quirks kept: ``prune`` replaces the parameters before ``optimizer.step()`` so that step is a no-op on prune
iterations; ``get_covisible_gaussians`` ignores its ``min_kf`` argument and uses 2 (:716).
"""
from __future__ import annotations

import time
from collections import defaultdict
from random import randint

import numpy as np
import torch

from .loss_utils import l1_loss, pearson_loss, ssim
from .pose_utils import apply_rigid, get_camera_from_tensor, rigid_inverse
from .sh_utils import RGB2SH


def seed_subset(n, idx, frac, dev):
    """Workload knob `mapping.seed_fraction` (not in the reference): the fixed pseudo-random subset of a frame's pixels that may seed a
    Gaussian, drawn on the device the map lives on (a 307 k-element draw on the host + its upload cost ~1 ms of every keyframe event)."""
    dev = torch.device(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + int(idx))
    return torch.rand(n, generator=gen, device=dev) < frac


class KeyFrame:
    def __init__(self, idx, gt_color, pose, gt_depth=None, est_depth=None, niqe=None):
        self.idx, self.gt_color, self.pose, self.gt_depth, self.est_depth, self.niqe = idx, gt_color, pose, gt_depth, est_depth, niqe


class Mapper:
    def __init__(self, cfg, gaussians, renderer, estimate_pose_list, n_img=0, window=None):
        if cfg["mapping"].get("niqe_kf", False):
            raise NotImplementedError("mapping.niqe_kf needs the pyiqa NIQE network (out of scope); set it to false")
        self.cfg = cfg
        self.gaussians = gaussians
        self.renderer = renderer
        self.estimate_pose_list = estimate_pose_list
        self.n_img = n_img
        self.num_iter = cfg["mapping"]["iters"]
        self.camera_extent = 0
        self.keyframes = []
        self.covisibility_graph = defaultdict(set)
        self.mapping_time_sum = 0.0
        self.mapping_iter_count = 0
        self.window = window      # WindowParallel or None (window_parallel.py): multi-GPU mapping window

    # ---- geometry helpers -----------------------------------------------------------------------------------------
    def _intr(self):
        c = self.cfg["cam"]
        return float(c["fx"]), float(c["fy"]), float(c["cx"]), float(c["cy"])

    def get_depth_pointcloud(self, depth, w2c, sampled_indices):
        fx, fy, cx, cy = self._intr()
        z = depth[sampled_indices[:, 0], sampled_indices[:, 1]]
        cam = torch.stack(((sampled_indices[:, 1] - cx) / fx * z, (sampled_indices[:, 0] - cy) / fy * z, z), -1)
        pts = apply_rigid(cam, rigid_inverse(w2c))
        keep = torch.round(pts, decimals=4).abs().sum(1) > 0          # drop points at the world origin
        return pts[keep]

    def is_covisible(self, depth_pcd, camera_pose, height, width, threshold=0.9):
        fx, fy, cx, cy = self._intr()
        w2c = get_camera_from_tensor(camera_pose)
        p = apply_rigid(depth_pcd, w2c)
        z = p[:, 2] + 1e-5
        u, v = (fx * p[:, 0] + cx * p[:, 2]) / z, (fy * p[:, 1] + cy * p[:, 2]) / z
        inside = (u < width) & (u > 0) & (v < height) & (v > 0) & (z > 0)
        return inside.sum() / max(p.shape[0], 1) > threshold

    def covisibility_ratio_dense(self, depth, sil, kf_pose, cur_pose):
        """Fraction of the keyframe's surface points (rendered depth where the silhouette is > 0.99) that project inside the
        current view: ``get_depth_pointcloud`` + ``is_covisible`` of slam/mapper.py:141-173,175-216 evaluated densely over
        the image with masks instead of index lists -- the same points, the same tests, the same ratio, but no
        ``nonzero`` / gather kernels and a single host synchronisation (4 ms -> ~1 ms per keyframe test)."""
        fx, fy, cx, cy = self._intr()
        H, W = depth.shape
        key = (H, W, str(depth.device))
        if getattr(self, "_pix_grid_key", None) != key:
            v, u = torch.meshgrid(torch.arange(H, device=depth.device).float(), torch.arange(W, device=depth.device).float(), indexing="ij")
            self._pix_grid, self._pix_grid_key = (u.reshape(-1), v.reshape(-1)), key
        u, v = self._pix_grid
        z = torch.where(sil > 0.99, depth, torch.zeros_like(depth)).reshape(-1)
        valid = z > 0
        cam = torch.stack(((u - cx) / fx * z, (v - cy) / fy * z, z), -1)
        pts = apply_rigid(cam, rigid_inverse(get_camera_from_tensor(kf_pose)))
        sel = valid & (torch.round(pts, decimals=4).abs().sum(1) > 0)          # drop points at the world origin
        p = apply_rigid(pts, get_camera_from_tensor(cur_pose))
        zc = p[:, 2] + 1e-5
        uu, vv = (fx * p[:, 0] + cx * p[:, 2]) / zc, (fy * p[:, 1] + cy * p[:, 2]) / zc
        inside = (uu < W) & (uu > 0) & (vv < H) & (vv > 0) & (zc > 0) & sel
        return inside.sum() / sel.sum().clamp_min(1)

    def _render_depth_sil(self, pose):
        """(alpha-weighted depth, silhouette) of the map seen from `pose` (no gradients)."""
        result = self.renderer.render(self.gaussians, camera_pose=pose)
        return result["depth"][0], result["depth"][1]

    def _rendered_depth_cloud(self, pose):
        with torch.no_grad():
            result = self.renderer.render(self.gaussians, camera_pose=pose)
            depth = result["depth"][0].clone()
            depth[~(result["depth"][1] > 0.99)] = 0
            idx = torch.stack(torch.where(depth > 0), dim=1)
            return self.get_depth_pointcloud(depth, get_camera_from_tensor(pose), idx), depth.shape

    # ---- keyframes ----------------------------------------------------------------------------------------------------
    def add_keyframe(self, idx, est_pose, gt_color, gt_depth=None, est_depth=None):
        kf = KeyFrame(idx, gt_color, est_pose, gt_depth, est_depth)
        self.keyframes.append(kf)
        if idx > 0:
            self.update_covisibility_graph(len(self.keyframes) - 1)
        return kf

    def need_new_keyframe(self, idx, est_pose, gt_color, gt_depth=None, est_depth=None) -> bool:
        m = self.cfg["mapping"]
        if self.cfg["method"].lower() == "splatam":
            return idx == 0 or (idx + 1) % m["kf_every"] == 0 or idx == self.n_img - 2
        if len(self.keyframes) == 0 or idx == 0:
            return True
        # The reference renders the last keyframe and tests covisibility first (slam/mapper.py:141-173), but a frame closer
        # than kf_every to the last keyframe is rejected on either branch: decide that without the render (same result).
        if idx - self.keyframes[-1].idx < m["kf_every"]:
            return False
        with torch.no_grad():
            depth, sil = self._render_depth_sil(self.keyframes[-1].pose)
            ratio = self.covisibility_ratio_dense(depth, sil, self.keyframes[-1].pose, self.estimate_pose_list[idx])
        if bool(ratio > m["min_covisibility"]):
            return False
        return idx - self.keyframes[-1].idx >= m["kf_every"]

    def update_covisibility_graph(self, key):
        pts, (h, w) = self._rendered_depth_cloud(self.keyframes[key].pose)
        for kid, kf in enumerate(self.keyframes[:-1]):
            if self.is_covisible(pts, kf.pose, h, w, threshold=self.cfg["mapping"]["kf_covisibility"]):
                self.covisibility_graph[key].add(kid)
                self.covisibility_graph[kid].add(key)

    def _overlap_window(self, camera_pose, gt_depth):
        """method == 'splatam' (slam/mapper.py:289-372): rank the earlier keyframes by the fraction of 1600 randomly sampled
        surface points of the current view (ground-truth depth, or the rendered depth where the silhouette is > 0.99) that
        project at least 20 px inside them; keep those with a non-zero fraction, cap the window with a random subset and
        append the most recent keyframe."""
        if self.cfg["use_gt_depth"]:
            depth = gt_depth
        else:
            d, sil = self._render_depth_sil(camera_pose)
            depth = torch.where(sil > 0.99, d, torch.zeros_like(d))
        H, W = depth.shape
        valid = torch.stack(torch.where(depth > 0), dim=1)
        if valid.shape[0] == 0:
            scores = []
        else:
            pick = valid[torch.randint(valid.shape[0], (1600,)).to(valid.device)]
            pts = self.get_depth_pointcloud(depth, get_camera_from_tensor(camera_pose), pick)
            fx, fy, cx, cy = self._intr()
            edge = 20
            scores = []
            for kid, kf in enumerate(self.keyframes[:-1]):
                p = apply_rigid(pts, get_camera_from_tensor(kf.pose))
                z = p[:, 2:3] + 1e-5
                u, v = (fx * p[:, 0:1] + cx * p[:, 2:3]) / z, (fy * p[:, 1:2] + cy * p[:, 2:3]) / z
                inside = (u < W - edge) & (u > edge) & (v < H - edge) & (v > edge) & (z > 0)
                scores.append((kid, float(inside.sum()) / max(p.shape[0], 1)))
        scores.sort(key=lambda t: t[1], reverse=True)          # (stable, like the reference's sorted())
        keep = np.array([kid for kid, frac in scores if frac > 0.0], dtype=np.int64)
        selected = [int(k) for k in np.random.permutation(keep)[: self.cfg["mapping"]["kf_window_size"] - 2]]
        if self.keyframes:
            selected.append(len(self.keyframes) - 1)
        return selected, [self.keyframes[k].idx for k in selected]

    def get_covisible_set(self, idx, camera_pose, gt_color, gt_depth=None, N=1):
        if idx == 0:
            return [], []
        if self.cfg["method"].lower() == "splatam":
            return self._overlap_window(camera_pose, gt_depth)
        cur = len(self.keyframes) - 1
        covisible = {cur}
        for _ in range(N):
            frontier = covisible.copy()
            for k in frontier:
                covisible.update(set(self.covisibility_graph[k]) - covisible)
            if frontier == covisible:
                break
        covisible.remove(cur)
        selected = list(np.random.permutation(np.array(list(covisible), dtype=np.int64))[: self.cfg["mapping"]["kf_window_size"] - 2])
        selected = [int(s) for s in selected] + [cur]
        return selected, [self.keyframes[s].idx for s in selected]

    # ---- seeding --------------------------------------------------------------------------------------------------------
    def get_pointcloud(self, color, depth, w2c, mask=None):
        """Back-project every pixel; returns ([n,6] xyz|rgb in the world frame, [n] squared pixel footprint)."""
        fx, fy, cx, cy = self._intr()
        H, W = depth.shape
        dev = depth.device
        v, u = torch.meshgrid(torch.arange(H, device=dev).float(), torch.arange(W, device=dev).float(), indexing="ij")
        z = depth.reshape(-1)
        cam = torch.stack(((u.reshape(-1) - cx) / fx * z, (v.reshape(-1) - cy) / fy * z, z), -1)
        pts = apply_rigid(cam, rigid_inverse(w2c))
        cld = torch.cat((pts, color.permute(1, 2, 0).reshape(-1, 3)), -1)
        msd = (z / ((fx + fy) / 2)) ** 2
        return (cld, msd) if mask is None else (cld[mask], msd[mask])

    def initialize_new_gaussians(self, idx, camera_pose, gt_color, gt_depth=None, est_depth=None):
        depth = gt_depth if self.cfg["use_gt_depth"] else est_depth
        dev = depth.device
        if idx == 0 and "iteration" not in self.cfg:
            non_presence = torch.ones(depth.numel(), dtype=torch.bool, device=dev)
        else:
            result = self.renderer.render(self.gaussians, camera_pose=camera_pose)
            sil, rdepth = result["depth"][1], result["depth"][0]
            err = (depth - rdepth).abs() * (depth > 0)
            if self.cfg["method"].lower() == "splatam":     # slam/mapper.py:520-526: only surfaces IN FRONT of the map, 50 x median
                far = (rdepth > depth) & (err > 50 * err.median())
            else:
                far = err > 10 * err.median()
            non_presence = ((sil < 0.5) | far).reshape(-1)
        non_presence = non_presence & (depth > 0).reshape(-1)
        if self.cfg["method"].lower() == "splatam" and not bool(non_presence.any()):
            return None, non_presence.reshape(depth.shape)          # (slam/mapper.py:532,590-591)
        frac = float(self.cfg["mapping"].get("seed_fraction", 1.0))
        if frac < 1.0:     # workload knob (not in the reference): seed only a fixed pseudo-random subset of the pixels
            non_presence = non_presence & seed_subset(non_presence.numel(), idx, frac, dev)
        cld, msd = self.get_pointcloud(gt_color, depth, get_camera_from_tensor(camera_pose), mask=non_presence)
        n = cld.shape[0]
        rgb = cld[:, 3:6].float()
        n_coef = (self.gaussians.max_sh_degree + 1) ** 2
        rots = torch.zeros((n, 4), device=dev)
        rots[:, 0] = 1
        self.gaussians.densification_postfix(
            new_xyz=cld[:, :3].float(), new_features_dc=RGB2SH(rgb)[:, None, :].contiguous(),
            new_features_rest=torch.zeros((n, n_coef - 1, 3), device=dev), new_opacities=torch.zeros((n, 1), device=dev),
            new_scaling=torch.log(torch.sqrt(msd))[:, None].repeat(1, 3), new_rotation=rots, new_rgb=rgb)
        new_mask = torch.zeros(self.gaussians.get_xyz.shape[0], dtype=torch.bool, device=dev)
        new_mask[-n:] = True
        return new_mask, non_presence.reshape(depth.shape)

    def get_covisible_gaussians(self, keyframe_idx_list, curr_camera_tensor, min_kf=2):
        with torch.no_grad():
            seen = torch.zeros(self.gaussians.get_xyz.shape[0], device=self.cfg["device"])
            for k in keyframe_idx_list:
                pose = curr_camera_tensor if k == -1 else self.keyframes[k].pose
                seen += self.renderer.render(self.gaussians, camera_pose=pose)["visibility_filter"].int()
        return seen >= 2

    # ---- optimisation -----------------------------------------------------------------------------------------------------
    def _loss(self, result, gt_color, gt_depth, est_depth):
        cfg, m = self.cfg, self.cfg["mapping"]
        image, depth = result["render"], result["depth"][0]
        photo = (1 - m["lambda_dssim"]) * l1_loss(image, gt_color) + m["lambda_dssim"] * (1.0 - ssim(image, gt_color))
        if cfg["method"].lower() == "splatam":
            unc = (result["depth"][2] - depth ** 2).detach()
            mask = ((gt_depth > 0) & ~torch.isnan(depth) & ~torch.isnan(unc)).detach()
            return (gt_depth - depth).abs()[mask].mean() + 0.5 * photo
        if m["use_depth_estimate_loss"]:
            if not cfg["use_gt_depth"]:
                photo = photo + m["pearson_weight"] * pearson_loss(depth, est_depth, invert_estimate=False)
            else:
                photo = photo + m["pearson_weight"] * pearson_loss(depth, gt_depth, mask=gt_depth > 0, invert_estimate=False)
        return photo

    def optimize_map(self, idx, num_iter, keyframe_idx_list, new_gaussians_mask, curr_camera_tensor, curr_gt_color,
                     curr_gt_depth=None, curr_est_depth=None):
        if num_iter == 0:
            return
        cfg, m, g = self.cfg, self.cfg["mapping"], self.gaussians
        cur_q, cur_T = curr_camera_tensor[:4], curr_camera_tensor[4:]
        do_ba = m["do_BA"] and idx > 0
        pose_opt = opt_mask = None
        if do_ba:
            qs = [cur_q.requires_grad_()] + [self.keyframes[k].pose[:4].requires_grad_() for k in keyframe_idx_list if k != -1]
            Ts = [cur_T.requires_grad_()] + [self.keyframes[k].pose[4:].requires_grad_() for k in keyframe_idx_list if k != -1]
            pose_opt = torch.optim.Adam([{"params": qs, "lr": m["cam_q_lr"], "name": "cam_rot"},
                                         {"params": Ts, "lr": m["cam_t_lr"], "name": "cam_pos"}], lr=0.0, eps=1e-15)
            opt_mask = self.get_covisible_gaussians(keyframe_idx_list, curr_camera_tensor, 2)
            if new_gaussians_mask is not None:
                opt_mask |= new_gaussians_mask
        stack = None
        stats = cfg["debug"]["get_runtime_stats"]
        for iteration in range(num_iter):
            t0 = time.perf_counter() if stats else 0.0
            def pop():
                nonlocal stack
                if not stack:
                    stack = list(keyframe_idx_list)
                return stack.pop(randint(0, len(stack) - 1))
            # one optimiser step = one view (the reference, slam/mapper.py:803-807) or, with a window, this rank's share of the
            # world x batch views of the step: autograd accumulates their gradients in .grad
            ids = self.window.take(pop) if self.window is not None else [pop()]
            wstats = None
            for k in ids:
                if k == -1:
                    q, T, gt_color, gt_depth, est_depth = cur_q, cur_T, curr_gt_color, curr_gt_depth, curr_est_depth
                else:
                    kf = self.keyframes[k]
                    q, T, gt_color, gt_depth, est_depth = kf.pose[:4], kf.pose[4:], kf.gt_color, kf.gt_depth, kf.est_depth
                result = self.renderer.render(g, camera_pose=torch.cat([q, T]))
                loss = self._loss(result, gt_color, gt_depth, est_depth)
                loss.backward()
                if self.window is not None:
                    with torch.no_grad():
                        wstats = self.window.merge_stats(wstats, self.window.view_stats(result["viewspace_points"], result["visibility_filter"],
                                                                                        result["radii"]))
            with torch.no_grad():
                reduced = None
                if self.window is not None:
                    reduced = self.window.reduce(g, wstats)
                if cfg["method"].lower() == "splatam":
                    if iteration <= 20 and iteration % 20 == 0:
                        g.prune(m["min_opacity"], self.camera_extent)
                elif iteration <= m["densify_until_iter"]:
                    if reduced is not None:
                        g.max_radii2D = torch.max(g.max_radii2D, reduced[2])
                        g.xyz_gradient_accum += reduced[0]
                        g.denom += reduced[1]
                    else:
                        vis, radii = result["visibility_filter"], result["radii"]
                        g.max_radii2D[vis] = torch.max(g.max_radii2D[vis], radii[vis].to(g.max_radii2D.dtype))
                        g.add_densification_stats(result["viewspace_points"], vis)
                    if iteration >= m["densify_from_iter"] and iteration % m["pruning_interval"] == 0:
                        pruned = g.prune(m["min_opacity"], self.camera_extent, m["size_threshold"])
                        if do_ba:
                            opt_mask = opt_mask[~pruned]
                if do_ba:
                    for group in g.optimizer.param_groups:
                        for p in group["params"]:
                            if p.grad is not None:
                                p.grad[~opt_mask] = 0
                g.optimizer.step()
                g.optimizer.zero_grad(set_to_none=True)
                if do_ba:
                    if self.window is not None and self.window._collective:
                        # every replica must take the identical pose step: sum the pose gradients of the ranks' views
                        self.window.reduce_pose_grads(qs + Ts)
                    pose_opt.step()
                    pose_opt.zero_grad(set_to_none=True)
            if stats:
                self.mapping_time_sum += time.perf_counter() - t0
                self.mapping_iter_count += 1

    def run_frame(self, idx, gt_color, gt_depth=None, est_depth=None, imu_meas=None):
        camera_pose = self.estimate_pose_list[idx]
        new_vis_mask = new_gaussians_mask = None
        with torch.no_grad():
            kf_list, _ = self.get_covisible_set(idx, camera_pose, gt_color, gt_depth, N=self.cfg["mapping"]["covisibility_level"])
            kf_list.append(-1)
            if self.need_new_keyframe(idx, camera_pose, gt_color, gt_depth, est_depth):
                new_gaussians_mask, new_vis_mask = self.initialize_new_gaussians(idx, camera_pose, gt_color, gt_depth, est_depth)
                self.add_keyframe(idx, camera_pose, gt_color, gt_depth, est_depth)
        self.optimize_map(idx, self.num_iter, kf_list, new_gaussians_mask, camera_pose, gt_color, gt_depth, est_depth)
        return new_vis_mask
