"""Trajectory / image metrics the harness writes into ``results.npz`` (reference ``utils/eval_utils.py:139-189,231-294``,
``utils/image_utils.py:17-19``).  Pose part only: LPIPS needs a downloaded VGG network (out of scope).

``evaluate_ate_rmse(est, gt, "umeyama")`` aligns the translation columns ``[:, 4:]`` of the ESTIMATE to the ground truth with the
similarity transform of Umeyama (IEEE PAMI 13(4), 1991; scale + rotation + translation, reflection guarded by the sign of
det(U) det(V)), rotates the estimated quaternions by the same rotation, and returns (aligned poses [n,7], RMSE of the residual
translation).  The reference calls it on the raw 7-vectors (world->camera), ``slam/SLAM.py:339-343``."""
from __future__ import annotations

import numpy as np
import torch

from .pose_utils import quad2rotation, rotation2quad


def align_umeyama(model: np.ndarray, data: np.ndarray, known_scale: bool = False):
    """s, R, t with model ~ s R data + t (least squares over the n x 3 point sets)."""
    mu_m, mu_d = model.mean(0), data.mean(0)
    m0, d0 = model - mu_m, data - mu_d
    n = model.shape[0]
    C = m0.T @ d0 / n
    sigma2 = (d0 * d0).sum() / n
    U, D, Vt = np.linalg.svd(C)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt.T) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    s = 1.0 if known_scale else float(np.trace(np.diag(D) @ S) / sigma2)
    t = (mu_m - s * R @ mu_d)[:, None]
    return s, R, t


def align_horn(model: np.ndarray, data: np.ndarray):
    """Closed-form rigid alignment of Horn (reference ``utils/eval_utils.py:193-228``): R, t with data ~ R model + t for 3 x n point
    sets, and the per-point residual norms."""
    mu_m, mu_d = model.mean(1, keepdims=True), data.mean(1, keepdims=True)
    Wm = (model - mu_m) @ (data - mu_d).T                   # sum of outer(model_i, data_i)
    U, _, Vh = np.linalg.svd(Wm.T)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vh) < 0:
        S[2, 2] = -1
    R = U @ S @ Vh
    t = mu_d - R @ mu_m
    err = R @ model + t - data
    return R, t, np.sqrt((err * err).sum(0))


def evaluate_ate_rmse(est_poses, gt_poses, method: str = "umeyama"):
    assert len(est_poses) == len(gt_poses), "Estimated trajectory and GT trajectory must have equal length"
    est = est_poses.detach().cpu().numpy() if isinstance(est_poses, torch.Tensor) else np.asarray(est_poses)
    gt = gt_poses.detach().cpu().numpy() if isinstance(gt_poses, torch.Tensor) else np.asarray(gt_poses)
    est_traj, gt_traj = est[:, 4:], gt[:, 4:]
    aligned = est.copy()
    if method.lower() == "umeyama":
        s, R, t = align_umeyama(gt_traj, est_traj)
        q = rotation2quad(torch.matmul(torch.tensor(R).float(), quad2rotation(torch.as_tensor(est[:, :4]).float()).float()))
        aligned[:, :4] = q.numpy()
        aligned[:, 4:] = (s * (R @ est_traj.T) + t).T
        ate = np.linalg.norm(aligned[:, 4:] - gt_traj, axis=1)
    elif method.lower() == "horn":       # rigid (no scale), utils/eval_utils.py:249-266
        R, t, ate = align_horn(est_traj.T, gt_traj.T)
        q = rotation2quad(torch.matmul(torch.tensor(R).float(), quad2rotation(torch.as_tensor(est[:, :4]).float()).float()))
        aligned[:, :4] = q.numpy()
        aligned[:, 4:] = (R @ est_traj.T + t).T
    else:                                # (the reference's fall-through, utils/eval_utils.py:231-294: no alignment -- said aloud here)
        import warnings
        warnings.warn(f"evaluate_ate_rmse: unknown alignment method {method!r} (umeyama / horn): the trajectories are compared without alignment")
        ate = np.linalg.norm(est_traj - gt_traj, axis=1)
    rmse = float(np.sqrt(np.dot(ate, ate) / len(ate)))
    if isinstance(est_poses, torch.Tensor):
        aligned = torch.tensor(aligned)
    return aligned, rmse


def psnr(img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))
