"""Pose algebra on 7-vectors (qw,qx,qy,qz,tx,ty,tz) = world->camera, device-agnostic and differentiable.

Behavioural mirror of the reference's ``utils/pose_utils.py`` (names and argument meaning kept so call sites read the
same): ``quad2rotation`` (:240-271), ``rotation2quad`` (:285-349), ``get_camera_from_tensor`` (:352-368),
``get_tensor_from_camera`` (:371-383), ``quadmultiply`` (:219-237), ``propagate_const_vel`` (:203-216),
``euler_matrix`` (:43-103, static 'sxyz'-family axes) and ``propagate_imu`` (:148-200).  Unlike the reference nothing
here hard-codes ``.cuda()``: tensors stay on the device of their inputs.
"""
from __future__ import annotations

import math

import torch

GRAVITY = (0.0, -9.80665, 0.0)  # camera optical frame (utils/pose_utils.py:40)


def quad2rotation(q: torch.Tensor) -> torch.Tensor:
    """[B,4] (w,x,y,z), normalised here -> [B,3,3]."""
    q = q / q.norm(dim=1, keepdim=True)
    w, x, y, z = q.unbind(1)
    row0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], 1)
    row1 = torch.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], 1)
    row2 = torch.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], 1)
    return torch.stack([row0, row1, row2], 1)


def rotation2quad(R: torch.Tensor) -> torch.Tensor:
    """[...,3,3] -> [...,4] real part first; picks the best-conditioned of the four candidate formulas
    (the largest of |w|,|x|,|y|,|z|), like the pytorch3d routine the reference borrows."""
    if R.shape[-2:] != (3, 3):
        raise ValueError(f"Invalid rotation matrix shape {R.shape}.")
    lead = R.shape[:-2]
    m = R.reshape(-1, 9)
    m00, m01, m02, m10, m11, m12, m20, m21, m22 = m.unbind(1)
    four_sq = torch.stack([1 + m00 + m11 + m22, 1 + m00 - m11 - m22, 1 - m00 + m11 - m22, 1 - m00 - m11 + m22], 1)
    mag = torch.sqrt(torch.clamp(four_sq, min=0.0))          # 2|w|, 2|x|, 2|y|, 2|z|
    cand = torch.stack([
        torch.stack([four_sq[:, 0], m21 - m12, m02 - m20, m10 - m01], 1),
        torch.stack([m21 - m12, four_sq[:, 1], m10 + m01, m02 + m20], 1),
        torch.stack([m02 - m20, m10 + m01, four_sq[:, 2], m12 + m21], 1),
        torch.stack([m10 - m01, m20 + m02, m21 + m12, four_sq[:, 3]], 1)], 1)      # [B, 4 candidates, 4]
    cand = cand / (2.0 * torch.clamp(mag, min=0.1))[:, :, None]
    best = mag.argmax(1)
    out = cand[torch.arange(cand.shape[0], device=R.device), best]
    return out.reshape(lead + (4,))


def get_camera_from_tensor(pose: torch.Tensor) -> torch.Tensor:
    """7-vector -> 4x4 world->camera matrix (differentiable)."""
    pose = pose if pose.dim() == 2 else pose[None]
    R = quad2rotation(pose[:, :4])[0]
    top = torch.cat([R, pose[0, 4:7, None]], 1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=pose.dtype, device=pose.device)
    return torch.cat([top, bottom], 0).float()


def get_tensor_from_camera(RT: torch.Tensor, Tquad: bool = False) -> torch.Tensor:
    """4x4 -> 7-vector, detached."""
    RT = RT.detach()
    return torch.cat([rotation2quad(RT[None, :3, :3])[0], RT[:3, 3]])


def quadmultiply(q1: torch.Tensor, q2: torch.Tensor) -> torch.Tensor:
    """Hamilton product, broadcasting over leading dims."""
    w1, x1, y1, z1 = q1.unbind(-1)
    w2, x2, y2, z2 = q2.unbind(-1)
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2,
                        w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                        w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2], -1)


def propagate_const_vel(camm1: torch.Tensor, camm2: torch.Tensor) -> torch.Tensor:
    """Constant-velocity prediction: apply the last inter-frame motion once more."""
    w2c1 = get_camera_from_tensor(camm1)
    step = w2c1 @ torch.linalg.inv(get_camera_from_tensor(camm2))
    return get_tensor_from_camera(step @ w2c1)


def euler_matrix(ai, aj, ak, axes: str = "sxyz") -> torch.Tensor:
    """Homogeneous rotation from static-frame x-y-z Euler angles (the only sequence the reference uses,
    utils/pose_utils.py:133,191): R = Rz(ak) Ry(aj) Rx(ai)."""
    if axes != "sxyz":
        raise NotImplementedError("only the 'sxyz' sequence is used by the SLAM path")
    ai, aj, ak = float(ai), float(aj), float(ak)
    si, sj, sk = math.sin(ai), math.sin(aj), math.sin(ak)
    ci, cj, ck = math.cos(ai), math.cos(aj), math.cos(ak)
    M = torch.eye(4)
    M[0, 0], M[0, 1], M[0, 2] = cj * ck, sj * si * ck - ci * sk, sj * ci * ck + si * sk
    M[1, 0], M[1, 1], M[1, 2] = cj * sk, sj * si * sk + ci * ck, sj * ci * sk - si * ck
    M[2, 0], M[2, 1], M[2, 2] = -sj, cj * si, cj * ci
    return M


def propagate_imu(camm1, camm2, imu_meas_list, c2i, dt_cam, dt_imu) -> torch.Tensor:
    """Dead-reckon the camera pose through the IMU samples between two frames (utils/pose_utils.py:148-200):
    velocity from the previous two poses, gravity removed in the IMU frame, angular velocity from columns 13:16 and
    linear acceleration from columns 25:28 of each sample.  NB the reference subtracts gravity from the sample
    in place; so does this (the caller's tensor is modified)."""
    dev = camm1.device
    c2i = c2i.to(dev)
    i2c = torch.linalg.inv(c2i)
    i2w1 = torch.linalg.inv(get_camera_from_tensor(camm1)) @ i2c
    i2w2 = torch.linalg.inv(get_camera_from_tensor(camm2)) @ i2c
    lin_vel = (torch.linalg.inv(i2w2) @ i2w1)[:3, 3] / dt_cam
    grav = torch.tensor(GRAVITY, dtype=i2w1.dtype, device=dev)
    i2w = i2w1.clone()
    for meas in imu_meas_list:
        acc = meas[25:28]
        acc -= i2w[:3, :3].T @ grav
        dpos = lin_vel * dt_imu + 0.5 * acc * dt_imu * dt_imu
        rot = meas[13:16] * dt_imu
        delta = euler_matrix(rot[0], rot[1], rot[2], "sxyz").to(i2w)
        delta[:3, 3] = dpos
        i2w = i2w @ delta
    return get_tensor_from_camera(torch.linalg.inv(i2w @ c2i))


def rigid_inverse(M: torch.Tensor) -> torch.Tensor:
    """Inverse of a rigid 4x4 [R|t; 0 1] in closed form ([R^T | -R^T t]): no solver call (the first ``torch.linalg.inv`` on
    the device costs tens of milliseconds of library initialisation inside a SLAM frame)."""
    R, t = M[:3, :3], M[:3, 3]
    Rt = R.t()
    top = torch.cat([Rt, -(Rt * t[None, :]).sum(1, keepdim=True)], 1)
    bottom = torch.tensor([[0.0, 0.0, 0.0, 1.0]], dtype=M.dtype, device=M.device)
    return torch.cat([top, bottom], 0)


def apply_rigid(pts: torch.Tensor, M: torch.Tensor) -> torch.Tensor:
    """[n,3] points through the rigid 4x4 M (x -> R x + t) with broadcast arithmetic instead of a BLAS call (differentiable)."""
    R, t = M[:3, :3], M[:3, 3]
    return pts[:, 0:1] * R[:, 0] + pts[:, 1:2] * R[:, 1] + pts[:, 2:3] * R[:, 2] + t


def propagate_const_vel_np(camm1, camm2):
    """``propagate_const_vel`` on the host in float64 numpy (7 floats in, 7 out): the per-frame pose prediction costs ~1 ms as
    a chain of ~100 tiny torch operators, ~30 us this way.  Same algebra (normalised quaternion -> R, step = W1 W2^-1 with
    the closed-form rigid inverse, W1' = step W1, best-conditioned matrix -> quaternion branch)."""
    import numpy as np

    def to_mat(p):
        q = np.asarray(p[:4], dtype=np.float64)
        q = q / np.linalg.norm(q)
        w, x, y, z = q
        M = np.eye(4)
        M[:3, :3] = [[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]]
        M[:3, 3] = np.asarray(p[4:7], dtype=np.float64)
        return M

    W1, W2 = to_mat(camm1), to_mat(camm2)
    W2i = np.eye(4)
    W2i[:3, :3] = W2[:3, :3].T
    W2i[:3, 3] = -W2[:3, :3].T @ W2[:3, 3]
    Wn = (W1 @ W2i) @ W1
    m = Wn[:3, :3]
    four_sq = np.array([1 + m[0, 0] + m[1, 1] + m[2, 2], 1 + m[0, 0] - m[1, 1] - m[2, 2], 1 - m[0, 0] + m[1, 1] - m[2, 2],
                        1 - m[0, 0] - m[1, 1] + m[2, 2]])
    mag = np.sqrt(np.clip(four_sq, 0.0, None))
    cand = np.array([[four_sq[0], m[2, 1] - m[1, 2], m[0, 2] - m[2, 0], m[1, 0] - m[0, 1]],
                     [m[2, 1] - m[1, 2], four_sq[1], m[1, 0] + m[0, 1], m[0, 2] + m[2, 0]],
                     [m[0, 2] - m[2, 0], m[1, 0] + m[0, 1], four_sq[2], m[1, 2] + m[2, 1]],
                     [m[1, 0] - m[0, 1], m[2, 0] + m[0, 2], m[2, 1] + m[1, 2], four_sq[3]]])
    best = int(np.argmax(mag))
    q = cand[best] / (2.0 * max(mag[best], 0.1))
    return np.concatenate([q, Wn[:3, 3]])

