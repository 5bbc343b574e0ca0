"""Synthetic inputs (no dataset is available offline): random Gaussian clouds for kernel parity tests and the
SLAM-shaped scene of SURVEY.md section 8(d) (plane + boxes depth map, smooth-noise RGB, Gaussians seeded one per
pixel exactly like the reference's first-frame initialisation, slam/mapper.py:437-474,644-668)."""
from __future__ import annotations

import math

import torch

TUM_INTRINSICS = dict(fx=517.3, fy=516.5, cx=318.6, cy=255.3, H=480, W=640)  # configs/TUM.yml:84-87,16-17


def projection_matrix(znear, zfar, fx, fy, cx, cy, h, w, dtype=torch.float32):
    """Same matrix as utils/graphics_utils.py:85-94 (getProjectionMatrix2)."""
    return torch.tensor([[2 * fx / w, 0.0, -(w - 2 * cx) / w, 0.0],
                         [0.0, 2 * fy / h, -(h - 2 * cy) / h, 0.0],
                         [0.0, 0.0, zfar / (zfar - znear), -(zfar * znear) / (zfar - znear)],
                         [0.0, 0.0, 1.0, 0.0]], dtype=dtype)


def random_cloud(P, H, W, fx=None, fy=None, seed=0, dtype=torch.float32, sh_coeffs=0, spread=1.2, log_scale=-3.0,
                 zmin=0.5, zmax=3.5):
    """Random anisotropic Gaussians filling (and overfilling by `spread`) the frustum of an identity camera."""
    g = torch.Generator().manual_seed(seed)
    fx = fx or 0.8 * W
    fy = fy or fx
    r = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    n = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    z = r(P) * (zmax - zmin) + zmin
    x = (r(P) * 2 - 1) * spread * z * W / (2 * fx)
    y = (r(P) * 2 - 1) * spread * z * H / (2 * fy)
    out = dict(
        means3D=torch.stack([x, y, z], -1),
        scales=torch.exp(n(P, 3) * 0.5 + log_scale) * z[:, None],
        rotations=torch.nn.functional.normalize(n(P, 4), dim=1),
        opacities=torch.sigmoid(n(P, 1) * 2.0),
        colors=r(P, 3),
    )
    if sh_coeffs:
        out["shs"] = n(P, sh_coeffs, 3) * 0.3
    return {k: v.to(dtype) for k, v in out.items()}, fx, fy


def camera_matrices(H, W, fx, fy, cx=None, cy=None, w2c=None, dtype=torch.float32, znear=0.01, zfar=100.0):
    """(viewmatrix, projmatrix, campos, tanfovx, tanfovy) in the reference's row-vector convention
    (slam/renderer.py:61-62,117-124)."""
    cx = (W / 2 - 0.3) if cx is None else cx
    cy = (H / 2 + 0.4) if cy is None else cy
    Pm = projection_matrix(znear, zfar, fx, fy, cx, cy, H, W, dtype=torch.float64).t()
    w2c = torch.eye(4, dtype=torch.float64) if w2c is None else w2c.to(torch.float64)
    view = w2c.t()
    proj = view @ Pm
    campos = torch.linalg.inv(view)[3, :3]
    return view.to(dtype), proj.to(dtype), campos.to(dtype), W / (2 * fx), H / (2 * fy)


def small_pose(seed=1, angle=0.08, trans=0.15):
    """A 4x4 world-to-camera matrix a few degrees / centimetres off identity."""
    g = torch.Generator().manual_seed(seed)
    ax = torch.nn.functional.normalize(torch.randn(3, generator=g, dtype=torch.float64), dim=0)
    K = torch.tensor([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]], dtype=torch.float64)
    R = torch.eye(3, dtype=torch.float64) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)
    M = torch.eye(4, dtype=torch.float64)
    M[:3, :3] = R
    M[:3, 3] = torch.randn(3, generator=g, dtype=torch.float64) * trans
    return M
