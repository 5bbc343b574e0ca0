"""Image / depth / pose losses that consume the rendered image (reference ``utils/loss_utils.py``): ``l1_loss``
(:64-68), ``ssim`` (:95-154, 11x11 Gaussian window sigma 1.5, C1=0.01^2, C2=0.03^2), ``pearson_loss`` (:43-61, with
the correlation computed directly instead of through torchmetrics), ``rel_pose_loss`` (:20-40)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from .pose_utils import quadmultiply


def l1_loss(network_output, gt, mask=None):
    d = (network_output - gt).abs()
    return d.mean() if mask is None else d[:, mask].mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


_window_cache = {}


def _window(size: int, channel: int, like: torch.Tensor) -> torch.Tensor:
    key = (size, channel, like.device, like.dtype)
    w = _window_cache.get(key)
    if w is None:
        g = torch.tensor([math.exp(-((i - size // 2) ** 2) / (2 * 1.5 ** 2)) for i in range(size)])
        g = (g / g.sum()).float()
        w = (g[:, None] @ g[None, :]).expand(channel, 1, size, size).contiguous().to(like)
        _window_cache[key] = w
    return w


def ssim(img1, img2, window_size: int = 11, size_average: bool = True):
    ch = img1.size(-3)
    win = _window(window_size, ch, img1)
    pad = window_size // 2
    x, y = (img1, img2) if img1.dim() == 4 else (img1[None], img2[None])

    def blur(t):
        return F.conv2d(t, win, padding=pad, groups=ch)

    mu1, mu2 = blur(x), blur(y)
    s11 = blur(x * x) - mu1 * mu1
    s22 = blur(y * y) - mu2 * mu2
    s12 = blur(x * y) - mu1 * mu2
    c1, c2 = 0.01 ** 2, 0.03 ** 2
    smap = ((2 * mu1 * mu2 + c1) * (2 * s12 + c2)) / ((mu1 * mu1 + mu2 * mu2 + c1) * (s11 + s22 + c2))
    return smap.mean() if size_average else smap.mean(1).mean(1).mean(1)


def pearson_corrcoef(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    a = a - a.mean()
    b = b - b.mean()
    return (a * b).sum() / (torch.sqrt((a * a).sum() * (b * b).sum()))


def pearson_loss(render, estimate, mask=None, invert_estimate=True):
    r = render if mask is None else render[mask]
    e = estimate if mask is None else estimate[mask]
    if invert_estimate:
        return torch.minimum(1 - pearson_corrcoef(-e, r), 1 - pearson_corrcoef(1 / (e + 200.0), r))
    return 1 - pearson_corrcoef(e, r)


def rel_pose_loss(camera_pose, initial_pose, safe: bool = False):
    """(squared translation error, rotation angle of the relative quaternion) -- utils/loss_utils.py:20-40.

    ``safe=False`` is the literal reference: at ``camera_pose[:4] == initial_pose[:4]`` (which is exactly the first
    tracking iteration, slam/tracker.py:87) the angle is acos(1) and autograd returns NaN (-inf times 0).  ``safe=True``
    takes the gradient of the angle term as 0 where |cos| >= 1, which is what the native tracking loop does."""
    t_err = ((camera_pose[4:] - initial_pose[4:]) ** 2).sum()
    conj = initial_pose[:4].detach() * torch.tensor([1.0, -1.0, -1.0, -1.0], device=initial_pose.device)
    diff = F.normalize(quadmultiply(camera_pose[:4], conj)[None], dim=1)[0]
    c = diff[0].abs()
    if safe:
        c = torch.where(c < 1.0, c, c.detach().clamp(max=1.0))
    return t_err, 2 * torch.acos(c)
