"""Per-frame alignment of a monocular depth estimate to the map (reference ``slam/SLAM.py:411-448`` with
``utils/depth_utils.py:44-99``): when the configuration has no sensor depth (``use_gt_depth: false`` -- what ``configs/TUM.yml:8``
ships), every frame renders the map once at the tracked pose (no gradients) and fits ``scale * est + shift`` to the INVERSE rendered
depth by least squares over { silhouette > 0.99, est > 1e-6 }; ``1 / (scale * est + shift)`` is the depth the mapper seeds from and the
target of its Pearson term.  The monocular network itself (MiDaS, a download) is out of scope: ``est`` is an input.

The reference gathers the valid pixels (``torch.where`` -> a host synchronisation) and solves the 2 x 2 normal equations with
``torch.inverse``; here the masked-out pixels stay in place as zero rows of ``H`` (the normal equations are the same sums), the 2 x 2
inverse is written out, and nothing leaves the device."""
from __future__ import annotations

import torch


def get_scale_shift_LS(est_depth, render_depth, mask=None):
    """(scale, shift) with ``scale * est + shift ~ 1 / render_depth`` over the pixels where ``mask`` holds and the inverse rendered
    depth is positive (utils/depth_utils.py:44-96, ``num_samples == -1``).  Returns two 1-element tensors on the inputs' device."""
    inv = 1.0 / render_depth                      # (the estimate is an inverse depth)
    if mask is not None:
        inv = torch.where(mask, inv, torch.zeros_like(inv))
    valid = inv > 0
    vf = valid.to(est_depth.dtype)
    H = torch.stack([(est_depth * vf).reshape(-1), vf.reshape(-1)], dim=1)          # zero rows where invalid
    z = torch.where(valid, inv, torch.zeros_like(inv)).reshape(-1, 1)
    A = H.t() @ H
    b = H.t() @ z
    det = A[0, 0] * A[1, 1] - A[0, 1] * A[1, 0]
    scale = (A[1, 1] * b[0] - A[0, 1] * b[1]) / det
    shift = (A[0, 0] * b[1] - A[1, 0] * b[0]) / det
    return scale, shift


def scale_depth_estimate(cfg, idx, est_depth, gt_depth, render_depth_sil, resumed=False):
    """``est_depth_scaled`` of slam/SLAM.py:411-448.  render_depth_sil: a callable returning (depth, silhouette) of the map at the
    frame's estimated pose (only called for idx > 0, or on a resumed run)."""
    with torch.no_grad():
        if idx == 0 and not resumed:
            if str(cfg.get("dataset", "")).lower() == "utmm":
                # "until visual-inertial initialization is implemented": the first estimate is fitted to the sensor depth
                scale, shift = get_scale_shift_LS(est_depth, gt_depth, gt_depth > 0)
                return 1.0 / (scale * est_depth + shift)
            return 1.0 / (est_depth + 0.001) * float(cfg["cam"]["png_depth_scale"]) / 10.0      # "arbitrarily scale the first frame"
        depth, sil = render_depth_sil()
        mask = (sil > 0.99) & (est_depth > 1e-6)
        scale, shift = get_scale_shift_LS(est_depth, depth, mask)
        return 1.0 / (scale * est_depth + shift)
