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


def get_scale_shift_LS(est_depth, render_depth, mask=None, return_ok=False):
    """(scale, shift) with ``scale * est + shift ~ 1 / render_depth`` over the pixels where ``mask`` holds and the inverse rendered
    depth is positive (utils/depth_utils.py:44-96, ``num_samples == -1``).  Returns two 1-element tensors on the inputs' device; with
    ``return_ok`` also a 0-dim bool tensor: whether the system had a fit (see below)."""
    inv = 1.0 / render_depth                      # (the estimate is an inverse depth)
    if mask is not None:
        inv = torch.where(mask, inv, torch.zeros_like(inv))
    valid = inv > 0
    # normal equations H^T H x = H^T z with H = [est, 1] over the valid pixels, as five masked sums (a [2, HW] x [HW, 2] product is a
    # skinny GEMM: 0.56 ms per call at 640x480 through hipBLASLt -- measured in the first round-4 trace -- against ~10 us per reduction)
    # (torch.where, not a product with the 0/1 mask: a NaN / inf of the network at a masked-out pixel must not reach the sums -- the
    #  reference gathers the valid pixels only)
    # The five sums and the 2 x 2 solve run in float64: det = a00 a11 - a01^2 cancels to ~1e-7 of its terms for a nearly
    # constant estimate, which is float32's own rounding -- a float32 det there is noise that passes any relative threshold.
    h = torch.where(valid, est_depth, torch.zeros_like(est_depth)).double()
    z = torch.where(valid, inv, torch.zeros_like(inv)).double()
    a00, a01, a11 = (h * h).sum(), h.sum(), valid.sum().double()
    b0, b1 = (h * z).sum(), z.sum()
    det = a00 * a11 - a01 * a01
    # a singular system (fewer than two valid pixels, or a constant estimate over them: the reference's torch.inverse raises there) has
    # no fit: the identity (scale 1, shift 0) is returned instead of NaN / inf that would silently flow into seeding and the Pearson
    # target -- decided on the device (no host read-back).  det / (a00 a11) = the estimate's variance over its mean square: below 1e-9
    # (a spread of 3e-5 of the mean) the inputs' own float32 rounding decides the fit.
    ok = (a11 >= 2) & (det.abs() > 1e-9 * (a00 * a11).abs().clamp_min(1e-300)) & torch.isfinite(det)
    safe = torch.where(ok, det, torch.ones_like(det))
    scale = torch.where(ok, (a11 * b0 - a01 * b1) / safe, torch.ones_like(det)).to(est_depth.dtype).reshape(1)
    shift = torch.where(ok, (a00 * b1 - a01 * b0) / safe, torch.zeros_like(det)).to(est_depth.dtype).reshape(1)
    return (scale, shift, ok) if return_ok else (scale, shift)


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
