"""Render boundary: ``Renderer.render(pc, camera_pose) -> dict`` (reference ``slam/renderer.py:46-224``).

Same constructor inputs (cfg keys ``desired_height/width``, ``cam.{fx,fy,cx,cy}``, ``white_background``, ``device``,
``pipeline.*``), same returned keys (``render``, ``depth`` = [z, silhouette, z^2], ``viewspace_points``,
``visibility_filter``, ``radii``), same gradient paths.  Two execution modes:

* ``mode="reference"`` -- literally the reference's sequence: two rasterizer calls (RGB, then the depth bundle as
  precomputed colours, ``slam/renderer.py:196-214``);
* ``mode="fused"`` (default) -- one 6-channel rasterizer call: both passes share projection, binning, sorting and
  compositing state, which is the whole point of owning the kernel.  Results are identical (the two passes use the
  same geometry; ``means2D.grad`` is the sum over both passes in the reference, ``renderer.py:156,198,209``, and the
  fused pass produces that sum directly).

Deliberate deviations, each behind a flag that defaults to the reference's behaviour:
* ``pipeline.compute_cov3D_python``: the reference computes ``cov3D_precomp`` but never passes it
  (``renderer.py:164-165`` vs ``:196-214``), which makes that flag unusable; here it is passed.
* quaternions are not rotated in ``transform_means_python`` mode, exactly like ``renderer.py:152,171-173``.

The rasterizer class is injectable (tests hand in the CPU oracle); the default is the HIP one and there is no fallback.
"""
from __future__ import annotations

import torch

from .graphics_utils import getProjectionMatrix2
from .pose_utils import apply_rigid, get_camera_from_tensor
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from .sh_utils import eval_sh


def get_depth_and_silhouette(means3D, w2c):
    """[z_cam, 1, z_cam^2] per Gaussian, evaluated at its centre (``slam/renderer.py:26-43``)."""
    z = (means3D * w2c[2, :3]).sum(1) + w2c[2, 3]      # (no BLAS: a [P,3] x [3] product is a 100 us gemv launch)
    return torch.stack([z, torch.ones_like(z), z * z], 1)


class Renderer:
    def __init__(self, cfg, rasterizer_cls=None, settings_cls=None, mode: str = "fused"):
        self.cfg = cfg
        self.mode = mode
        self.rasterizer_cls = rasterizer_cls or GaussianRasterizer
        self.settings_cls = settings_cls or GaussianRasterizationSettings
        self.zfar, self.znear = 100.0, 0.01
        self.image_height = int(cfg["desired_height"])
        self.image_width = int(cfg["desired_width"])
        cam = cfg["cam"]
        self.cx, self.cy, self.fovx, self.fovy = (float(cam[k]) for k in ("cx", "cy", "fx", "fy"))
        self.tanfovx = self.image_width / (2 * self.fovx)
        self.tanfovy = self.image_height / (2 * self.fovy)
        dev = cfg["device"]
        self.projection_matrix = getProjectionMatrix2(self.znear, self.zfar, self.fovx, self.fovy, self.cx, self.cy,
                                                      self.image_height, self.image_width).t().contiguous().to(dev)
        self.background = torch.tensor([1.0, 1.0, 1.0] if cfg["white_background"] else [0.0, 0.0, 0.0], device=dev)
        self._eye = torch.eye(4, device=dev)

    def render(self, pc, camera_pose, scaling_modifier=1.0, override_color=None):
        pipe = self.cfg["pipeline"]
        xyz = pc.get_xyz
        screenspace_points = torch.zeros_like(xyz, requires_grad=True) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        rel_w2c = get_camera_from_tensor(camera_pose)
        if pipe["transform_means_python"]:
            w2c = self._eye
            projmatrix = self.projection_matrix
            camera_pos = torch.zeros(3, device=xyz.device)
            means3D = apply_rigid(xyz, rel_w2c)
        else:
            w2c = rel_w2c.t()
            projmatrix = w2c @ self.projection_matrix
            camera_pos = torch.linalg.inv(w2c)[3, :3]
            means3D = xyz
        settings = self.settings_cls(image_height=self.image_height, image_width=self.image_width, tanfovx=self.tanfovx,
                                     tanfovy=self.tanfovy, bg=self.background, scale_modifier=scaling_modifier,
                                     viewmatrix=w2c, projmatrix=projmatrix, sh_degree=pc.active_sh_degree,
                                     campos=camera_pos, prefiltered=False, debug=False)
        rasterizer = self.rasterizer_cls(raster_settings=settings)
        opacity = pc.get_opacity
        scales = rotations = cov3D_precomp = None
        if pipe["compute_cov3D_python"]:
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales = torch.exp(pc._scaling[:, :1]).expand(-1, 3) if pipe["force_isotropic"] else pc.get_scaling
            rotations = pc.get_rotation
        shs = colors_precomp = None
        if override_color is not None:
            colors_precomp = override_color
        elif pipe["convert_SHs_python"]:
            n_coef = (pc.max_sh_degree + 1) ** 2
            shs_view = pc.get_features.transpose(1, 2).reshape(-1, 3, n_coef)
            dirs = torch.nn.functional.normalize(pc.get_xyz - camera_pos[None, :], dim=1)
            colors_precomp = torch.clamp_min(eval_sh(pc.active_sh_degree, shs_view, dirs) + 0.5, 0.0)
        else:
            shs = pc.get_features
        # the depth bundle is evaluated with the `viewmatrix` variable exactly as renderer.py:207-214 does: identity
        # when the means were pre-transformed (the shipped configs); in the other mode the reference hands the
        # TRANSPOSED matrix to a column-vector product -- reproduced literally unless pipeline.fix_depth_transpose.
        dmat = w2c.t() if (not pipe["transform_means_python"] and pipe.get("fix_depth_transpose", False)) else w2c
        depth_sil = get_depth_and_silhouette(means3D, dmat)
        common = dict(means3D=means3D, means2D=screenspace_points, opacities=opacity, scales=scales, rotations=rotations,
                      cov3D_precomp=cov3D_precomp)
        if self.mode == "fused":
            out, radii = rasterizer(shs=shs, colors_precomp=colors_precomp, extra_channels=depth_sil, **common)
            rendered_image, rendered_depth = out[:3], out[3:6]
            if self.cfg["white_background"]:      # the reference composites the depth bundle over bg as well
                rendered_depth = rendered_depth + (1.0 - out[4:5]) * self.background[:, None, None]
        else:
            rendered_image, radii = rasterizer(shs=shs, colors_precomp=colors_precomp, **common)
            rendered_depth, _ = rasterizer(colors_precomp=depth_sil, **common)
        return {"render": rendered_image, "depth": rendered_depth, "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0, "radii": radii}
