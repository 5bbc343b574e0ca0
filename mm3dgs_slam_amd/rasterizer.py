"""Python surface of the reference's missing ``diff_gaussian_rasterization`` extension, rebuilt on libmm3dgs_hip.so.

Mirrors what ``slam/renderer.py`` consumes (reference file:line):

* ``GaussianRasterizationSettings`` -- the 12-field record keyword-constructed at ``slam/renderer.py:125-138``
  (``tanfovx/tanfovy`` may be 0-dim tensors because ``slam/SLAM.py:65-69`` stores intrinsics as tensors);
* ``GaussianRasterizer(raster_settings=...)`` -- callable returning ``(color[C,H,W], radii[P])`` with keyword
  arguments ``means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp``
  (``slam/renderer.py:140,196-214``); exactly one of ``shs``/``colors_precomp`` and one of
  (``scales``,``rotations``)/``cov3D_precomp``, otherwise an ``Exception`` is raised before any launch;
* autograd: gradients for every tensor input including the ``means2D`` sink (x,y = screen-space gradient,
  ``slam/gaussian_model.py:594-598``) and -- the "-w-pose" behaviour -- ``viewmatrix``, ``projmatrix``, ``campos``
  (``slam/renderer.py:115-124``).

Extension over the lineage (used by the fused SLAM render, one pass instead of the two at ``renderer.py:196-214``):
``extra_channels=[P,E]`` may accompany ``shs`` (or ``colors_precomp``), giving ``3+E <= 6`` output channels that share
projection, binning and sorting.  Extra channels composite over a zero background.

There is no CPU path here: tensors must live on a HIP device and the shared library must be present.
"""
from __future__ import annotations

import ctypes as C
from typing import NamedTuple, Optional

import torch

from . import _lib

MAX_CHANNELS = 6


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


# ---------------------------------------------------------------------------------------------------------------------
# binning capacity policy
#   "exact": stage 1 -> read num_rendered back (one 4-byte D2H + stream sync, as the lineage does) -> exact buffer.
#   "async": no host sync; capacity = headroom x P x the largest pairs-per-Gaussian ratio seen for this image size (the first
#            call for a size is sized exactly); the count of every call lands in pinned memory asynchronously and is
#            checked on the NEXT call.  An overflow raises then (the overflowing render itself was incomplete but
#            memory-safe) and the capacity is grown.
_policy = {"mode": "exact", "headroom": 2.0}
_capacity_cache = {}
_pending = []  # [(event, pinned_hdr, capacity, key)]


_last = {}


def last_header():
    """Counters of the most recent forward (synchronises): num_rendered, overflow, max_tile_len."""
    img = _last.get("img")
    if img is None:
        return None
    h = img[:32].view(torch.int32).cpu()
    return dict(num_rendered=int(h[0]), overflow=int(h[1]), max_tile_len=int(h[2]), fwd_wave_iters=int(h[4]),
                bwd_wave_iters=int(h[5]), bwd_wave_visits=int(h[6]))


def last_depths():
    """View depths [P] (float32) the most recent forward sorted its tile lists by (geom_state: csrc/mm3dgs_common.h geom_view); the entries
    of culled Gaussians (radii == 0) are unwritten.  Diagnostics / tests."""
    geom, P = _last.get("geom"), _last.get("P", 0)
    if geom is None:
        return None
    off = (P * 48 + 255) // 256 * 256
    return geom[off:off + 4 * P].view(torch.float32)


def set_binning_policy(mode: str = "exact", headroom: float = 2.0):
    if mode not in ("exact", "async"):
        raise ValueError("mode must be 'exact' or 'async'")
    _policy["mode"] = mode
    _policy["headroom"] = float(headroom)


def get_binning_policy():
    return dict(_policy)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32c(t: Optional[torch.Tensor], name: str):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor: mm3dgs_slam_amd has no CPU rasterizer (got device {t.device})")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _camera(rs: GaussianRasterizationSettings, bg, view, proj, campos):
    cam = _lib.Mm3dgsCamera()
    cam.image_height = int(rs.image_height)
    cam.image_width = int(rs.image_width)
    cam.tanfovx = float(rs.tanfovx)
    cam.tanfovy = float(rs.tanfovy)
    cam.scale_modifier = float(rs.scale_modifier)
    cam.sh_degree = int(rs.sh_degree)
    cam.prefiltered = int(bool(rs.prefiltered))
    cam.debug = int(bool(rs.debug))
    cam.bg = bg.data_ptr()
    cam.viewmatrix = view.data_ptr()
    cam.projmatrix = proj.data_ptr()
    cam.campos = campos.data_ptr()
    return cam


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _drain_pending(block: bool = False):
    keep = []
    for ev, hdr, cap, (key, npts) in _pending:
        if block:
            ev.synchronize()
        if ev.query():
            n = int(hdr[0])
            _capacity_cache[key] = max(_capacity_cache.get(key, 0.0), n / npts)
            if n > cap:
                _pending.clear()
                raise RuntimeError(
                    f"mm3dgs: an earlier async render overflowed its binning capacity ({n} > {cap}); that image was "
                    f"incomplete. Capacity has been raised; re-render, or use set_binning_policy('exact').")
        else:
            keep.append((ev, hdr, cap, (key, npts)))
    _pending[:] = keep


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, viewmatrix,
                projmatrix, campos, raster_settings):
        lib = _lib.load()
        rs = raster_settings
        dev = means3D.device
        means3D = _f32c(means3D, "means3D")
        shs = _f32c(shs, "shs")
        colors_precomp = _f32c(colors_precomp, "colors_precomp")
        opacities = _f32c(opacities, "opacities")
        scales = _f32c(scales, "scales")
        rotations = _f32c(rotations, "rotations")
        cov3D_precomp = _f32c(cov3D_precomp, "cov3D_precomp")
        view = _f32c(viewmatrix, "viewmatrix")
        proj = _f32c(projmatrix, "projmatrix")
        cpos = _f32c(campos, "campos")
        bg = _f32c(rs.bg, "bg").reshape(-1)
        P = int(means3D.shape[0])
        M = int(shs.shape[1]) if shs is not None else 0
        n_extra = int(colors_precomp.shape[1]) if colors_precomp is not None else 0
        Cn = (3 if shs is not None else 0) + n_extra
        H, W = int(rs.image_height), int(rs.image_width)
        cam = _camera(rs, bg, view, proj, cpos)

        u8 = dict(dtype=torch.uint8, device=dev)
        out = torch.empty((Cn, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        geom = torch.empty((lib.mm3dgs_geom_bytes(P),), **u8)
        img = torch.empty((lib.mm3dgs_image_bytes(H, W),), **u8)
        st = _stream()
        args = (P, M, Cn, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities), _ptr(scales), _ptr(rotations),
                _ptr(cov3D_precomp))
        with torch.cuda.device(dev):
            key = (H, W)
            if _policy["mode"] == "async":
                _drain_pending()
            seen = _capacity_cache.get(key)
            if _policy["mode"] == "exact" or seen is None:
                # exact sizing (also the first async call for an image size, to learn N): one 4-byte read-back
                host_n = torch.empty((4,), dtype=torch.int32).pin_memory()
                _lib.check(lib.mm3dgs_forward_geom(C.byref(cam), *args, _ptr(radii), _ptr(geom), _ptr(img),
                                                   C.c_void_p(host_n.data_ptr()), st))
                torch.cuda.current_stream().synchronize()
                n_cap = max(int(host_n[0]), 1)
                _capacity_cache[key] = max(seen or 0.0, n_cap / max(P, 1))     # pairs per Gaussian
                binning = torch.empty((lib.mm3dgs_binning_bytes(n_cap),), **u8)
                _lib.check(lib.mm3dgs_forward_raster(C.byref(cam), P, Cn, _ptr(geom), _ptr(img), _ptr(binning), n_cap,
                                                     _ptr(out), st))
            else:
                n_cap = int(seen * max(P, 1) * _policy["headroom"]) + 65536
                binning = torch.empty((lib.mm3dgs_binning_bytes(n_cap),), **u8)
                _lib.check(lib.mm3dgs_forward(C.byref(cam), *args, _ptr(out), _ptr(radii), _ptr(geom), _ptr(img),
                                              _ptr(binning), n_cap, st))
                hdr = torch.empty((4,), dtype=torch.int32).pin_memory()
                hdr.copy_(img[:16].view(torch.int32), non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                _pending.append((ev, hdr, n_cap, (key, max(P, 1))))
        _last["img"], _last["geom"], _last["P"] = img, geom, P
        ctx.rs = rs
        ctx.dims = (P, M, Cn, n_cap)
        ctx.save_for_backward(means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view, proj, cpos,
                              bg, radii, geom, img, binning)
        ctx.mark_non_differentiable(radii)
        return out, radii

    @staticmethod
    def backward(ctx, grad_out, _grad_radii):
        lib = _lib.load()
        (means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, view, proj, cpos, bg, radii, geom,
         img, binning) = ctx.saved_tensors
        rs = ctx.rs
        P, M, Cn, n_cap = ctx.dims
        dev = means3D.device
        grad_out = _f32c(grad_out, "grad_out")
        cam = _camera(rs, bg, view, proj, cpos)
        need = ctx.needs_input_grad
        # inputs: 0 means3D 1 means2D 2 shs 3 colors 4 opac 5 scales 6 rots 7 cov3D 8 view 9 proj 10 campos
        f32 = dict(dtype=torch.float32, device=dev)
        gaussian_side = need[2] or need[4] or need[5] or need[6] or need[7]
        d_means3D = torch.empty((P, 3), **f32)
        d_means2D = torch.empty((P, 3), **f32)
        d_shs = torch.empty((P, M, 3), **f32) if (shs is not None and need[2]) else None
        d_colors = torch.empty_like(colors_precomp) if colors_precomp is not None else None
        d_opac = torch.empty((P, 1), **f32) if gaussian_side else None
        d_scales = torch.empty((P, 3), **f32) if (scales is not None and gaussian_side) else None
        d_rots = torch.empty((P, 4), **f32) if (rotations is not None and gaussian_side) else None
        d_cov = torch.empty((P, 6), **f32) if (cov3D_precomp is not None and gaussian_side) else None
        d_view = torch.empty((4, 4), **f32) if need[8] else None
        d_proj = torch.empty((4, 4), **f32) if need[9] else None
        d_cpos = torch.empty((3,), **f32) if need[10] else None
        scratch = torch.empty((lib.mm3dgs_backward_scratch_bytes(P, n_cap),), dtype=torch.uint8, device=dev)
        flags = 0 if gaussian_side else 1
        with torch.cuda.device(dev):
            _lib.check(lib.mm3dgs_backward(
                C.byref(cam), P, M, Cn, _ptr(means3D), _ptr(shs), _ptr(colors_precomp), _ptr(opacities), _ptr(scales),
                _ptr(rotations), _ptr(cov3D_precomp), _ptr(radii), _ptr(geom), _ptr(img), _ptr(binning), n_cap,
                _ptr(grad_out), _ptr(scratch), _ptr(d_means3D), _ptr(d_means2D), _ptr(d_shs), _ptr(d_colors),
                _ptr(d_opac), _ptr(d_scales), _ptr(d_rots), _ptr(d_cov), _ptr(d_view), _ptr(d_proj), _ptr(d_cpos),
                flags, _stream()))
        return (d_means3D, d_means2D, d_shs, d_colors, d_opac if need[4] else None, d_scales if need[5] else None,
                d_rots if need[6] else None, d_cov if need[7] else None, d_view, d_proj, d_cpos, None)


def rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                        raster_settings):
    rs = raster_settings
    return _RasterizeGaussians.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                     rs.viewmatrix, rs.projmatrix, rs.campos, rs)


class GaussianRasterizer(torch.nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Boolean mask of points in front of the camera's near cut (z_view > 0.2)."""
        lib = _lib.load()
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions, "positions")
            view = _f32c(rs.viewmatrix, "viewmatrix")
            proj = _f32c(rs.projmatrix, "projmatrix")
            cpos = _f32c(rs.campos, "campos")
            bg = _f32c(rs.bg, "bg")
            cam = _camera(rs, bg, view, proj, cpos)
            vis = torch.empty((pos.shape[0],), dtype=torch.uint8, device=pos.device)
            with torch.cuda.device(pos.device):
                _lib.check(lib.mm3dgs_mark_visible(C.byref(cam), int(pos.shape[0]), _ptr(pos), _ptr(vis), _stream()))
        return vis.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extra_channels=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or (
                (scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        if extra_channels is not None:
            colors_precomp = extra_channels if colors_precomp is None else torch.cat([colors_precomp, extra_channels], 1)
        n_ch = (3 if shs is not None else 0) + (0 if colors_precomp is None else colors_precomp.shape[1])
        if n_ch < 1 or n_ch > MAX_CHANNELS:
            raise Exception(f"channel count {n_ch} outside 1..{MAX_CHANNELS}")
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   self.raster_settings)
