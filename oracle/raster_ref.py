"""CPU oracle for the differentiable 3D-Gaussian tile rasterizer.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this file.
The product path (``mm3dgs_slam_amd``) never imports it and has no CPU fallback.

PARITY UNPINNED: the arithmetic this restates lives in the reference's un-vendored submodule
``submodules/diff-gaussian-rasterization`` (``/root/reference/.gitmodules:1-3``,
github.com/codeysun/diff-gaussian-rasterization-w-pose, commit unknown, directory empty).  The reference
holds no test or golden vector for it.  What *is* pinned by reference Python and is
followed here:

* call contract, argument shapes, ``[z, 1, z^2]`` depth bundle, background add, ``radii > 0`` visibility:
  ``slam/renderer.py:85-224``
* SH basis / signs and the ``+0.5`` / clamp: ``utils/sh_utils.py:57-112``, ``slam/renderer.py:188-189``
* quaternion order (w,x,y,z) and R(q): ``utils/general_utils.py:78-99``
* cov3D = (R S)(R S)^T as 6 upper-triangular values: ``utils/general_utils.py:64-76,101-110``,
  ``slam/gaussian_model.py:33-37``
* row-vector (transposed) view / projection matrices: ``slam/renderer.py:117-124``

Everything else is the published 3D-Gaussian-Splatting tile rasterization algorithm,
restated in vectorised PyTorch so that ``torch.autograd`` provides the backward pass independently of the
hand-derived HIP backward.  Works in float32 or float64 (dtype follows ``means3D``).

Three deliberate "as the CUDA lineage does it" choices that plain autograd would do differently are
implemented with straight-through terms and are each covered by a test:

* alpha = min(0.99, o*G): the gradient is passed as if un-clamped;
* the +-1.3*tanfov clamp of t.x/t.z, t.y/t.z inside the EWA Jacobian: a clamped coordinate is a constant;
* integer decisions (radius, tile rectangle, 1/255 skip, T<1e-4 stop, depth order) carry no gradient.
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

TILE = 16
SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
         -0.4570457994644658, 1.445305721320277, -0.5900435899266435)


class RefSettings(NamedTuple):
    """Same twelve fields, same meaning, as the record built at ``slam/renderer.py:125-138``."""
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
    prefiltered: bool = False
    debug: bool = False


def sh_to_rgb_ref(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """SH colour before the +0.5/clamp.  sh: [P, M, 3] (coefficient-major, as ``get_features``), dirs [P,3] unit.
    Basis follows utils/sh_utils.py:57-112 (degrees 0..3, which is what the rasterizer kernel supports)."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6] + SH_C2[3] * xz * sh[:, 7]
                   + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9] + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13] + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res


def quat_to_rot_ref(q: torch.Tensor) -> torch.Tensor:
    """R(q) for q = (w,x,y,z), *not* normalised here (the caller normalises, slam/gaussian_model.py:116-118)."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rows = [
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)], -1),
        torch.stack([2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)], -1),
        torch.stack([2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1),
    ]
    return torch.stack(rows, 1)


def cov3d_ref(scales: torch.Tensor, rotations: torch.Tensor, mod: float) -> torch.Tensor:
    """[P,3,3] covariance (R S)(R S)^T, S = diag(mod * scales)."""
    R = quat_to_rot_ref(rotations)
    M = R * (mod * scales)[:, None, :]
    return M @ M.transpose(1, 2)


def cov6_to_mat(c6: torch.Tensor) -> torch.Tensor:
    """6 upper-triangular values [xx,xy,xz,yy,yz,zz] -> symmetric [P,3,3] (utils/general_utils.py:64-76)."""
    xx, xy, xz, yy, yz, zz = c6.unbind(-1)
    return torch.stack([torch.stack([xx, xy, xz], -1), torch.stack([xy, yy, yz], -1),
                        torch.stack([xz, yz, zz], -1)], 1)


def preprocess_ref(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                   s: RefSettings):
    """Per-Gaussian projection stage.  Returns a dict of differentiable screen-space quantities plus the
    integer radius / tile rectangle decisions."""
    dt = means3D.dtype
    P = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    tanx, tany = float(s.tanfovx), float(s.tanfovy)
    V = s.viewmatrix.to(dt)
    PV = s.projmatrix.to(dt)
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    t = ph @ V[:, :3]                      # row-vector convention: p_view = [p,1] . V
    hom = ph @ PV
    pw = 1.0 / (hom[:, 3] + 1e-7)
    ndc = hom[:, :2] * pw[:, None]
    if means2D is not None:                # gradient sink only: value-neutral
        ndc = ndc + (means2D[:, :2] - means2D[:, :2].detach()).to(dt)
    in_front = t[:, 2] > 0.2

    if cov3D_precomp is not None:
        S3 = cov6_to_mat(cov3D_precomp.to(dt))
    else:
        S3 = cov3d_ref(scales.to(dt), rotations.to(dt), float(s.scale_modifier))

    tz = torch.where(in_front, t[:, 2], torch.ones_like(t[:, 2]))   # keep culled rows finite
    fx = W / (2.0 * tanx)
    fy = H / (2.0 * tany)
    limx, limy = 1.3 * tanx, 1.3 * tany
    txtz, tytz = t[:, 0] / tz, t[:, 1] / tz
    in_x = (txtz >= -limx) & (txtz <= limx)
    in_y = (tytz >= -limy) & (tytz <= limy)
    txc = torch.where(in_x, t[:, 0], (txtz.clamp(-limx, limx) * tz).detach())
    tyc = torch.where(in_y, t[:, 1], (tytz.clamp(-limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([torch.stack([fx / tz, zero, -fx * txc / (tz * tz)], -1),
                     torch.stack([zero, fy / tz, -fy * tyc / (tz * tz)], -1)], 1)       # [P,2,3]
    Wr = V[:3, :3].t()                                                                  # world -> view rotation
    A = J @ Wr
    cov2 = A @ S3 @ A.transpose(1, 2)
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    ok = in_front & (det != 0)
    det_safe = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / det_safe, -b / det_safe, a / det_safe], -1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam.detach())).to(torch.int64)
    px = ((ndc[:, 0] + 1.0) * W - 1.0) * 0.5
    py = ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    rf = radius.to(dt)
    pxd, pyd = px.detach(), py.detach()

    def _clampi(v, hi):
        return torch.clamp(torch.trunc(v).to(torch.int64), 0, hi)

    rminx = _clampi((pxd - rf) / TILE, gx)
    rminy = _clampi((pyd - rf) / TILE, gy)
    rmaxx = _clampi((pxd + rf + TILE - 1) / TILE, gx)
    rmaxy = _clampi((pyd + rf + TILE - 1) / TILE, gy)
    tiles = (rmaxx - rminx) * (rmaxy - rminy)
    ok = ok & (tiles > 0) & torch.isfinite(pxd) & torch.isfinite(pyd)
    tiles = torch.where(ok, tiles, torch.zeros_like(tiles))
    radii = torch.where(ok, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is not None and shs is None:
        colors = colors_precomp.to(dt)
    else:
        d = means3D - s.campos.to(dt)[None, :]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(sh_to_rgb_ref(int(s.sh_degree), shs.to(dt), d) + 0.5, 0.0)
        colors = rgb if colors_precomp is None else torch.cat([rgb, colors_precomp.to(dt)], 1)
    return dict(xy=torch.stack([px, py], -1), depth=t[:, 2].detach(), conic=conic,
                opacity=opacities.to(dt).reshape(P), colors=colors, radii=radii, tiles=tiles,
                rect=(rminx, rminy, rmaxx, rmaxy), ok=ok, grid=(gx, gy))


def bin_ref(pre):
    """(tile, depth, id)-ordered duplicate list and per-tile [start, end) ranges."""
    gx, gy = pre["grid"]
    rminx, rminy, rmaxx, rmaxy = pre["rect"]
    ids = torch.nonzero(pre["tiles"] > 0).flatten()
    if ids.numel() == 0:
        return torch.zeros(0, dtype=torch.int64), torch.zeros(gx * gy + 1, dtype=torch.int64)
    cnt = pre["tiles"][ids]
    rep = torch.repeat_interleave(ids, cnt)
    first = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(rep.numel()) - torch.repeat_interleave(first, cnt)
    w = (rmaxx - rminx)[rep]
    ty = rminy[rep] + local // w
    tx = rminx[rep] + local % w
    tile = ty * gx + tx
    o1 = torch.sort(pre["depth"][rep], stable=True).indices      # depth, ties keep Gaussian-index order
    o2 = torch.sort(tile[o1], stable=True).indices
    order = o1[o2]
    tile_sorted = tile[order]
    ranges = torch.searchsorted(tile_sorted, torch.arange(gx * gy + 1))
    return rep[order], ranges


def _composite_tile(pre, ids, tx, ty, bg):
    """One 16x16 tile: front-to-back compositing of the depth-ordered splats `ids` over `bg`.  Returns the tile's colours
    [C,16,16], its final transmittance [16,16] and, per pixel, the list position of the last contributor + 1 [16,16]."""
    dt = pre["xy"].dtype
    C = pre["colors"].shape[1]
    lx = torch.arange(TILE, dtype=dt)
    pxs = (tx * TILE + lx)[None, :].expand(TILE, TILE).reshape(-1)
    pys = (ty * TILE + lx)[:, None].expand(TILE, TILE).reshape(-1)
    xy = pre["xy"][ids]
    con = pre["conic"][ids]
    op = pre["opacity"][ids]
    dx = xy[None, :, 0] - pxs[:, None]
    dy = xy[None, :, 1] - pys[:, None]
    power = -0.5 * (con[None, :, 0] * dx * dx + con[None, :, 2] * dy * dy) - con[None, :, 1] * dx * dy
    G = torch.exp(torch.clamp(power, max=0.0))
    araw = op[None, :] * G
    alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()      # straight-through 0.99 clamp
    valid = (power <= 0) & (alpha.detach() >= 1.0 / 255.0)
    a = torch.where(valid, alpha, torch.zeros_like(alpha))
    Tincl = torch.cumprod(1.0 - a, dim=1)
    Text = torch.cat([torch.ones(a.shape[0], 1, dtype=dt), Tincl], 1)      # T before j, and after the last
    done_here = valid & (Tincl.detach() < 1e-4)
    done_cum = torch.cumsum(done_here.to(torch.int32), 1) > 0
    active = valid & ~done_cum
    wgt = torch.where(active, a * Text[:, :-1], torch.zeros_like(a))
    col = wgt @ pre["colors"][ids]
    L = a.shape[1]
    any_done = done_cum[:, -1]
    first_done = torch.argmax(done_cum.to(torch.int32), 1)
    stop = torch.where(any_done, first_done, torch.full_like(first_done, L))
    Tfin = torch.gather(Text, 1, stop[:, None])[:, 0]
    out = col + Tfin[:, None] * bg[None, :]
    pos = torch.arange(1, L + 1, dtype=torch.int32)[None, :]
    last = torch.max(torch.where(active, pos, torch.zeros_like(pos)), 1).values
    return out.t().reshape(C, TILE, TILE), Tfin.detach().reshape(TILE, TILE), last.reshape(TILE, TILE)


def _background(pre, s: RefSettings):
    dt = pre["xy"].dtype
    C = pre["colors"].shape[1]
    bg = s.bg.to(dt).reshape(-1)
    if bg.numel() < C:                       # extra channels (fused depth bundle) get a zero background
        bg = torch.cat([bg, torch.zeros(C - bg.numel(), dtype=dt)])
    return bg


def composite_ref(pre, point_list, ranges, s: RefSettings):
    """Front-to-back alpha compositing, one 16x16 tile at a time.  Returns image [C,H,W], final_T [H,W],
    n_contrib [H,W] (position in the tile list of the last contributor + 1)."""
    dt = pre["xy"].dtype
    H, W = int(s.image_height), int(s.image_width)
    gx, gy = pre["grid"]
    C = pre["colors"].shape[1]
    bg = _background(pre, s)
    rows = []
    finalT = torch.ones(gy * TILE, gx * TILE, dtype=dt)
    ncontrib = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int32)
    for ty in range(gy):
        row = []
        for tx in range(gx):
            tid = ty * gx + tx
            lo, hi = int(ranges[tid]), int(ranges[tid + 1])
            if hi == lo:
                row.append((bg[:, None, None]).expand(C, TILE, TILE))
                continue
            out, Tfin, last = _composite_tile(pre, point_list[lo:hi], tx, ty, bg)
            row.append(out)
            finalT[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = Tfin
            ncontrib[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = last
        rows.append(torch.cat(row, 2))
    img = torch.cat(rows, 1)[:, :H, :W]
    return img.contiguous(), finalT[:H, :W].contiguous(), ncontrib[:H, :W].contiguous()


def _take(t, idx):
    return None if t is None else t[idx]


def tile_rects_ref(means3D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, s: RefSettings, chunk=1 << 18):
    """Integer decisions of the projection stage for ALL Gaussians, without autograd and in chunks (3 M Gaussians with 16 SH
    coefficients in float64 fit comfortably): radii [P] int32, tile rectangle (minx, miny, maxx, maxy) [P] int64 each, view depth
    [P], and the number of (tile, splat) pairs of every tile [gy, gx] (a 2-D difference array over the rectangles)."""
    P = means3D.shape[0]
    radii, rect, depth = [], [[], [], [], []], []
    with torch.no_grad():
        for a in range(0, max(P, 1), chunk):
            sl = slice(a, min(a + chunk, P))
            pre = preprocess_ref(means3D[sl], None, opacities[sl], _take(shs, sl), _take(colors_precomp, sl), _take(scales, sl),
                                 _take(rotations, sl), _take(cov3D_precomp, sl), s)
            radii.append(pre["radii"]); depth.append(pre["depth"])
            live = pre["tiles"] > 0
            for k in range(4):       # (a culled Gaussian has the empty rectangle)
                rect[k].append(torch.where(live, pre["rect"][k], torch.zeros_like(pre["rect"][k])))
            grid = pre["grid"]
    radii, depth = torch.cat(radii), torch.cat(depth)
    rminx, rminy, rmaxx, rmaxy = (torch.cat(r) for r in rect)
    gx, gy = grid
    diff = torch.zeros((gy + 1) * (gx + 1), dtype=torch.int64)
    for (yy, xx, sign) in ((rminy, rminx, 1), (rminy, rmaxx, -1), (rmaxy, rminx, -1), (rmaxy, rmaxx, 1)):
        diff.index_add_(0, yy * (gx + 1) + xx, torch.full_like(yy, sign))
    counts = diff.reshape(gy + 1, gx + 1).cumsum(0).cumsum(1)[:gy, :gx]
    return radii, (rminx, rminy, rmaxx, rmaxy), depth, counts


def rasterize_tiles_ref(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, s: RefSettings, tiles,
                        depth_key=None):
    """The rasterizer restricted to the 16x16 tiles `tiles` (tile ids ty * gx + tx), for inputs too large for the whole-image oracle
    (the BASELINE configurations: 640x480 / 150 k ... 1920x1080 / 3 M Gaussians).  The projection stage's integer decisions are taken
    for ALL Gaussians (tile_rects_ref: radii are complete); the differentiable projection, the (tile, depth, id) order and the
    compositing run for the chosen tiles only, over exactly the splats whose tile rectangle holds them -- per tile the same
    arithmetic as rasterize_ref (every step of the rule is per Gaussian or per tile; both go through preprocess_ref /
    _composite_tile).  Returns (image [C,H,W] with zeros outside the chosen tiles, radii [P], aux); gradients flow to the full
    input tensors through the index selection, so a loss over the chosen tiles' pixels yields the same gradients as the whole-image
    oracle would for a gradient image that is zero elsewhere.

    depth_key ([P], optional): the values the (depth, id) order is taken from instead of this evaluation's own view depths.  A map
    seeded from a surface packs the hundreds of splats of a tile into centimetres of depth; neighbours in the order are then closer
    than float32 resolves, and ANY float32 rasterizer -- the CUDA lineage's as much as the one under test -- breaks those near-ties
    by its own last-bit rounding.  The full-size tests hand in the float32 depths the implementation under test sorted by (after
    checking them against this evaluation's float64 depths, to float32 rounding), so that what is compared is the arithmetic under the
    same integer decision -- "no gradient through integer decisions" holds for the order either way."""
    H, W = int(s.image_height), int(s.image_width)
    radii, (rminx, rminy, rmaxx, rmaxy), depth, counts = tile_rects_ref(means3D, opacities, shs, colors_precomp, scales, rotations,
                                                                         cov3D_precomp, s)
    gy, gx = counts.shape
    if callable(tiles):      # chosen from the pair counts (tests pick the heaviest tile, ...)
        tiles = tiles(counts)
    tiles = [int(t) for t in tiles]
    touch = torch.zeros(means3D.shape[0], dtype=torch.bool)
    for t in tiles:
        tx, ty = t % gx, t // gx
        touch |= (rminx <= tx) & (tx < rmaxx) & (rminy <= ty) & (ty < rmaxy)
    sel = torch.nonzero(touch).flatten()            # ascending: the subset keeps the Gaussian-index order (ties of the depth sort)
    pre = preprocess_ref(means3D[sel], _take(means2D, sel), opacities[sel], _take(shs, sel), _take(colors_precomp, sel), _take(scales, sel),
                         _take(rotations, sel), _take(cov3D_precomp, sel), s)
    assert torch.equal(pre["radii"], radii[sel])
    C = pre["colors"].shape[1]
    bg = _background(pre, s)
    img = torch.zeros(C, gy * TILE, gx * TILE, dtype=means3D.dtype)
    finalT = torch.ones(gy * TILE, gx * TILE, dtype=means3D.dtype)
    ncontrib = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int32)
    lists = {}
    prx0, pry0, prx1, pry1 = pre["rect"]
    for t in tiles:
        tx, ty = t % gx, t // gx
        ysl, xsl = slice(ty * TILE, (ty + 1) * TILE), slice(tx * TILE, (tx + 1) * TILE)
        inside = torch.nonzero((pre["tiles"] > 0) & (prx0 <= tx) & (tx < prx1) & (pry0 <= ty) & (ty < pry1)).flatten()
        if inside.numel() == 0:
            img[:, ysl, xsl] = bg[:, None, None].expand(C, TILE, TILE)
            lists[t] = sel[inside]
            continue
        key = pre["depth"] if depth_key is None else depth_key[sel]
        ids = inside[torch.sort(key[inside], stable=True).indices]      # depth, ties keep Gaussian-index order
        assert int(counts[ty, tx]) == ids.numel()
        out, Tfin, last = _composite_tile(pre, ids, tx, ty, bg)
        img[:, ysl, xsl] = out
        finalT[ysl, xsl] = Tfin
        ncontrib[ysl, xsl] = last
        lists[t] = sel[ids]
    mask = torch.zeros(gy * TILE, gx * TILE, dtype=torch.bool)
    for t in tiles:
        tx, ty = t % gx, t // gx
        mask[ty * TILE:(ty + 1) * TILE, tx * TILE:(tx + 1) * TILE] = True
    aux = dict(tiles=tiles, tile_mask=mask[:H, :W].contiguous(), tile_counts=counts, touched=sel, lists=lists, depth=depth,
               final_T=finalT[:H, :W].contiguous(), n_contrib=ncontrib[:H, :W].contiguous(), num_rendered=int(counts.sum()))
    return img[:, :H, :W].contiguous(), radii, aux


def rasterize_ref(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None, settings: Optional[RefSettings] = None, return_aux: bool = False, tiles=None, depth_key=None):
    """Oracle counterpart of ``GaussianRasterizer.forward`` as called at ``slam/renderer.py:196-214``.

    Extension used by the fused path: when *both* ``shs`` and ``colors_precomp`` are given, channels are
    ``[rgb(SH) | colors_precomp]`` and extra channels get a zero background.

    ``tiles``: a list of tile ids (ty * ceil(W / 16) + tx), or a callable that picks them from the [gy, gx] pair counts -- composite only those tiles (rasterize_tiles_ref: the sizes of
    BASELINE.json's configurations); pixels outside them are returned as zeros."""
    if shs is None and colors_precomp is None:
        raise ValueError("Please provide excatly one of either SHs or precomputed colors!")
    if (scales is None or rotations is None) == (cov3D_precomp is None):
        raise ValueError("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    if tiles is not None:
        img, radii, aux = rasterize_tiles_ref(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                                              settings, tiles, depth_key=depth_key)
        return (img, radii, aux) if return_aux else (img, radii)
    pre = preprocess_ref(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp,
                         settings)
    plist, ranges = bin_ref(pre)
    img, finalT, ncontrib = composite_ref(pre, plist, ranges, settings)
    if return_aux:
        return img, pre["radii"], dict(pre=pre, point_list=plist, ranges=ranges, final_T=finalT,
                                       n_contrib=ncontrib, num_rendered=int(plist.numel()))
    return img, pre["radii"]


class RefRasterizer(torch.nn.Module):
    """Drop-in shaped like ``diff_gaussian_rasterization.GaussianRasterizer`` but running the CPU oracle.
    Injected explicitly by tests / the cpu_baseline leg; never selected automatically."""

    def __init__(self, raster_settings, tiles=None, depth_key=None):
        super().__init__()
        self.raster_settings = raster_settings
        self.depth_key = depth_key      # rasterize_tiles_ref: the values the depth order is taken from
        self.tiles = tiles      # None: the whole image; a list of tile ids (or a callable on the pair counts): rasterize_tiles_ref

    def markVisible(self, positions):
        V = self.raster_settings.viewmatrix.to(positions.dtype)
        z = positions @ V[:3, 2] + V[3, 2]
        return z > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, extra_channels=None):
        if extra_channels is not None:
            colors_precomp = extra_channels if colors_precomp is None else torch.cat([colors_precomp, extra_channels], 1)
        rs = self.raster_settings
        s = RefSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy), rs.bg,
                        float(rs.scale_modifier), rs.viewmatrix, rs.projmatrix, int(rs.sh_degree), rs.campos,
                        bool(rs.prefiltered), bool(rs.debug))
        return rasterize_ref(means3D, means2D, opacities, shs, colors_precomp, scales, rotations, cov3D_precomp, s, tiles=self.tiles,
                             depth_key=self.depth_key)


def dense_render_ref(means3D, opacities, colors, scales, rotations, s: RefSettings):
    """Independent second statement of the compositing rule used to cross-check ``composite_ref``:
    an explicit per-pixel Python loop (tiny cases only)."""
    pre = preprocess_ref(means3D, None, opacities, None, colors, scales, rotations, None, s)
    plist, ranges = bin_ref(pre)
    H, W = int(s.image_height), int(s.image_width)
    gx, _ = pre["grid"]
    C = colors.shape[1]
    out = torch.zeros(C, H, W, dtype=means3D.dtype)
    for y in range(H):
        for x in range(W):
            tid = (y // TILE) * gx + (x // TILE)
            T = 1.0
            acc = [0.0] * C
            for k in range(int(ranges[tid]), int(ranges[tid + 1])):
                g = int(plist[k])
                dx = float(pre["xy"][g, 0]) - x
                dy = float(pre["xy"][g, 1]) - y
                A, B, Cc = (float(v) for v in pre["conic"][g])
                power = -0.5 * (A * dx * dx + Cc * dy * dy) - B * dx * dy
                if power > 0:
                    continue
                alpha = min(0.99, float(pre["opacity"][g]) * math.exp(power))
                if alpha < 1.0 / 255.0:
                    continue
                test_T = T * (1 - alpha)
                if test_T < 1e-4:
                    break
                for ch in range(C):
                    acc[ch] += float(pre["colors"][g, ch]) * alpha * T
                T = test_T
            for ch in range(C):
                out[ch, y, x] = acc[ch] + T * float(s.bg[ch])
    return out
