"""Shared helpers for the HIP-vs-oracle parity tests (tests only; imports oracle/)."""
from __future__ import annotations

import torch

from mm3dgs_slam_amd import synthetic
from oracle.raster_ref import RefSettings, rasterize_ref

# Tolerances from BASELINE.json north_star: RGB/depth <= 1e-4 rel-L2; pose gradients <= 1e-5 ... the latter is
# meaningful against a float64 oracle only up to float32 evaluation noise of the whole chain, so pose/camera
# gradients are held to 1e-5 when compared with the float32 run of the same oracle and 1e-4 against float64.
IMG_TOL = 1e-4
GRAD_TOL = 2e-4      # per-Gaussian gradients (north_star states no bar for them; float32 evaluation noise of the chain)
POSE_TOL = 1e-5      # north_star: "pose gradients to <= 1e-5" -- applied to dL/d{viewmatrix, projmatrix, campos} and dL/dpose


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    den = b.norm().item()
    if den == 0:
        return a.norm().item()
    return (a - b).norm().item() / den


def make_case(P=2000, H=100, W=130, seed=0, sh_degree=None, posed=False, cov_precomp=False, extras=0, bg=(0.1, 0.2, 0.3),
              scale_modifier=1.0, log_scale=-3.0, spread=1.2, zmin=0.5, zmax=3.5):
    """Inputs in float64 on CPU (leaf tensors), plus camera tensors."""
    M = 0 if sh_degree is None else 16
    cloud, fx, fy = synthetic.random_cloud(P, H, W, seed=seed, dtype=torch.float64, sh_coeffs=M, log_scale=log_scale,
                                           spread=spread, zmin=zmin, zmax=zmax)
    w2c = synthetic.small_pose(seed + 1) if posed else None
    view, proj, campos, tanx, tany = synthetic.camera_matrices(H, W, fx, fy, w2c=w2c, dtype=torch.float64)
    case = dict(H=H, W=W, tanx=tanx, tany=tany, view=view, proj=proj, campos=campos, bg=torch.tensor(bg, dtype=torch.float64),
                sh_degree=0 if sh_degree is None else sh_degree, scale_modifier=scale_modifier,
                means3D=cloud["means3D"], opacities=cloud["opacities"], scales=cloud["scales"],
                rotations=cloud["rotations"], shs=None, colors=None, cov3D=None, extras=None)
    if sh_degree is None:
        case["colors"] = cloud["colors"]
    else:
        case["shs"] = cloud["shs"]
    if extras:
        g = torch.Generator().manual_seed(seed + 7)
        case["extras"] = torch.rand(P, extras, generator=g, dtype=torch.float64) * 2.0
    if cov_precomp:
        from oracle.raster_ref import cov3d_ref
        S = cov3d_ref(cloud["scales"], cloud["rotations"], 1.0)
        case["cov3D"] = torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], -1)
        case["scales"] = None
        case["rotations"] = None
    # every input is a float32-representable number (held in float64): the float64 oracle and the float32 kernels then see the
    # SAME inputs, so the comparison measures arithmetic, not the 1e-8 rounding of the inputs (which can flip a 1/255 or
    # T < 1e-4 decision and move a camera gradient by 5e-5 on its own)
    for k, v in list(case.items()):
        if torch.is_tensor(v) and v.is_floating_point():
            case[k] = v.float().double()
    return case


_TENSOR_KEYS = ("means3D", "opacities", "scales", "rotations", "shs", "colors", "cov3D", "extras", "view", "proj", "campos")


def leaves(case, dtype, device):
    out = {}
    for k in _TENSOR_KEYS:
        v = case[k]
        out[k] = None if v is None else v.detach().to(dtype=dtype, device=device).clone().requires_grad_(True)
    out["means2D"] = torch.zeros(case["means3D"].shape[0], 3, dtype=dtype, device=device, requires_grad=True)
    return out


def run_oracle(case, dtype=torch.float64, weights=None, need_grad=True, tiles=None, depth_key=None):
    """tiles: a list of tile ids -- the tile-sampled oracle (oracle.raster_ref.rasterize_tiles_ref): the image is zero outside them,
    and so must `weights` be for the gradients to mean anything (tile_weights)."""
    L = leaves(case, dtype, "cpu")
    s = RefSettings(case["H"], case["W"], case["tanx"], case["tany"], case["bg"].to(dtype), case["scale_modifier"],
                    L["view"], L["proj"], case["sh_degree"], L["campos"])
    cp = L["colors"]
    if L["extras"] is not None:
        cp = L["extras"] if cp is None else torch.cat([cp, L["extras"]], 1)
    img, radii, aux = rasterize_ref(L["means3D"], L["means2D"], L["opacities"], L["shs"], cp, L["scales"], L["rotations"],
                                    L["cov3D"], s, return_aux=True, tiles=tiles, depth_key=depth_key)
    grads = {}
    if need_grad:
        if weights is None:
            weights = loss_weights(img.shape, 123)
        (img * weights.to(dtype)).sum().backward()
        grads = {k: (v.grad if v is not None else None) for k, v in L.items()}
    return img.detach(), radii, aux, grads


def tile_mask(H, W, tiles):
    """[H,W] bool: the pixels of the 16x16 tiles `tiles` (ids ty * ceil(W / 16) + tx)."""
    gx = (W + 15) // 16
    m = torch.zeros(H, W, dtype=torch.bool)
    for t in tiles:
        tx, ty = int(t) % gx, int(t) // gx
        m[ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16] = True
    return m


CLEAN_PIXEL_TOL = 4e-6     # largest |difference| / (1 + |value|) over the pixels of a tile whose float32 decisions (1/255, T < 1e-4) all agree with float64
                           # (float32 accumulation: ~1e-6); ONE differing 1/255 decision moves its pixel by alpha c T ~ 2e-3 T
FLIP_TILE_TOL = 2e-2       # image rel-L2 of a tile that does carry such a decision


def tile_errors(img_a, img_b, tiles):
    """Per 16x16 tile of `tiles`: (rel-L2 of img_a against img_b ([C,H,W]), max over its pixels of |a - b| / (1 + |b|))."""
    H, W = img_b.shape[-2:]
    gx = (W + 15) // 16
    out = {}
    for t in tiles:
        tx, ty = int(t) % gx, int(t) // gx
        ys, xs = slice(ty * 16, (ty + 1) * 16), slice(tx * 16, (tx + 1) * 16)
        a, b = img_a[:, ys, xs].double(), img_b[:, ys, xs].double()
        out[int(t)] = (rel_l2(a, b), float(((a - b).abs() / (1.0 + b.abs())).max()) if a.numel() else 0.0)
    return out


def clean_tiles(errs):
    return [t for t, (_, px) in errs.items() if px <= CLEAN_PIXEL_TOL]


def pick_tiles(counts, n=32, seed=0):
    """Tile sample for the full-size parity tests: the heaviest tile (longest list), the lightest non-empty one, the four image
    corners (the bottom / right ones are partial tiles when H or W is not a multiple of 16) and seeded random ones, `n` in all.
    counts: [gy, gx] pairs per tile (oracle.raster_ref.tile_rects_ref)."""
    gy, gx = counts.shape
    flat = counts.reshape(-1)
    chosen = [int(flat.argmax())]
    nz = torch.nonzero(flat > 0).flatten()
    if nz.numel():
        chosen.append(int(nz[flat[nz].argmin()]))
    chosen += [0, gx - 1, (gy - 1) * gx, gy * gx - 1]
    g = torch.Generator().manual_seed(seed)
    for t in torch.randperm(gy * gx, generator=g).tolist():
        if len(set(chosen)) >= n:
            break
        chosen.append(int(t))
    out = []
    for t in chosen:
        if t not in out:
            out.append(t)
    return out[:n]


def loss_weights(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64)


def run_hip(case, weights=None, need_grad=True, device="cuda", gaussian_grads=True, camera_grads=True):
    from mm3dgs_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    L = leaves(case, torch.float32, device)
    if not gaussian_grads:
        for k in ("opacities", "scales", "rotations", "shs", "cov3D"):
            if L[k] is not None:
                L[k].requires_grad_(False)
    if not camera_grads:
        for k in ("view", "proj", "campos"):
            L[k].requires_grad_(False)
    rs = GaussianRasterizationSettings(case["H"], case["W"], case["tanx"], case["tany"],
                                       case["bg"].to(torch.float32).to(device), case["scale_modifier"], L["view"],
                                       L["proj"], case["sh_degree"], L["campos"], False, False)
    rast = GaussianRasterizer(rs)
    img, radii = rast(means3D=L["means3D"], means2D=L["means2D"], opacities=L["opacities"], shs=L["shs"],
                      colors_precomp=L["colors"], scales=L["scales"], rotations=L["rotations"],
                      cov3D_precomp=L["cov3D"], extra_channels=L["extras"])
    grads = {}
    if need_grad:
        if weights is None:
            weights = loss_weights(img.shape, 123)
        (img * weights.to(torch.float32).to(device)).sum().backward()
        grads = {k: (v.grad if v is not None else None) for k, v in L.items()}
    return img.detach(), radii, grads


def compare(case, verbose=False, **hip_kw):
    """Run both and return a dict of error metrics."""
    img_o, radii_o, aux, g_o = run_oracle(case)
    img_h, radii_h, g_h = run_hip(case, **hip_kw)
    m = {"img": rel_l2(img_h, img_o), "radii_mismatch": int((radii_h.cpu() != radii_o).sum()),
         "num_rendered": aux["num_rendered"], "visible": int((radii_o > 0).sum())}
    for k, go in g_o.items():
        gh = g_h.get(k)
        if go is None or gh is None:
            continue
        m["d_" + k] = rel_l2(gh, go)
    if verbose:
        print({k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in m.items()})
    m["case"] = case
    return m


def f32_floor(case):
    """Error of the float32 evaluation of the oracle itself against its float64 evaluation, per gradient."""
    _, _, _, g64 = run_oracle(case, torch.float64)
    _, _, _, g32 = run_oracle(case, torch.float32)
    return {"d_" + k: rel_l2(g32[k], g64[k]) for k in g64 if g64[k] is not None and g32[k] is not None}
