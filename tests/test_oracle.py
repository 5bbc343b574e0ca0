"""The CPU oracle itself: second statement of the compositing rule, finite-difference check of autograd, and the
three straight-through conventions."""
import torch

from oracle.raster_ref import RefSettings, dense_render_ref, rasterize_ref
from tests import parity_util as pu


def _settings(case, dt=torch.float64):
    return RefSettings(case["H"], case["W"], case["tanx"], case["tany"], case["bg"].to(dt), case["scale_modifier"],
                       case["view"].to(dt), case["proj"].to(dt), case["sh_degree"], case["campos"].to(dt))


def test_tiled_composite_equals_explicit_pixel_loop():
    case = pu.make_case(P=250, H=40, W=56, seed=3, posed=True)
    s = _settings(case)
    img, _ = rasterize_ref(case["means3D"], None, case["opacities"], None, case["colors"], case["scales"], case["rotations"], None, s)
    ref = dense_render_ref(case["means3D"], case["opacities"], case["colors"], case["scales"], case["rotations"], s)
    assert (img - ref).abs().max() < 1e-12


def test_tile_sampled_oracle_equals_the_whole_image_oracle_on_its_tiles():
    """rasterize_ref(..., tiles=[...]) -- what the full-size parity tests run (tests/test_gpu_fullsize.py) -- is the whole-image oracle
    restricted to those tiles: same pixels bit for bit, same radii, and for a gradient image that is zero outside the tiles the same
    gradients of every input (SH colours, a posed camera with view / projection / campos gradients, extra channels)."""
    from oracle.raster_ref import tile_rects_ref
    case = pu.make_case(P=900, H=70, W=100, seed=11, sh_degree=2, posed=True, extras=3)      # 7 x 5 tiles, the last row / column partial
    img_full, radii_full, aux_full, _ = pu.run_oracle(case, need_grad=False)
    s = _settings(case)
    cp = case["extras"]
    radii, rect, depth, counts = tile_rects_ref(case["means3D"], case["opacities"], case["shs"], cp, case["scales"], case["rotations"], None, s)
    assert torch.equal(radii, radii_full) and int(counts.sum()) == aux_full["num_rendered"]
    rg = aux_full["ranges"]
    assert torch.equal(counts.reshape(-1), rg[1:] - rg[:-1])
    tiles = pu.pick_tiles(counts, n=9, seed=1)
    assert len(tiles) == 9 and int(counts.reshape(-1).argmax()) in tiles and 34 in tiles
    m = pu.tile_mask(case["H"], case["W"], tiles)
    w = pu.loss_weights(img_full.shape, 7) * m
    img_t, radii_t, aux_t, g_t = pu.run_oracle(case, weights=w, tiles=tiles)
    _, _, _, g_f = pu.run_oracle(case, weights=w)
    assert torch.equal(radii_t, radii_full)
    assert torch.equal(img_t[:, m], img_full[:, m]) and float(img_t[:, ~m].abs().max()) == 0.0
    assert torch.equal(aux_t["n_contrib"][m], aux_full["n_contrib"][m]) and torch.equal(aux_t["final_T"][m], aux_full["final_T"][m])
    for t in tiles:
        assert torch.equal(aux_t["lists"][t], aux_full["point_list"][int(rg[t]):int(rg[t + 1])])
    for k, gf in g_f.items():
        if gf is None:
            continue
        assert g_t[k] is not None, k
        # (the per-Gaussian sums run over the same terms in the same order; the camera gradients sum over another set of Gaussians)
        assert pu.rel_l2(g_t[k], gf) < 1e-12, (k, pu.rel_l2(g_t[k], gf))
    assert float(g_f["means3D"].abs().max()) > 0


def test_autograd_matches_finite_differences():
    # sparse scene: no pixel saturates, so the T<1e-4 stop (a genuine discontinuity) is not crossed by the probe
    case = pu.make_case(P=40, H=32, W=32, seed=4, posed=True, log_scale=-2.8)
    w = pu.loss_weights((3, 32, 32), 5)

    def loss(means, scales, opac, view):
        s = RefSettings(32, 32, case["tanx"], case["tany"], case["bg"], 1.0, view, view @ (torch.linalg.inv(case["view"]) @ case["proj"]), 0, case["campos"])
        img, _ = rasterize_ref(means, None, opac, None, case["colors"], scales, case["rotations"], None, s)
        return (img * w).sum()

    args = [case["means3D"].clone().requires_grad_(True), case["scales"].clone().requires_grad_(True),
            case["opacities"].clone().requires_grad_(True), case["view"].clone().requires_grad_(True)]
    loss(*args).backward()
    g = torch.Generator().manual_seed(0)
    for i, a in enumerate(args):
        d = torch.randn(a.shape, generator=g, dtype=torch.float64)
        eps = 2e-7
        with torch.no_grad():
            plus = [x.detach() + (eps * d if j == i else 0) for j, x in enumerate(args)]
            minus = [x.detach() - (eps * d if j == i else 0) for j, x in enumerate(args)]
            fd = (loss(*plus) - loss(*minus)) / (2 * eps)
        an = (a.grad * d).sum()
        assert abs(fd - an) <= 2e-4 * max(1.0, abs(an)), (i, float(fd), float(an))


def test_alpha_clamp_is_straight_through():
    # one opaque splat in the middle of a 16x16 image: o*G > 0.99 at the centre, gradient wrt opacity must not vanish
    dt = torch.float64
    means = torch.tensor([[0.0, 0.0, 1.0]], dtype=dt)
    view = torch.eye(4, dtype=dt)
    from mm3dgs_slam_amd.synthetic import camera_matrices
    v, p, c, tx, ty = camera_matrices(16, 16, 16.0, 16.0, cx=7.5, cy=7.5, dtype=dt)
    s = RefSettings(16, 16, tx, ty, torch.zeros(3, dtype=dt), 1.0, v, p, 0, c)
    op = torch.tensor([[0.999]], dtype=dt, requires_grad=True)
    img, radii = rasterize_ref(means, None, op, None, torch.ones(1, 3, dtype=dt), torch.full((1, 3), 0.2, dtype=dt),
                               torch.tensor([[1.0, 0, 0, 0]], dtype=dt), None, s)
    assert abs(float(img.detach().max()) - 0.99) < 1e-9      # clamped at the centre pixel
    img.max().backward()
    assert op.grad.abs().item() > 0.5


def test_frustum_clamp_treats_clamped_coordinate_as_constant():
    dt = torch.float64
    from mm3dgs_slam_amd.synthetic import camera_matrices
    v, p, c, tx, ty = camera_matrices(32, 32, 16.0, 16.0, dtype=dt)
    s = RefSettings(32, 32, tx, ty, torch.zeros(3, dtype=dt), 1.0, v, p, 0, c)
    # far outside 1.3 * tanfov in x, but big enough to reach the image
    means = torch.tensor([[2.0, 0.0, 1.0]], dtype=dt, requires_grad=True)
    img, radii = rasterize_ref(means, None, torch.ones(1, 1, dtype=dt) * 0.9, None, torch.ones(1, 3, dtype=dt),
                               torch.full((1, 3), 0.6, dtype=dt), torch.tensor([[1.0, 0, 0, 0]], dtype=dt), None, s)
    assert int(radii[0]) > 0 and img.sum() > 0
    img.sum().backward()
    assert torch.isfinite(means.grad).all()


def test_background_and_extra_channels():
    case = pu.make_case(P=100, H=32, W=32, seed=6, extras=3, bg=(1.0, 1.0, 1.0))
    s = _settings(case)
    cp = torch.cat([case["colors"], case["extras"]], 1)
    img, _ = rasterize_ref(case["means3D"], None, case["opacities"], None, cp, case["scales"], case["rotations"], None, s)
    img3, _ = rasterize_ref(case["means3D"], None, case["opacities"], None, case["colors"], case["scales"], case["rotations"], None, s)
    assert torch.allclose(img[:3], img3) and img.shape[0] == 6
    zero_bg = s._replace(bg=torch.zeros(3, dtype=torch.float64))
    ex, _ = rasterize_ref(case["means3D"], None, case["opacities"], None, case["extras"], case["scales"], case["rotations"], None, zero_bg)
    assert torch.allclose(img[3:], ex)     # extra channels composite over a zero background


def test_argument_validation():
    import pytest
    case = pu.make_case(P=10, H=16, W=16, seed=7)
    s = _settings(case)
    with pytest.raises(ValueError):
        rasterize_ref(case["means3D"], None, case["opacities"], None, None, case["scales"], case["rotations"], None, s)
    with pytest.raises(ValueError):
        rasterize_ref(case["means3D"], None, case["opacities"], None, case["colors"], None, None, None, s)


# ---- the oracle's own pieces against the fixtures the reference's Python produced (tests/golden/make_golden.py) ---------
def _golden(name):
    import os
    import numpy as np
    return {k: torch.from_numpy(v) for k, v in np.load(os.path.join(os.path.dirname(__file__), "golden", name)).items()}


def test_oracle_sh_basis_matches_reference_eval_sh():
    """sh_to_rgb_ref (the basis the rasterizer oracle evaluates) == utils/sh_utils.py eval_sh, degrees 0..3 (G3)."""
    from oracle.raster_ref import sh_to_rgb_ref
    d = _golden("g3_sh.npz")
    sh = d["sh"].double().permute(0, 2, 1)[:, :16]            # reference layout [P,3,25] -> oracle layout [P,M,3]
    for deg in range(4):
        got = sh_to_rgb_ref(deg, sh, d["dirs"].double())
        assert (got - d[f"deg{deg}"].double()).abs().max() < 2e-6, deg


def test_oracle_rotation_and_covariance_match_reference_helpers():
    """quat_to_rot_ref / cov3d_ref == utils/general_utils.py build_rotation / build_scaling_rotation -> strip_symmetric (G5)."""
    from oracle.raster_ref import cov3d_ref, cov6_to_mat, quat_to_rot_ref
    d = _golden("g5_cov.npz")
    qn = torch.nn.functional.normalize(d["r"].double(), dim=1)      # the reference normalises inside build_rotation
    assert (quat_to_rot_ref(qn) - d["R"].double()).abs().max() < 2e-6
    S = cov3d_ref(d["s"].double(), qn, 1.0)
    assert (S - cov6_to_mat(d["cov6"].double())).abs().max() < 1e-5 * d["cov6"].abs().max()
    L = d["L"].double()
    assert (S - L @ L.transpose(1, 2)).abs().max() < 1e-5 * d["cov6"].abs().max()


def test_oracle_rotation_matches_reference_pose_algebra():
    """quat_to_rot_ref == the rotation block of utils/pose_utils.py get_camera_from_tensor (G1: 64 poses)."""
    from oracle.raster_ref import quat_to_rot_ref
    d = _golden("g1_pose.npz")
    qn = torch.nn.functional.normalize(d["poses"][:, :4].double(), dim=1)
    assert (quat_to_rot_ref(qn) - d["w2c"][:, :3, :3].double()).abs().max() < 2e-6


def test_oracle_projection_conventions_match_reference_glue():
    """The oracle fed with exactly what the reference's Renderer hands its rasterizer (G6, transform_means_python false:
    viewmatrix = w2c^T, projmatrix = viewmatrix @ P^T, campos) projects the means where the reference's own
    pose + projection matrix put them (utils/graphics_utils.py:85-94, slam/renderer.py:117-124)."""
    from oracle.raster_ref import RefSettings, preprocess_ref
    d = _golden("g6_glue.npz")
    g4 = _golden("g4_proj.npz")
    H, W = 48, 64
    tag = "tm0_iso0"
    dt = torch.float64
    s = RefSettings(H, W, float(d[f"{tag}_tanfov"][0]), float(d[f"{tag}_tanfov"][1]), torch.zeros(3, dtype=dt), 1.0,
                    d[f"{tag}_view"].to(dt), d[f"{tag}_proj"].to(dt), 0, d[f"{tag}_campos"].to(dt))
    means = d[f"{tag}_means3D"].to(dt)
    P = means.shape[0]
    pre = preprocess_ref(means, None, d[f"{tag}_opacities"].to(dt), None, torch.zeros(P, 3, dtype=dt), d[f"{tag}_scales"].to(dt),
                         d[f"{tag}_rotations"].to(dt), None, s)
    # independent statement with the reference's pose algebra (G1-pinned) and pinhole intrinsics
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
    w2c = get_camera_from_tensor(d["pose"]).to(dt)
    cam = means @ w2c[:3, :3].t() + w2c[:3, 3]
    fx, fy, cx, cy = 51.73, 51.65, 31.86, 25.53
    u = fx * cam[:, 0] / cam[:, 2] + cx
    v = fy * cam[:, 1] / cam[:, 2] + cy
    vis = cam[:, 2] > 0.2
    assert (pre["depth"][vis] - cam[vis, 2]).abs().max() < 1e-6
    # ndc -> pixel: ((ndc + 1) * S - 1) / 2 is the pixel-centre convention of a pinhole with principal point (cx, cy) - 0.5
    assert (pre["xy"][vis, 0] - (u[vis] - 0.5)).abs().max() < 2e-4
    assert (pre["xy"][vis, 1] - (v[vis] - 0.5)).abs().max() < 2e-4
    assert g4["P"].shape[0] == 4
