"""Host-side mirrors vs fixtures produced by the reference's own helpers (tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest
import torch

from mm3dgs_slam_amd import general_utils, graphics_utils, pose_utils, sh_utils

G = os.path.join(os.path.dirname(__file__), "golden")
HERE = os.path.dirname(os.path.abspath(__file__))


def load(name):
    return {k: torch.from_numpy(v) if v.dtype != object else v for k, v in np.load(os.path.join(G, name)).items()}


def close(a, b, tol=1e-5):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert (a - b).abs().max().item() <= tol * max(1.0, b.abs().max().item()), (a - b).abs().max().item()


def test_g1_pose_algebra():
    d = load("g1_pose.npz")
    poses = d["poses"]
    close(torch.stack([pose_utils.get_camera_from_tensor(p) for p in poses]), d["w2c"])
    close(torch.stack([pose_utils.get_tensor_from_camera(m) for m in d["w2c"]]), d["back"])
    close(pose_utils.quadmultiply(poses[:32, :4], poses[32:, :4]), d["qmul"])
    close(torch.stack([pose_utils.propagate_const_vel(poses[i], poses[i + 1]) for i in range(32)]), d["const_vel"], 2e-5)
    grads = []
    for p in poses[:16]:
        q = p.clone().requires_grad_(True)
        (pose_utils.get_camera_from_tensor(q) * torch.arange(16.0).reshape(4, 4)).sum().backward()
        grads.append(q.grad)
    close(torch.stack(grads), d["dpose"], 2e-5)


def test_g2_imu_propagation_and_euler():
    d = load("g2_imu.npz")
    close(torch.stack([pose_utils.euler_matrix(*a) for a in d["euler_in"]]), d["euler_out"])
    out = pose_utils.propagate_imu(d["camm1"], d["camm2"], d["imu"].clone(), d["c2i"], 1.0 / 30.0, 1.0 / 100.0)
    close(out, d["out"], 2e-5)


def test_g3_spherical_harmonics():
    d = load("g3_sh.npz")
    for deg in range(5):
        close(sh_utils.eval_sh(deg, d["sh"], d["dirs"]), d[f"deg{deg}"])
    close(sh_utils.RGB2SH(d["rgb"]), d["rgb2sh"])
    close(sh_utils.SH2RGB(d["rgb"]), d["sh2rgb"])


def test_g4_projection_matrix():
    d = load("g4_proj.npz")
    for row, P in zip(d["intr"], d["P"]):
        close(graphics_utils.getProjectionMatrix2(0.01, 100.0, *[float(v) for v in row[:4]], int(row[4]), int(row[5])), P)


def test_g5_covariance_helpers():
    d = load("g5_cov.npz")
    L = general_utils.build_scaling_rotation(d["s"], d["r"])
    close(L, d["L"])
    close(general_utils.strip_symmetric(L @ L.transpose(1, 2)), d["cov6"])
    close(general_utils.build_rotation(d["r"]), d["R"])
    close(general_utils.inverse_sigmoid(d["inv_sig_in"]), d["inv_sig"])


class _Recorder:
    calls = []

    def __init__(self, raster_settings):
        self.rs = raster_settings

    def __call__(self, **kw):
        _Recorder.calls.append((self.rs, kw))
        P = kw["means3D"].shape[0]
        return torch.zeros(3, self.rs.image_height, self.rs.image_width), torch.ones(P, dtype=torch.int32)


@pytest.mark.parametrize("tm", [True, False])
@pytest.mark.parametrize("iso", [True, False])
def test_g6_renderer_glue_feeds_the_rasterizer_what_the_reference_does(tm, iso):
    """Our Renderer (reference two-pass mode) must hand the rasterizer the same matrices and tensors as
    slam/renderer.py does for the same model and pose."""
    from mm3dgs_slam_amd.gaussian_model import GaussianModel
    from mm3dgs_slam_amd.renderer import Renderer
    d = load("g6_glue.npz")
    cfg = {"device": "cpu", "desired_height": 48, "desired_width": 64, "white_background": False,
           "cam": {"fx": 51.73, "fy": 51.65, "cx": 31.86, "cy": 25.53}, "mapping": {"sh_degree": 0},
           "pipeline": {"convert_SHs_python": False, "compute_cov3D_python": False, "transform_means_python": tm,
                        "force_isotropic": iso}}
    pc = GaussianModel(cfg)
    pc._xyz, pc._features_dc, pc._opacity = d["xyz"], d["f_dc"], d["opacity"]
    pc._features_rest = torch.zeros(d["xyz"].shape[0], 0, 3)
    pc._scaling, pc._rotation = d["scaling"], d["rotation"]
    _Recorder.calls = []
    Renderer(cfg, rasterizer_cls=_Recorder, mode="reference").render(pc, d["pose"])
    (rs, kw1), (_, kw2) = _Recorder.calls
    tag = f"tm{int(tm)}_iso{int(iso)}"
    close(rs.viewmatrix, d[f"{tag}_view"]); close(rs.projmatrix, d[f"{tag}_proj"]); close(rs.campos, d[f"{tag}_campos"], 1e-4)
    close(torch.tensor([rs.tanfovx, rs.tanfovy]), d[f"{tag}_tanfov"])
    close(kw1["means3D"], d[f"{tag}_means3D"]); close(kw1["scales"], d[f"{tag}_scales"])
    close(kw1["rotations"], d[f"{tag}_rotations"]); close(kw1["opacities"], d[f"{tag}_opacities"])
    close(kw1["shs"], d[f"{tag}_shs"]); close(kw2["colors_precomp"], d[f"{tag}_depthsil"], 2e-5)
    assert kw2.get("shs") is None


def test_g7_seeding_pointcloud():
    from mm3dgs_slam_amd.mapper import Mapper
    d = load("g7_seed.npz")
    fx, fy, cx, cy = [float(v) for v in d["intr"]]
    m = Mapper.__new__(Mapper)
    m.cfg = {"cam": {"fx": fx, "fy": fy, "cx": cx, "cy": cy}}
    mask = (d["depth"] > 0).reshape(-1)
    cld, msd = m.get_pointcloud(d["color"], d["depth"], pose_utils.get_camera_from_tensor(d["pose"]), mask=mask)
    close(cld, d["cld"], 2e-5); close(msd, d["msd"]); close(torch.log(torch.sqrt(msd)), d["log_scale"])


def test_numpy_constant_velocity_prediction_matches_the_torch_one():
    """Tracker.predict_pose runs propagate_const_vel_np on the host; it must agree with propagate_const_vel (itself pinned to
    the reference's utils/pose_utils.py:203-216 by the G1 fixture) for every quaternion branch."""
    import numpy as np
    import torch
    from mm3dgs_slam_amd.pose_utils import propagate_const_vel, propagate_const_vel_np
    gen = torch.Generator().manual_seed(11)
    for _ in range(200):
        a = torch.randn(7, generator=gen); b = a + 0.05 * torch.randn(7, generator=gen)
        a[:4] = a[:4] / a[:4].norm() * (0.5 + torch.rand(1, generator=gen)); b[:4] = b[:4] / b[:4].norm()
        want = propagate_const_vel(a.double(), b.double()).numpy()
        got = propagate_const_vel_np(a.numpy(), b.numpy())
        # q and -q are the same rotation: compare up to the sign convention of the branch
        if np.dot(want[:4], got[:4]) < 0:
            got = np.concatenate([-got[:4], got[4:]])
        assert np.abs(want - got).max() < 1e-5, (want, got)


def test_g8_losses_match_the_reference():
    """utils/loss_utils.py l1_loss / ssim / mapping photometric loss: values and autograd gradients (G8)."""
    from mm3dgs_slam_amd import loss_utils
    d = load("g8_loss.npz")
    img, gt, mask = d["img"], d["gt"], d["sil"] > 0.99

    def val_grad(fn):
        x = img.clone().requires_grad_(True)
        v = fn(x)
        v.backward()
        return v.detach(), x.grad

    for fn, kv, kg in ((lambda x: loss_utils.l1_loss(x, gt), "l1", "d_l1"),
                       (lambda x: loss_utils.l1_loss(x, gt, mask), "l1_masked", "d_l1_masked"),
                       (lambda x: loss_utils.ssim(x, gt), "ssim", "d_ssim"),
                       (lambda x: 0.8 * loss_utils.l1_loss(x, gt) + 0.2 * (1.0 - loss_utils.ssim(x, gt)), "map_photo", "d_map_photo")):
        v, g = val_grad(fn)
        close(v, d[kv], 2e-6)
        assert (g.double() - d[kg].double()).norm() <= 2e-5 * d[kg].double().norm(), kg
    close(loss_utils.ssim(img[None], gt[None], size_average=False), d["ssim_per_image"], 2e-6)


def test_g8_pearson_call_patterns():
    """pearson_loss (utils/loss_utils.py:43-61) as slam/tracker.py:127-144 and slam/mapper.py:859-873 call it."""
    from mm3dgs_slam_amd import loss_utils
    d = load("g8_loss.npz")
    depth, ref, mask = d["depth"], d["ref_depth"], d["sil"] > 0.99
    close(loss_utils.pearson_loss(depth, ref.clamp_min(0.5), mask=mask, invert_estimate=True), d["pearson_track_est"].float(), 2e-5)
    close(loss_utils.pearson_loss(depth, ref, mask=mask & (ref > 0), invert_estimate=True), d["pearson_track_gt"].float(), 2e-5)
    close(loss_utils.pearson_loss(depth, ref.clamp_min(0.5), invert_estimate=False), d["pearson_map_est"].float(), 2e-5)
    close(loss_utils.pearson_loss(depth, ref, mask=ref > 0, invert_estimate=False), d["pearson_map_gt"].float(), 2e-5)


def test_g8_rel_pose_loss():
    """rel_pose_loss (utils/loss_utils.py:20-40): values and gradients w.r.t. the current pose (slam/tracker.py:146-155)."""
    from mm3dgs_slam_amd import loss_utils
    d = load("g8_loss.npz")
    for i in range(d["rel_cur"].shape[0]):
        c = d["rel_cur"][i].clone().requires_grad_(True)
        t_l, q_l = loss_utils.rel_pose_loss(c, d["rel_init"][i])
        close(t_l.detach(), d["rel_t"][i], 1e-6)
        close(q_l.detach(), d["rel_q"][i], 2e-5)
        t_l.backward(retain_graph=True)
        close(c.grad, d["rel_dt"][i], 1e-5)
        c.grad = None
        q_l.backward()
        close(c.grad, d["rel_dq"][i], 2e-4)


# ---- G10: harness outputs (slam/SLAM.py:294-373 executed by tests/golden/make_golden_results.py) ---------------------------------
def test_g10_ate_rmse_matches_the_reference_umeyama_alignment():
    from mm3dgs_slam_amd.eval_utils import evaluate_ate_rmse
    F = np.load(os.path.join(HERE, "golden", "g10_results.npz"))
    for k in range(3):
        aligned, rmse = evaluate_ate_rmse(torch.from_numpy(F[f"ate{k}_est"]), torch.from_numpy(F[f"ate{k}_gt"]), method="umeyama")
        assert abs(rmse - float(F[f"ate{k}_rmse"])) < 1e-6 * max(1.0, float(F[f"ate{k}_rmse"])), (k, rmse, F[f"ate{k}_rmse"])
        a, b = np.asarray(aligned), F[f"ate{k}_aligned"]
        assert np.allclose(a[:, 4:], b[:, 4:], atol=1e-5)
        # quaternions up to sign (q and -q are the same rotation; the reference's rotation2quad picks a branch per matrix)
        assert np.allclose(np.abs((a[:, :4] * b[:, :4]).sum(1)), 1.0, atol=1e-5)


def test_g10_ate_rmse_matches_the_reference_horn_alignment():
    """method="horn" (utils/eval_utils.py:193-228,249-266): rigid alignment, no scale -- on the same three trajectories."""
    from mm3dgs_slam_amd.eval_utils import evaluate_ate_rmse
    F = np.load(os.path.join(HERE, "golden", "g10_results.npz"))
    for k in range(3):
        aligned, rmse = evaluate_ate_rmse(torch.from_numpy(F[f"ate{k}_est"]), torch.from_numpy(F[f"ate{k}_gt"]), method="horn")
        assert abs(rmse - float(F[f"ate{k}_horn_rmse"])) < 1e-6 * max(1.0, float(F[f"ate{k}_horn_rmse"])), (k, rmse, F[f"ate{k}_horn_rmse"])
        a, b = np.asarray(aligned), F[f"ate{k}_horn_aligned"]
        assert np.allclose(a[:, 4:], b[:, 4:], atol=1e-5)
        assert np.allclose(np.abs((a[:, :4] * b[:, :4]).sum(1)), 1.0, atol=1e-5)


def test_g10_save_results_writes_the_reference_key_set_and_values(tmp_path):
    """This repository's SLAM.save_results on the state the reference's own save_results was run on: same keys in the same order,
    same shapes / dtypes / values, keyframe dicts that slam/mapper.py:65-71 could read back (KeyFrame(**kf))."""
    import types
    from mm3dgs_slam_amd.mapper import KeyFrame
    from mm3dgs_slam_amd.slam import SLAM
    F = np.load(os.path.join(HERE, "golden", "g10_results.npz"))
    est, gt, last = torch.from_numpy(F["sr_est"]), torch.from_numpy(F["sr_gt"]), int(F["sr_last_idx"])
    kfs = [KeyFrame(int(i), torch.from_numpy(c), torch.from_numpy(p), torch.from_numpy(d), None)
           for i, c, p, d in zip(F["sr_kf_idx"], F["sr_kf_gt_color"], F["sr_kf_est_pose"], F["sr_kf_gt_depth"])]
    t_sum, t_n, m_sum, m_n = (float(v) for v in F["sr_timing_inputs"])
    ev = tuple([np.float32(v) for v in row] for row in F["sr_eval_lists"])
    fake = types.SimpleNamespace(cfg={"outputdir": str(tmp_path), "debug": {"create_video": False, "get_runtime_stats": True}},
                                 estimate_pose_list=list(est), gt_pose_list=list(gt),
                                 mapper=types.SimpleNamespace(keyframes=kfs, mapping_time_sum=m_sum, mapping_iter_count=int(m_n)),
                                 tracker=types.SimpleNamespace(tracking_time_sum=t_sum, tracking_iter_count=int(t_n)),
                                 evaluate_images=lambda last_idx: ev)
    SLAM.save_results(fake, last)
    res = np.load(tmp_path / "results.npz", allow_pickle=True)
    assert list(res.keys()) == [str(k) for k in F["keys"]]
    for k in ("pose_est", "pose_gt", "psnr_list", "ssim_list", "lpips_list"):
        assert res[k].shape == F["sr_" + k].shape and res[k].dtype == F["sr_" + k].dtype, (k, res[k].shape, res[k].dtype, F["sr_" + k].dtype)
        assert np.allclose(res[k], F["sr_" + k], atol=1e-7), k
    for k in ("ate_rmse", "avg_tracking_it_time", "avg_mapping_it_time"):
        assert res[k].shape == () and abs(float(res[k]) - float(F["sr_" + k])) < 1e-6 * max(1.0, abs(float(F["sr_" + k]))), (k, res[k], F["sr_" + k])
    kf_read = list(res["keyframes"])
    assert sorted(kf_read[0].keys()) == [str(k) for k in F["sr_kf_keys"]]
    assert [kf["idx"] for kf in kf_read] == [int(i) for i in F["sr_kf_idx"]]
    for kf, c, p, d, none in zip(kf_read, F["sr_kf_gt_color"], F["sr_kf_est_pose"], F["sr_kf_gt_depth"], F["sr_kf_est_depth_is_none"]):
        assert torch.is_tensor(kf["gt_color"]) and np.array_equal(kf["gt_color"].numpy(), c) and np.array_equal(kf["est_pose"].numpy(), p)
        assert np.array_equal(kf["gt_depth"].numpy(), d) and (kf["est_depth"] is None) == bool(none)


# ---- G11: per-frame depth alignment (utils/depth_utils.py:44-99 executed by tests/golden/make_golden_depth.py) -------------------------
def test_g11_depth_alignment_matches_the_reference_least_squares():
    """`use_gt_depth: false` (configs/TUM.yml:8): slam/SLAM.py:411-448 fits the monocular estimate to the rendered depth every frame.
    depth_utils.get_scale_shift_LS (masked normal equations, nothing leaves the device) against the reference's gather + torch.inverse
    version: scale and shift to 2e-4 relative (both are float32 solves of a 2 x 2 system with condition number ~1e6), the scaled
    depth image they produce to 1e-3."""
    from mm3dgs_slam_amd.depth_utils import get_scale_shift_LS
    F = np.load(os.path.join(HERE, "golden", "g11_depth_align.npz"))
    for k in range(3):
        est, depth, mask = (torch.from_numpy(F[f"c{k}_{n}"]) for n in ("est", "depth", "mask"))
        scale, shift = get_scale_shift_LS(est, depth, mask)
        rs, rt = float(F[f"c{k}_scale"].reshape(-1)[0]), float(F[f"c{k}_shift"].reshape(-1)[0])
        assert abs(float(scale) - rs) <= 2e-4 * abs(rs), (k, float(scale), rs)
        assert abs(float(shift) - rt) <= 2e-4 * abs(rt) + 1e-6, (k, float(shift), rt)
        scaled = 1.0 / (scale * est + shift)
        ref = torch.from_numpy(F[f"c{k}_scaled"])
        ok = mask & torch.isfinite(ref) & (ref.abs() < 50)
        assert ((scaled - ref).abs()[ok] <= 1e-3 * ref.abs()[ok] + 1e-4).all(), k


def test_depth_alignment_ignores_masked_out_garbage_and_survives_a_singular_fit():
    """a NaN / inf of the monocular estimate at a pixel OUTSIDE the mask must not reach the normal equations (the reference gathers
    the valid pixels only, utils/depth_utils.py:44-96), and a system without a solution (no valid pixel; a constant estimate) yields the identity
    fit instead of NaN that would flow into seeding and the Pearson target."""
    from mm3dgs_slam_amd.depth_utils import get_scale_shift_LS
    g = torch.Generator().manual_seed(3)
    depth = 1.0 + 3.0 * torch.rand(24, 32, generator=g)
    est = 0.7 / depth + 0.05                      # scale 0.7... of the INVERSE depth: est = a / z + b  <=>  1 / z = (est - b) / a
    mask = torch.rand(24, 32, generator=g) > 0.3
    clean = get_scale_shift_LS(est, depth, mask)
    dirty_est = est.clone()
    dirty_est[~mask] = float("nan")
    dirty_est[0, 0] = float("inf") if not bool(mask[0, 0]) else dirty_est[0, 0]
    dirty = get_scale_shift_LS(dirty_est, depth, mask)
    assert torch.isfinite(dirty[0]).all() and torch.isfinite(dirty[1]).all()
    assert torch.allclose(clean[0], dirty[0]) and torch.allclose(clean[1], dirty[1])
    assert torch.allclose(clean[0] * est + clean[1], 1.0 / depth, atol=1e-4)          # and the fit is the right one
    # no valid pixel / a constant estimate (0.25: exact sums; 0.3: float32 sums that do NOT cancel exactly -- ADVICE round 5 -- and a
    # nearly constant one, whose spread is below the inputs' own rounding): scale 1, shift 0, flagged
    nearly = torch.full_like(est, 0.3) * (1.0 + 2e-7 * torch.randn(24, 32, generator=g))
    for e, m in ((est, torch.zeros_like(mask)), (torch.full_like(est, 0.25), mask), (torch.full_like(est, 0.3), mask), (nearly, mask)):
        s, t, ok = get_scale_shift_LS(e, depth, m, return_ok=True)
        assert float(s) == 1.0 and float(t) == 0.0 and not bool(ok)
    assert bool(get_scale_shift_LS(est, depth, mask, return_ok=True)[2])
