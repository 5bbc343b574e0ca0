"""Generates tests/golden/*.npz by IMPORTING the reference's own Python helpers from /root/reference (run in the
build container only; the reference never travels).  Inputs are seeded; outputs are what the reference computes.
Nothing here copies reference source: fixtures are data (inputs + expected outputs).

    python tests/golden/make_golden.py

G1 pose algebra      utils/pose_utils.py: get_camera_from_tensor, get_tensor_from_camera, quadmultiply,
                     propagate_const_vel, d(sum W2C)/dpose
G2 propagate_imu     utils/pose_utils.py:148-200 with synthetic IMU rows and tf/tf.txt extrinsics
G3 SH                utils/sh_utils.py: eval_sh deg 0..3 (+4), RGB2SH
G4 projection        utils/graphics_utils.py:getProjectionMatrix2 at four intrinsics
G5 covariance        utils/general_utils.py: build_scaling_rotation -> strip_symmetric, build_rotation
G6 render glue       slam/renderer.py Renderer.render with a recording stub rasterizer: the exact kwargs / settings
                     the reference hands to the rasterizer for transform_means_python x force_isotropic
G7 seeding           slam/mapper.py get_pointcloud + the scale/opacity/rotation initialisation on a 32x24 RGB-D
G8 losses            utils/loss_utils.py: l1_loss (:64-68, with and without mask), ssim (:95-154) and their autograd
                     gradients w.r.t. the rendered image, the mapping photometric loss 0.8 L1 + 0.2 (1 - SSIM)
                     (slam/mapper.py:856-858) with its gradient, rel_pose_loss (:20-40) values and gradients, and
                     pearson_loss (:43-61).  torchmetrics is not installed here: the reference's `pearson_corrcoef`
                     import is bound to scipy.stats.pearsonr (an independent third-party implementation of the same
                     textbook definition torchmetrics.functional.regression.pearson_corrcoef computes), so the
                     Pearson rows pin the reference's masking / min-of-two-targets logic, not torchmetrics' arithmetic.

G9 (end-to-end runs of the reference's Tracker / Mapper / GaussianModel / Renderer classes) has its own generator:
tests/golden/make_golden_slam.py.

    python tests/golden/make_golden.py            # everything
    python tests/golden/make_golden.py g8         # only the named fixture(s)
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)


class _CpuMode(torch.overrides.TorchFunctionMode):
    """The reference hard-codes device='cuda' / .cuda(); run it on CPU without touching it."""

    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if kwargs.get("device") is not None and str(kwargs["device"]).startswith("cuda"):
            kwargs["device"] = "cpu"
        if func is torch.Tensor.cuda:
            return args[0]
        return func(*args, **kwargs)


def stub_modules():
    calls = []

    class Settings:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    class Rasterizer:
        def __init__(self, raster_settings):
            self.rs = raster_settings

        def __call__(self, **kw):
            calls.append((self.rs, kw))
            P = kw["means3D"].shape[0]
            return torch.zeros(3, self.rs.image_height, self.rs.image_width), torch.ones(P, dtype=torch.int32)

    dgr = types.ModuleType("diff_gaussian_rasterization")
    dgr.GaussianRasterizationSettings = Settings
    dgr.GaussianRasterizer = Rasterizer
    sys.modules["diff_gaussian_rasterization"] = dgr
    for name in ("plyfile", "cv2", "pyiqa", "torchvision", "torchvision.utils", "torchmetrics", "torchmetrics.functional",
                 "torchmetrics.functional.regression", "imageio", "natsort"):
        m = types.ModuleType(name)
        sys.modules.setdefault(name, m)
    sys.modules["plyfile"].PlyData = sys.modules["plyfile"].PlyElement = object
    def _pearson(a, b):
        from scipy.stats import pearsonr
        return torch.tensor(pearsonr(a.detach().double().numpy().ravel(), b.detach().double().numpy().ravel())[0], dtype=torch.float64)
    sys.modules["torchmetrics.functional.regression"].pearson_corrcoef = _pearson
    return calls


def t2n(x):
    return x.detach().cpu().numpy()


def make_g8():
    """Loss fixture: the reference's own loss functions on seeded 48x64 images / poses."""
    from utils import loss_utils
    g = torch.Generator().manual_seed(8)
    H, W = 48, 64
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    base = torch.stack([0.5 + 0.4 * torch.sin(xx / 7.0 + c) * torch.cos(yy / 5.0 - c) for c in range(3)])
    gt = (base + 0.05 * torch.randn(3, H, W, generator=g)).clamp(0, 1)
    img = (base + 0.08 * torch.randn(3, H, W, generator=g) + 0.03).clamp(0, 1)
    sil = (torch.rand(H, W, generator=g) * 0.2 + 0.85).clamp(0, 1)
    depth = 2.0 + 0.5 * torch.sin(xx / 9.0) + 0.02 * torch.randn(H, W, generator=g)
    ref_depth = depth + 0.05 * torch.randn(H, W, generator=g) + 0.1 * torch.cos(yy / 6.0)
    ref_depth[torch.rand(H, W, generator=g) < 0.08] = 0
    out = dict(img=t2n(img), gt=t2n(gt), sil=t2n(sil), depth=t2n(depth), ref_depth=t2n(ref_depth))
    mask = sil > 0.99
    x = img.clone().requires_grad_(True)
    l1 = loss_utils.l1_loss(x, gt); l1.backward()
    out["l1"], out["d_l1"] = t2n(l1), t2n(x.grad)
    x = img.clone().requires_grad_(True)
    l1m = loss_utils.l1_loss(x, gt, mask); l1m.backward()
    out["l1_masked"], out["d_l1_masked"] = t2n(l1m), t2n(x.grad)
    x = img.clone().requires_grad_(True)
    s = loss_utils.ssim(x, gt); s.backward()
    out["ssim"], out["d_ssim"] = t2n(s), t2n(x.grad)
    out["ssim_per_image"] = t2n(loss_utils.ssim(img[None], gt[None], size_average=False))
    x = img.clone().requires_grad_(True)
    lam = 0.2
    photo = (1.0 - lam) * loss_utils.l1_loss(x, gt) + lam * (1.0 - loss_utils.ssim(x, gt)); photo.backward()
    out["map_photo"], out["d_map_photo"], out["lambda_dssim"] = t2n(photo), t2n(x.grad), np.float32(lam)
    # Pearson: values only (the stub is not differentiable); both call patterns of the reference
    out["pearson_track_est"] = t2n(loss_utils.pearson_loss(depth, ref_depth.clamp_min(0.5), mask=mask, invert_estimate=True))
    out["pearson_track_gt"] = t2n(loss_utils.pearson_loss(depth, ref_depth, mask=mask & (ref_depth > 0), invert_estimate=True))
    out["pearson_map_est"] = t2n(loss_utils.pearson_loss(depth, ref_depth.clamp_min(0.5), invert_estimate=False))
    out["pearson_map_gt"] = t2n(loss_utils.pearson_loss(depth, ref_depth, mask=ref_depth > 0, invert_estimate=False))
    # rel_pose_loss (slam/tracker.py:146-155 call pattern: current pose with grad, initial pose constant)
    poses = torch.randn(16, 7, generator=g)
    poses[:, :4] = torch.nn.functional.normalize(poses[:, :4], dim=1)
    cur = poses.clone()
    cur[:, :4] = torch.nn.functional.normalize(cur[:, :4] + 0.05 * torch.randn(16, 4, generator=g), dim=1) * (0.9 + 0.2 * torch.rand(16, 1, generator=g))
    cur[:, 4:] += 0.03 * torch.randn(16, 3, generator=g)
    tl, ql, dt, dq = [], [], [], []
    for c, i in zip(cur, poses):
        c1 = c.clone().requires_grad_(True)
        t_l, q_l = loss_utils.rel_pose_loss(c1, i)
        tl.append(t_l.detach()); ql.append(q_l.detach())
        t_l.backward(retain_graph=True); dt.append(c1.grad.clone()); c1.grad = None
        q_l.backward(); dq.append(c1.grad.clone())
    out.update(rel_cur=t2n(cur), rel_init=t2n(poses), rel_t=t2n(torch.stack(tl)), rel_q=t2n(torch.stack(ql)),
               rel_dt=t2n(torch.stack(dt)), rel_dq=t2n(torch.stack(dq)))
    np.savez(os.path.join(OUT, "g8_loss.npz"), **out)


def main():
    calls = stub_modules()
    torch.manual_seed(0)
    from utils import general_utils, graphics_utils, pose_utils, sh_utils
    only = [a.lower() for a in sys.argv[1:]]
    if only:
        with _CpuMode():
            for name in only:
                {"g8": make_g8}[name]()
        print("written:", only)
        return

    with _CpuMode():
        # ---- G1
        g = torch.Generator().manual_seed(0)
        poses = torch.randn(64, 7, generator=g)
        poses[:, :4] = torch.nn.functional.normalize(poses[:, :4], dim=1) * (0.8 + 0.4 * torch.rand(64, 1, generator=g))
        w2c = torch.stack([pose_utils.get_camera_from_tensor(p) for p in poses])
        back = torch.stack([pose_utils.get_tensor_from_camera(m) for m in w2c])
        qmul = pose_utils.quadmultiply(poses[:32, :4], poses[32:, :4])
        cv = torch.stack([pose_utils.propagate_const_vel(poses[i], poses[i + 1]) for i in range(0, 32)])
        grads = []
        for p in poses[:16]:
            q = p.clone().requires_grad_(True)
            (pose_utils.get_camera_from_tensor(q) * torch.arange(16.0).reshape(4, 4)).sum().backward()
            grads.append(q.grad)
        np.savez(os.path.join(OUT, "g1_pose.npz"), poses=t2n(poses), w2c=t2n(w2c), back=t2n(back), qmul=t2n(qmul),
                 const_vel=t2n(cv), dpose=t2n(torch.stack(grads)))
        # ---- G2
        tf = np.loadtxt(os.path.join(REF, "tf", "tf.txt"), comments="#")
        tf = np.atleast_2d(tf)[-1]
        c2i_pose = torch.tensor([tf[6], tf[3], tf[4], tf[5], tf[0], tf[1], tf[2]], dtype=torch.float32)  # (qw qx qy qz t)
        c2i = pose_utils.get_camera_from_tensor(c2i_pose)
        imu = torch.zeros(8, 30)
        imu[:, 13:16] = torch.randn(8, 3, generator=g) * 0.05
        imu[:, 25:28] = torch.randn(8, 3, generator=g) * 0.3 + torch.tensor([0.0, -9.8, 0.0])
        p1, p2 = poses[0] / 1.0, poses[1] / 1.0
        p1 = torch.cat([torch.nn.functional.normalize(p1[:4], dim=0), p1[4:] * 0.1])
        p2 = torch.cat([torch.nn.functional.normalize(p1[:4] + 0.01 * p2[:4], dim=0), p1[4:] + 0.01])
        imu_in = imu.clone()
        prop = pose_utils.propagate_imu(p1, p2, imu, c2i, 1.0 / 30.0, 1.0 / 100.0)
        em = torch.stack([pose_utils.euler_matrix(*a, axes="sxyz") for a in (torch.tensor([0.1, -0.2, 0.3]), torch.tensor([1.0, 2.0, 3.0]))])
        np.savez(os.path.join(OUT, "g2_imu.npz"), c2i=t2n(c2i), imu=t2n(imu_in), camm1=t2n(p1), camm2=t2n(p2), out=t2n(prop),
                 euler_in=np.array([[0.1, -0.2, 0.3], [1.0, 2.0, 3.0]], dtype=np.float32), euler_out=t2n(em))
        # ---- G3
        dirs = torch.nn.functional.normalize(torch.randn(256, 3, generator=g), dim=1)
        sh = torch.randn(256, 3, 25, generator=g)
        ev = {f"deg{d}": t2n(sh_utils.eval_sh(d, sh, dirs)) for d in range(5)}
        rgb = torch.rand(64, 3, generator=g)
        np.savez(os.path.join(OUT, "g3_sh.npz"), dirs=t2n(dirs), sh=t2n(sh), rgb=t2n(rgb), rgb2sh=t2n(sh_utils.RGB2SH(rgb)),
                 sh2rgb=t2n(sh_utils.SH2RGB(rgb)), **ev)
        # ---- G4
        intr = np.array([[517.3, 516.5, 318.6, 255.3, 480, 640], [457.1, 457.3, 324.4, 166.5, 330, 640],
                         [600.0, 600.0, 599.5, 339.5, 680, 1200], [1662.8, 1662.8, 959.5, 539.5, 1080, 1920]], dtype=np.float64)
        P = np.stack([t2n(graphics_utils.getProjectionMatrix2(0.01, 100.0, *row[:4], int(row[4]), int(row[5]))) for row in intr])
        np.savez(os.path.join(OUT, "g4_proj.npz"), intr=intr, P=P)
        # ---- G5
        s = torch.exp(torch.randn(256, 3, generator=g) * 0.5)
        r = torch.randn(256, 4, generator=g)
        L = general_utils.build_scaling_rotation(s, r)
        cov6 = general_utils.strip_symmetric(L @ L.transpose(1, 2))
        np.savez(os.path.join(OUT, "g5_cov.npz"), s=t2n(s), r=t2n(r), L=t2n(L), cov6=t2n(cov6), R=t2n(general_utils.build_rotation(r)),
                 inv_sig_in=np.linspace(0.05, 0.95, 19, dtype=np.float32),
                 inv_sig=t2n(general_utils.inverse_sigmoid(torch.linspace(0.05, 0.95, 19))))
        # ---- G6 glue capture
        from slam import gaussian_model as ref_gm, renderer as ref_renderer
        Pn = 1000
        base_cfg = {"device": "cpu", "desired_height": 48, "desired_width": 64, "white_background": False,
                    "cam": {"fx": 51.73, "fy": 51.65, "cx": 31.86, "cy": 25.53},
                    "mapping": {"sh_degree": 0}, "pipeline": {"convert_SHs_python": False, "compute_cov3D_python": False,
                                                              "transform_means_python": True, "force_isotropic": False}}
        pc = ref_gm.GaussianModel(base_cfg)
        pc._xyz = torch.randn(Pn, 3, generator=g) * 0.5 + torch.tensor([0.0, 0.0, 2.5])
        pc._features_dc = torch.randn(Pn, 1, 3, generator=g)
        pc._features_rest = torch.zeros(Pn, 0, 3)
        pc._opacity = torch.randn(Pn, 1, generator=g)
        pc._scaling = torch.randn(Pn, 3, generator=g) * 0.3 - 3.0
        pc._rotation = torch.randn(Pn, 4, generator=g)
        pose = torch.tensor([0.98, 0.05, -0.03, 0.02, 0.1, -0.05, 0.2])
        glue = dict(xyz=t2n(pc._xyz), f_dc=t2n(pc._features_dc), opacity=t2n(pc._opacity), scaling=t2n(pc._scaling),
                    rotation=t2n(pc._rotation), pose=t2n(pose))
        for tm in (True, False):
            for iso in (True, False):
                cfg = dict(base_cfg)
                cfg["pipeline"] = dict(base_cfg["pipeline"], transform_means_python=tm, force_isotropic=iso)
                calls.clear()
                R = ref_renderer.Renderer(cfg)
                R.render(pc, pose)
                tag = f"tm{int(tm)}_iso{int(iso)}"
                rs, kw1 = calls[0]
                _, kw2 = calls[1]
                glue[f"{tag}_view"] = t2n(rs.viewmatrix); glue[f"{tag}_proj"] = t2n(rs.projmatrix)
                glue[f"{tag}_campos"] = t2n(rs.campos); glue[f"{tag}_tanfov"] = np.array([rs.tanfovx, rs.tanfovy])
                glue[f"{tag}_means3D"] = t2n(kw1["means3D"]); glue[f"{tag}_scales"] = t2n(kw1["scales"])
                glue[f"{tag}_rotations"] = t2n(kw1["rotations"]); glue[f"{tag}_opacities"] = t2n(kw1["opacities"])
                glue[f"{tag}_shs"] = t2n(kw1["shs"]); glue[f"{tag}_depthsil"] = t2n(kw2["colors_precomp"])
        np.savez(os.path.join(OUT, "g6_glue.npz"), **glue)
        # ---- G7 seeding
        from slam import mapper as ref_mapper
        H, W = 24, 32
        color = torch.rand(3, H, W, generator=g)
        depth = torch.rand(H, W, generator=g) * 2 + 1
        depth[torch.rand(H, W, generator=g) < 0.1] = 0
        mcfg = {"cam": {"fx": 25.9, "fy": 25.8, "cx": 15.9, "cy": 12.7}}
        fake = types.SimpleNamespace(cfg=mcfg)
        w2c = pose_utils.get_camera_from_tensor(pose)
        mask = (depth > 0).reshape(-1)
        cld, msd = ref_mapper.Mapper.get_pointcloud(fake, color, depth, w2c, mask=mask, compute_mean_sq_dist=True)
        np.savez(os.path.join(OUT, "g7_seed.npz"), color=t2n(color), depth=t2n(depth), pose=t2n(pose), cld=t2n(cld), msd=t2n(msd),
                 log_scale=t2n(torch.log(torch.sqrt(msd))), intr=np.array([25.9, 25.8, 15.9, 12.7]))
        make_g8()
    print("golden fixtures written to", OUT)
    for f in sorted(os.listdir(OUT)):
        if f.endswith(".npz"):
            print(" ", f, os.path.getsize(os.path.join(OUT, f)), "bytes")


if __name__ == "__main__":
    main()
