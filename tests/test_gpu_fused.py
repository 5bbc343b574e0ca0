"""Native fused SLAM iteration (mm3dgs_slam_forward / mm3dgs_loss / mm3dgs_slam_backward / mm3dgs_adam) vs the
torch-graph path built from the same (oracle-checked) generic rasterizer and the reference-mirroring torch losses."""
import random

import numpy as np
import pytest
import torch

from tests import parity_util as pu

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _setup(P=20000, H=240, W=320, iso=False, seed=0):
    from mm3dgs_slam_amd import synthetic as syn
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.gaussian_model import GaussianModel
    from mm3dgs_slam_amd.renderer import Renderer
    cfg = default_config(device=DEV, height=H, width=W, pipeline={"force_isotropic": iso})
    c = cfg["cam"]
    color, depth = syn.rgbd_frame(H, W, seed=seed)
    G = syn.seed_gaussians(color, depth, c["fx"], c["fy"], c["cx"], c["cy"], P, seed=seed, isotropic=False)
    g = GaussianModel(cfg)
    g.training_setup()
    gen = torch.Generator().manual_seed(seed)
    # strongly anisotropic splats: otherwise the rotation gradient (the tangential part of d/dq, a small difference of
    # large radial terms) sits at the float32 noise floor of BOTH implementations and cannot be compared
    G["scaling"] = G["scaling"] + torch.tensor([1.2, -0.8, 0.0])
    g.densification_postfix(G["xyz"].to(DEV), G["f_dc"].to(DEV), torch.zeros(P, 0, 3, device=DEV),
                            (torch.randn(P, 1, generator=gen) * 1.5).to(DEV), G["scaling"].to(DEV),
                            (G["rotation"] * (0.5 + torch.rand(P, 1, generator=gen))).to(DEV), G["rgb"].to(DEV))
    pose = torch.tensor([0.995, 0.03, -0.02, 0.04, 0.03, -0.02, 0.05], device=DEV) * 1.3
    pose[4:] /= 1.3
    return cfg, g, Renderer(cfg), pose, color.to(DEV), depth.to(DEV)


def test_fused_forward_and_backward_match_torch_graph():
    from mm3dgs_slam_amd.fused import FusedEngine
    for iso in (False, True):
        cfg, g, R, pose, color, depth = _setup(iso=iso)
        eng = FusedEngine(R)
        si = eng.forward(pose, g, need_grads=True)
        eng.check_capacity()
        p = pose.clone().requires_grad_(True)
        res = R.render(g, p)
        ref = torch.cat([res["render"], res["depth"]], 0)
        assert pu.rel_l2(eng.out, ref) < 1e-5
        assert torch.equal(eng.radii, res["radii"])
        w = torch.randn(6, eng.H, eng.W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        (ref * w).sum().backward()
        eng.dL.copy_(w)
        stats = (torch.zeros_like(g.max_radii2D), torch.zeros_like(g.xyz_gradient_accum), torch.zeros_like(g.denom))
        eng.backward(si, grads=eng.grads, stats=stats, dpose=eng.dpose)
        tol = 5e-4
        assert pu.rel_l2(eng.dpose, p.grad) < tol, (eng.dpose, p.grad)
        for name, param in (("xyz", g._xyz), ("f_dc", g._features_dc), ("opacity", g._opacity), ("scaling", g._scaling),
                            ("rotation", g._rotation)):
            if name == "rotation" and iso:
                # an isotropic covariance does not depend on the rotation: both gradients are pure rounding noise
                assert eng.grads[name].abs().max() < 1e-3 * (eng.grads["scaling"].abs().max() + 1e-6)
                continue
            assert pu.rel_l2(eng.grads[name], param.grad) < (5e-3 if name == "rotation" else tol), (name, iso)
        vis = res["visibility_filter"]
        assert torch.allclose(stats[2][:, 0], vis.float())
        assert torch.allclose(stats[0], torch.where(vis, res["radii"].float(), torch.zeros_like(stats[0])))
        gn = torch.norm(res["viewspace_points"].grad[:, :2], dim=-1) * vis
        assert pu.rel_l2(stats[1][:, 0], gn) < tol
        for prm in (g._xyz, g._features_dc, g._opacity, g._scaling, g._rotation):
            prm.grad = None


@pytest.mark.parametrize("kind", ["track", "track_pearson", "map", "map_estdepth"])
def test_fused_loss_matches_torch_losses(kind):
    from mm3dgs_slam_amd.fused import FusedEngine, _loss_cfg
    from mm3dgs_slam_amd.loss_utils import l1_loss, pearson_loss, ssim
    cfg, g, R, pose, color, depth = _setup(P=15000, H=200, W=272)
    eng = FusedEngine(R)
    eng.forward(pose, g)
    out = eng.out.clone().requires_grad_(True)
    image, d, sil = out[:3], out[3], out[4]
    est = depth * 0.8 + 0.3
    if kind.startswith("track"):
        presence = sil > 0.99
        loss = (image - color).abs()[:, presence].mean()
        lc = _loss_cfg(eng.H, eng.W, 1.0, 0.0, 0.0, 1, 0, 1, 0.99)
        ref = None
        if kind == "track_pearson":
            loss = loss + 0.05 * pearson_loss(d, depth, mask=presence & (depth > 0), invert_estimate=True)
            lc = _loss_cfg(eng.H, eng.W, 1.0, 0.0, 0.05, 1, 3, 1, 0.99)
            ref = depth
    else:
        loss = 0.8 * l1_loss(image, color) + 0.2 * (1.0 - ssim(image, color))
        if kind == "map":
            loss = loss + 0.05 * pearson_loss(d, depth, mask=depth > 0, invert_estimate=False)
            lc, ref = _loss_cfg(eng.H, eng.W, 0.8, 0.2, 0.05, 0, 2, 0, 0.5), depth
        else:
            loss = loss + 0.05 * pearson_loss(d, est, invert_estimate=False)
            lc, ref = _loss_cfg(eng.H, eng.W, 0.8, 0.2, 0.05, 0, 0, 0, 0.5), est
    loss.backward()
    eng.loss_call(lc, color, ref)
    torch.cuda.synchronize()
    assert abs(float(eng.loss[0]) - float(loss)) < 2e-5 * max(1.0, abs(float(loss))), (eng.loss, loss)
    assert pu.rel_l2(eng.dL, out.grad) < 2e-4


def test_fused_adam_matches_torch_adam():
    from mm3dgs_slam_amd import _lib
    from mm3dgs_slam_amd.rasterizer import _stream
    torch.manual_seed(0)
    p1 = torch.randn(1000, 3, device=DEV); p2 = torch.randn(1000, 1, device=DEV)
    q1, q2 = p1.clone().requires_grad_(True), p2.clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [q1], "lr": 1e-3}, {"params": [q2], "lr": 5e-2}], lr=0.0, eps=1e-15)
    m1, v1, m2, v2 = (torch.zeros_like(t) for t in (p1, p1, p2, p2))
    lib = _lib.load()
    for step in range(1, 6):
        g1, g2 = torch.randn_like(p1), torch.randn_like(p2) * 10
        q1.grad, q2.grad = g1.clone(), g2.clone()
        opt.step()
        tab = (_lib.Mm3dgsAdamGroup * 8)()
        for e, (p, gr, m, v, lr) in zip(tab, ((p1, g1, m1, v1, 1e-3), (p2, g2, m2, v2, 5e-2))):
            e.param, e.grad, e.exp_avg, e.exp_avg_sq, e.n, e.lr = p.data_ptr(), gr.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr
        _lib.check(lib.mm3dgs_adam(tab, 2, step, 0.9, 0.999, 1e-15, _stream()))
    torch.cuda.synchronize()
    assert torch.allclose(p1, q1.detach(), atol=1e-6, rtol=1e-5) and torch.allclose(p2, q2.detach(), atol=1e-5, rtol=1e-5)


def test_fused_tracker_and_mapper_follow_the_torch_graph_loops():
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    results = {}
    for native in (False, True):
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        cfg = default_config(device=DEV, height=240, width=320, tracking={"iters": 40}, mapping={"iters": 12})
        seq = SyntheticSequence(cfg, 3, 30000, seed=4)
        slam = SLAM(cfg, seq, native_loops=native)
        random.seed(1)
        for i in range(3):
            slam.step(i)
        results[native] = (torch.stack(slam.estimate_pose_list[:3]).cpu(), slam.gaussians._xyz.detach().cpu(),
                           slam.gaussians._opacity.detach().cpu(), slam.pose_errors())
    a, b = results[False], results[True]
    assert a[1].shape == b[1].shape
    # two float32 pipelines + Adam: same trajectory, not the same bits (the torch graph's MIOpen convolutions are not even
    # run-to-run deterministic)
    assert (a[0] - b[0]).abs().max() < 4e-3, (a[0], b[0])
    # Adam(eps=1e-15) turns the sign of a ~0 gradient into a full-size step, so individual Gaussians may diverge between
    # two float32 implementations; the population must not
    assert pu.rel_l2(b[1], a[1]) < 1e-3 and (a[2] - b[2]).abs().median() < 5e-3 and torch.quantile((a[2] - b[2]).abs().flatten()[:100000], 0.99) < 0.1, ((a[2] - b[2]).abs().median(), torch.quantile((a[2] - b[2]).abs().flatten()[:100000], 0.99))
    assert b[3][1] < 0.01 and b[3][2] < 0.01, b[3]


def test_fused_path_matches_float64_oracle():
    """The fused kernels (pose transform, activations, depth bundle, chain rules) against the float64 CPU oracle driven
    through the torch-graph Renderer: the strongest statement of parity for the SLAM path."""
    import copy
    import mm3dgs_slam_amd.pose_utils as P
    import mm3dgs_slam_amd.renderer as rmod
    from mm3dgs_slam_amd.fused import FusedEngine
    from mm3dgs_slam_amd.renderer import Renderer
    from oracle.raster_ref import RefRasterizer
    cfg, g, R, pose, color, depth = _setup(P=3000, H=120, W=160)
    eng = FusedEngine(R)
    si = eng.forward(pose, g, need_grads=True)
    eng.check_capacity()
    w = torch.randn(6, eng.H, eng.W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
    eng.dL.copy_(w)
    eng.backward(si, grads=eng.grads, dpose=eng.dpose)

    class PC:
        active_sh_degree = 0
        max_sh_degree = 0
    pc = PC()
    keys = ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation")
    leaf = {k: getattr(g, k).detach().double().cpu().requires_grad_(True) for k in keys}
    pc._xyz, pc._scaling, pc._rotation = leaf["_xyz"], leaf["_scaling"], leaf["_rotation"]
    pc.get_xyz, pc.get_opacity, pc.get_scaling = leaf["_xyz"], torch.sigmoid(leaf["_opacity"]), torch.exp(leaf["_scaling"])
    pc.get_rotation, pc.get_features = torch.nn.functional.normalize(leaf["_rotation"]), leaf["_features_dc"]
    ccfg = copy.deepcopy(cfg)
    ccfg["device"] = "cpu"
    Rc = Renderer(ccfg, rasterizer_cls=RefRasterizer)
    Rc.projection_matrix, Rc.background, Rc._eye = Rc.projection_matrix.double(), Rc.background.double(), Rc._eye.double()
    orig = rmod.get_camera_from_tensor

    def cam64(t):     # the float64 twin of get_camera_from_tensor (which casts to float32 like the reference does)
        return torch.cat([torch.cat([P.quad2rotation(t[None, :4])[0], t[4:7, None]], 1),
                          torch.tensor([[0.0, 0, 0, 1]], dtype=t.dtype)], 0)
    rmod.get_camera_from_tensor = cam64
    try:
        p64 = pose.detach().double().cpu().requires_grad_(True)
        r64 = Rc.render(pc, p64)
        ref = torch.cat([r64["render"], r64["depth"]], 0)
        (ref * w.double().cpu()).sum().backward()
    finally:
        rmod.get_camera_from_tensor = orig
    assert pu.rel_l2(eng.out, ref) <= pu.IMG_TOL
    assert pu.rel_l2(eng.dpose, p64.grad) <= 1e-5          # north_star: pose gradients <= 1e-5
    for name, key in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("opacity", "_opacity"), ("scaling", "_scaling"), ("rotation", "_rotation")):
        assert pu.rel_l2(eng.grads[name], leaf[key].grad) <= pu.GRAD_TOL, name


def _window_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # both ranks share the one GPU of the test box
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    cfg = default_config(device=DEV, height=120, width=160, tracking={"iters": 5}, mapping={"iters": 6, "kf_every": 1})
    seq = SyntheticSequence(cfg, 3, 6000, seed=6)
    slam = SLAM(cfg, seq, window=WindowParallel(rank, world))
    assert type(slam.mapper).__name__ == "FusedMapper"
    for i in range(3):
        slam.step(i)
    torch.save({"xyz": slam.gaussians._xyz.detach().cpu(), "op": slam.gaussians._opacity.detach().cpu(),
                "acc": slam.gaussians.xyz_gradient_accum.cpu()}, os.path.join(out, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_fused_mapper_window_parallel_two_ranks(tmp_path):
    """Native mapping loop with the keyframe window sharded over 2 ranks (gloo, both on this GPU): ranks stay identical."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.spawn(_window_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert a["xyz"].shape == b["xyz"].shape and a["xyz"].shape[0] > 0
    assert torch.equal(a["xyz"], b["xyz"]) and torch.equal(a["op"], b["op"]) and torch.equal(a["acc"], b["acc"])


@pytest.mark.parametrize("long_lists", [False, True])
def test_sort_fused_into_the_forward_launch_is_bit_identical(long_lists):
    """MM3DGS_FWD_SHORT_LISTS selects the sort + forward-composite kernel; it must reproduce the separate launches bit for
    bit (same sort order, same lists, same records) -- also when a tile list exceeds its 2048-key LDS tier and takes the
    global-memory path inside the fused kernel."""
    from mm3dgs_slam_amd.fused import FusedEngine
    if long_lists:
        cfg, g, R, pose, color, depth = _setup(P=12000, H=48, W=64, seed=3)
        with torch.no_grad():
            g._scaling += 2.0          # every splat covers most of the small image: thousands of splats per tile
    else:
        cfg, g, R, pose, color, depth = _setup(P=20000, H=120, W=168, seed=3)
    eng = FusedEngine(R)
    outs = []
    for hint in (1 << 30, 100):        # no hint -> separate sort launches; "short lists" -> fused kernel
        eng.max_tile_len = hint
        si = eng.forward(pose, g, need_grads=True)
        torch.cuda.synchronize()
        hdr = eng.img_state[:16].view(torch.int32).cpu()
        eng.dL.copy_(torch.randn(6, eng.H, eng.W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(5)))
        eng.backward(si, grads=eng.grads, dpose=eng.dpose)
        outs.append((eng.out.clone(), eng.radii.clone(), eng.dpose.clone(), {k: v.clone() for k, v in eng.grads.items()}, int(hdr[2])))
    a, b = outs
    if long_lists:
        assert a[4] > 2048, a[4]       # the case really exercises the long-list path
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert torch.equal(a[2], b[2])
    for k in a[3]:
        assert torch.equal(a[3][k], b[3][k]), k


@pytest.mark.parametrize("pearson", [False, True])
def test_tracking_loss_folded_into_the_compositors_matches_the_loss_kernels(pearson, monkeypatch):
    """mm3dgs_slam_track folds an SSIM-free loss into the forward epilogue / backward prologue; the pose trajectory must
    match the three-kernel loss path (same arithmetic per pixel, only the summation order of the tile sums differs)."""
    from mm3dgs_slam_amd import _lib
    from mm3dgs_slam_amd.fused import FusedEngine, _loss_cfg
    cfg, g, R, pose0, color, depth = _setup(P=20000, H=120, W=168, seed=4)
    with torch.no_grad():
        gt = torch.cat([R.render(g, pose0)["render"]], 0).contiguous()
        ref = R.render(g, pose0)["depth"][0].contiguous()
    results = []
    for no_fold in ("1", "0"):
        monkeypatch.setenv("MM3DGS_NO_FOLDED_LOSS", no_fold)
        eng = FusedEngine(R)
        eng.max_tile_len = 100                       # "short lists" hint: the fused sort kernel (required for folding)
        pose = (pose0 + torch.tensor([0.0, 0.004, -0.003, 0.002, 0.01, -0.008, 0.012], device=DEV)).contiguous()
        m, v = torch.zeros(7, device=DEV), torch.zeros(7, device=DEV)
        step = torch.zeros(1, dtype=torch.int32, device=DEV)
        lcfg = _loss_cfg(eng.H, eng.W, 1.0, 0.0, 0.05 if pearson else 0.0, 1, 1 if pearson else 0, 1, 0.99)
        ad = _lib.Mm3dgsPoseAdam()
        ad.pose, ad.m, ad.v, ad.step = pose.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr()
        ad.lr_q, ad.lr_t, ad.beta1, ad.beta2, ad.eps = 0.002, 0.002, 0.9, 0.999, 1e-8
        eng.track_loop(12, pose, g, lcfg, gt, ref if pearson else None, ad)
        torch.cuda.synchronize()
        results.append((pose.clone(), eng.loss.clone(), int(step)))
    (pa, la, sa), (pb, lb, sb) = results
    assert sa == sb == 12
    assert (pa - pb).abs().max() < 2e-5, (pa, pb)
    assert (la - lb).abs().max() < 1e-5 * max(1.0, float(la.abs().max())), (la, lb)


def test_fused_path_with_huge_splats_matches_torch_graph():
    """Same for the native SLAM path: a map whose Gaussians each cover dozens of tiles (wave-cooperative gather of the
    dense gradient records, block rectangles far larger than a tile)."""
    from mm3dgs_slam_amd.fused import FusedEngine
    cfg, g, R, pose, color, depth = _setup(P=300, H=128, W=160, seed=6)
    with torch.no_grad():
        g._scaling += 3.0
    eng = FusedEngine(R)
    si = eng.forward(pose, g, need_grads=True)
    eng.check_capacity()
    p = pose.clone().requires_grad_(True)
    res = R.render(g, p)
    ref = torch.cat([res["render"], res["depth"]], 0)
    assert int((res["radii"] > 40).sum()) > 100       # the case really has huge splats
    assert pu.rel_l2(eng.out, ref) < 1e-5
    w = torch.randn(6, eng.H, eng.W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(2))
    (ref * w).sum().backward()
    eng.dL.copy_(w)
    eng.backward(si, grads=eng.grads, dpose=eng.dpose)
    assert pu.rel_l2(eng.dpose, p.grad) < 5e-4, (eng.dpose, p.grad)
    for name, param in (("xyz", g._xyz), ("f_dc", g._features_dc), ("opacity", g._opacity), ("scaling", g._scaling)):
        assert pu.rel_l2(eng.grads[name], param.grad) < 5e-4, name
