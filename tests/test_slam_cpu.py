"""Host logic of the tracker / mapper / SLAM loop on CPU with the oracle rasterizer INJECTED as a test double."""
import random

import numpy as np
import torch

from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.renderer import Renderer
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
from oracle.raster_ref import RefRasterizer


def _cfg(**kw):
    cfg = default_config(device="cpu", height=32, width=48, tracking={"iters": 12}, mapping={"iters": 6, "kf_every": 2})
    cfg.update(kw)
    return cfg


def test_fused_and_reference_render_modes_agree():
    cfg = _cfg()
    seq = SyntheticSequence(cfg, 1, 500, seed=1, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    from mm3dgs_slam_amd.slam import _FixedMap
    pc = _FixedMap(seq.seed_params, cfg)
    pose = seq.poses[0]
    a = Renderer(cfg, rasterizer_cls=RefRasterizer, mode="fused").render(pc, pose)
    b = Renderer(cfg, rasterizer_cls=RefRasterizer, mode="reference").render(pc, pose)
    assert torch.allclose(a["render"], b["render"], atol=1e-6) and torch.allclose(a["depth"], b["depth"], atol=1e-5)
    assert torch.equal(a["radii"], b["radii"])


def test_slam_loop_tracks_and_maps():
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    cfg = _cfg()
    seq = SyntheticSequence(cfg, 3, 700, seed=2, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer)
    slam.step(0)
    n0 = slam.gaussians.get_xyz.shape[0]
    assert n0 > 0 and len(slam.mapper.keyframes) == 1
    # frame 1: start from frame 0's pose, tracking must move towards the ground truth
    before = float((seq.poses[0][4:] - seq.poses[1][4:]).norm())
    slam.step(1)
    errs = slam.pose_errors()
    # 32x48 pixels and 12 Adam steps is a plumbing check, not an accuracy claim (the GPU suite checks convergence)
    assert len(errs) == 2 and errs[1] < 0.05 and all(torch.isfinite(p).all() for p in slam.estimate_pose_list[:2])
    slam.step(2)
    assert slam.gaussians.optimizer is not None and slam.gaussians.get_xyz.shape[0] > 0


def test_prune_step_is_a_noop_for_adam_like_the_reference():
    """prune replaces parameters before optimizer.step(), so the step on a prune iteration changes nothing."""
    cfg = _cfg()
    seq = SyntheticSequence(cfg, 1, 400, seed=3, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer)
    cfg["mapping"]["iters"] = 1           # iteration 0 prunes (densify_from_iter 0, pruning_interval 50)
    slam.mapper.num_iter = 1
    color, depth, pose = seq[0]
    slam.estimate_pose_list[0] = pose
    slam.mapper.camera_extent = float(depth.max()) / 2
    with torch.no_grad():
        mask, _ = slam.mapper.initialize_new_gaussians(0, pose, color, depth, depth)
    before = slam.gaussians._xyz.detach().clone()
    slam.mapper.optimize_map(0, 1, [-1], mask, pose, color, depth, depth)
    after = slam.gaussians._xyz.detach()
    assert after.shape[0] <= before.shape[0]
    keep = ~((torch.sigmoid(slam.gaussians._opacity.detach()) < 0.005).squeeze(-1))
    assert torch.equal(after, before[: after.shape[0]]) or after.shape[0] < before.shape[0]


def test_dense_covisibility_ratio_equals_the_index_list_formulation():
    """Mapper.covisibility_ratio_dense (masks over the whole image) vs get_depth_pointcloud + is_covisible (the reference's
    nonzero / gather formulation, slam/mapper.py:141-216): identical ratio."""
    import torch
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.gaussian_model import GaussianModel
    from mm3dgs_slam_amd.mapper import Mapper
    from mm3dgs_slam_amd.pose_utils import get_camera_from_tensor
    H, W = 40, 56
    cfg = default_config(device="cpu", height=H, width=W)
    mp = Mapper(cfg, GaussianModel(cfg), renderer=None, estimate_pose_list=[None])
    gen = torch.Generator().manual_seed(3)
    depth = torch.rand(H, W, generator=gen) * 3 + 0.5
    sil = torch.rand(H, W, generator=gen) * 0.2 + 0.85          # ~1/3 of the pixels fail the 0.99 silhouette test
    depth[5:9, 7:20] = 0.0
    kf_pose = torch.tensor([0.99, 0.05, -0.03, 0.02, 0.1, -0.05, 0.2])
    for cur in (torch.tensor([0.98, 0.1, 0.08, -0.05, 0.6, 0.2, -0.3]), kf_pose.clone(), torch.tensor([0.7, 0.0, 0.7, 0.0, 1.5, 0.0, 1.0])):
        d = depth.clone()
        d[~(sil > 0.99)] = 0
        idx = torch.stack(torch.where(d > 0), dim=1)
        pts = mp.get_depth_pointcloud(d, get_camera_from_tensor(kf_pose), idx)
        fx, fy, cx, cy = mp._intr()
        w2c = get_camera_from_tensor(cur)
        p = pts @ w2c[:3, :3].t() + w2c[:3, 3]
        z = p[:, 2] + 1e-5
        u, v = (fx * p[:, 0] + cx * p[:, 2]) / z, (fy * p[:, 1] + cy * p[:, 2]) / z
        want = ((u < W) & (u > 0) & (v < H) & (v > 0) & (z > 0)).sum() / max(p.shape[0], 1)
        got = mp.covisibility_ratio_dense(depth, sil, kf_pose, cur)
        # points that reproject within 1e-3 px of the image border (pixel row / column 0 seen from the same pose) are
        # decided by float rounding in either formulation: allow exactly those
        edge = ((u.abs() < 1e-3) | ((u - W).abs() < 1e-3) | (v.abs() < 1e-3) | ((v - H).abs() < 1e-3)).sum()
        assert abs(float(got) - float(want)) <= (float(edge) + 1e-3) / max(p.shape[0], 1), (got, want, edge)
        for thr in (0.1, 0.5, 0.95):
            if abs(float(want) - thr) > 0.02:
                assert bool(got > thr) == bool(mp.is_covisible(pts, cur, H, W, threshold=thr))


def test_native_mapper_groups_iterations_into_runs_between_pruning_steps(monkeypatch):
    """FusedMapper.optimize_map hands runs of iterations to the C loop (mm3dgs_slam_map) and keeps the pruning iterations
    in Python: check the grouping against the per-iteration rule of slam/mapper.py:887-942 with a recording fake engine
    (control flow only -- the kernels are covered by the GPU suite)."""
    import torch
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.gaussian_model import GaussianModel

    class FakeEngine:
        H, W = 24, 32
        def __init__(self):
            self.calls, self.P = [], 0
            self.grads = {}
        def forward(self, pose, g, need_grads=False):
            self.P = g._xyz.shape[0]
            self.calls.append(("forward",))
            return "si"
        def loss_call(self, *a):
            self.calls.append(("loss",))
        def backward(self, si, grads=None, stats=None, dpose=None, pose_adam=None, map_adam=None):
            self.calls.append(("backward", grads is not None, stats is not None, map_adam is not None))
        def _ensure(self, P, need_grads):
            pass
        def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, **hints):
            self.calls.append(("map_loop", len(views), stats is not None, int(map_adam.step) if map_adam is not None else None, grads is not None))
        pruned_count = 0
        class _Img:
            @staticmethod
            def data_ptr():
                return 0
        img_state = _Img()
        def check_capacity(self):
            self.calls.append(("check",))
            return True

    for (d_from, d_until, interval, iters, method) in ((0, 50, 50, 150, "vigs"), (10, 30, 10, 45, "vigs"), (0, 0, 50, 7, "vigs"), (5, 100, 4, 12, "vigs"),
                                                       (0, 50, 50, 45, "splatam"), (0, 50, 50, 150, "splatam")):
        cfg = default_config(device="cpu", height=24, width=32, method=method,
                             mapping={"iters": iters, "densify_from_iter": d_from, "densify_until_iter": d_until, "pruning_interval": interval})
        g = GaussianModel(cfg); g.training_setup()
        n = 50
        g.densification_postfix(torch.randn(n, 3), torch.randn(n, 1, 3), torch.zeros(n, 0, 3), torch.zeros(n, 1), torch.full((n, 3), -3.0),
                                torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1), torch.rand(n, 3))
        eng = FakeEngine()
        monkeypatch.setattr(fused.FusedEngine, "eligible", staticmethod(lambda cfg, gaussians: True))
        monkeypatch.setattr(fused, "_engine", lambda renderer: eng)
        monkeypatch.setattr(GaussianModel, "_native", lambda self: False)      # (the device surgery kernels: GPU suite)
        mp = fused.FusedMapper(cfg, g, renderer=None, estimate_pose_list=[None])
        mp.camera_extent = 10.0
        pose = torch.tensor([1.0, 0, 0, 0, 0, 0, 0])
        mp.optimize_map(3, iters, [-1], None, pose, torch.rand(3, 24, 32), torch.rand(24, 32), torch.rand(24, 32))
        # expected sequence from the per-iteration rule
        prune = lambda it: it <= d_until and it >= d_from and it % interval == 0
        dens = lambda it: it <= d_until
        if method == "splatam":     # slam/mapper.py:879-884: prunes at iterations 0 and 20, never collects densification statistics
            prune = lambda it: it <= 20 and it % 20 == 0
            dens = lambda it: False
        want, it, step = [], 0, 1
        while it < iters:
            densify = dens(it)
            if prune(it):
                want.append(("map_loop", 1, densify, None, True))     # gradients + statistics only: the Adam step is a no-op
                it += 1
                continue
            m = 1
            while it + m < iters and not prune(it + m) and dens(it + m) == densify:
                m += 1
            want.append(("map_loop", m, densify, step, False))
            step += m
            it += m
        want.append(("check",))
        assert eng.calls == want, (d_from, d_until, interval, iters, eng.calls, want)
        assert sum(c[1] for c in eng.calls if c[0] == "map_loop") == iters


def test_native_mapper_restores_and_reruns_after_a_binning_overflow(monkeypatch):
    """A forward that overflows its binning capacity anywhere in the mapping loop is reported by the sticky header flag at
    the loop's single read-back; FusedMapper must then put the map, the optimiser state, the statistics and the
    keyframe-pick RNG back and run the same loop again (previously it raised after the Adam steps were applied)."""
    import random
    import torch
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.gaussian_model import GaussianModel

    class FakeEngine:
        H, W = 24, 32
        dev = "cpu"
        def __init__(self, fail_first):
            self.fail, self.picks, self.seen_xyz, self.grads = fail_first, [[]], [], {}
        def forward(self, pose, g, need_grads=False):
            self.picks[-1].append(float(pose[4])); return "si"
        def loss_call(self, *a):
            pass
        def backward(self, si, grads=None, stats=None, dpose=None, pose_adam=None, map_adam=None):
            pass
        def _ensure(self, P, need_grads):
            pass
        def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, **hints):
            self.picks[-1] += [float(v[0][4]) for v in views]
            self.seen_xyz.append(g._xyz.detach().clone())
            with torch.no_grad():                     # what the in-kernel Adam would do: the map moves
                g._xyz += 1.0
                g.xyz_gradient_accum += 1.0
        pruned_count = 0
        class _Img:
            @staticmethod
            def data_ptr():
                return 0
        img_state = _Img()
        def check_capacity(self):
            self.picks.append([])
            if self.fail:
                self.fail -= 1
                return False
            return True

    cfg = default_config(device="cpu", height=24, width=32, mapping={"iters": 20, "densify_until_iter": 10, "pruning_interval": 50})
    g = GaussianModel(cfg); g.training_setup()
    n = 40
    g.densification_postfix(torch.randn(n, 3), torch.randn(n, 1, 3), torch.zeros(n, 0, 3), torch.zeros(n, 1), torch.full((n, 3), -3.0),
                            torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1), torch.rand(n, 3))
    xyz0 = g._xyz.detach().clone()
    eng = FakeEngine(fail_first=2)
    monkeypatch.setattr(fused.FusedEngine, "eligible", staticmethod(lambda cfg, gaussians: True))
    monkeypatch.setattr(fused, "_engine", lambda renderer: eng)
    monkeypatch.setattr(GaussianModel, "_native", lambda self: False)
    mp = fused.FusedMapper(cfg, g, renderer=None, estimate_pose_list=[None])
    mp.camera_extent = 10.0
    kfs = []
    for i in range(4):      # four keyframes told apart by their tx
        kfs.append(fused.Mapper.add_keyframe.__globals__["KeyFrame"](i, torch.rand(3, 24, 32), torch.tensor([1.0, 0, 0, 0, float(i + 1), 0, 0]),
                                                                     torch.rand(24, 32), torch.rand(24, 32)))
    mp.keyframes = kfs
    random.seed(3)
    mp.optimize_map(3, 20, [0, 1, 2, 3, -1], None, torch.tensor([1.0, 0, 0, 0, 0, 0, 0]), torch.rand(3, 24, 32), torch.rand(24, 32), torch.rand(24, 32))
    attempts = [p for p in eng.picks if p]
    assert len(attempts) == 3                                   # two overflowing runs + the good one
    assert attempts[0] == attempts[1] == attempts[2]            # same keyframe picks every time (RNG restored)
    assert len(attempts[0]) == 20
    first_of_attempt = [x for i, x in enumerate(eng.seen_xyz) if i % (len(eng.seen_xyz) // 3) == 0]
    assert all(torch.equal(x, xyz0) for x in first_of_attempt)  # every attempt started from the restored map
    assert mp.mapping_iter_count == 20


def test_native_mapper_with_ample_headroom_reads_the_header_only_at_pruning_steps(monkeypatch):
    """Round 3: with ample headroom (FusedEngine.headroom() >= 1.5) the mapper takes no snapshot, reads the capacity header only where a
    pruning step drains the device anyway (check_capacity_begin before the step's read-back, check_capacity_end after it) and skips the
    end-of-loop read-back; an overflow reported there cannot be rolled back (no snapshot): it is counted and warned about."""
    import warnings
    import torch
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.gaussian_model import GaussianModel

    class FakeEngine:
        H, W = 24, 32
        dev = "cpu"
        def __init__(self, overflow_at=None):
            self.calls, self.grads, self.overflow_at, self.ends = [], {}, overflow_at, 0
        def _ensure(self, P, need_grads):
            pass
        def headroom(self):
            return 2.0
        def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, **hints):
            self.calls.append(("map_loop", len(views)))
        def check_capacity_begin(self):
            self.calls.append(("begin",))
            return ("token",)
        def check_capacity_end(self, token):
            assert token == ("token",)
            self.calls.append(("end",))
            self.ends += 1
            return self.ends != self.overflow_at
        def check_capacity(self):
            self.calls.append(("check",))
            return True

    snaps = []
    monkeypatch.setattr(GaussianModel, "snapshot", lambda self: snaps.append(1) or {})
    for overflow_at in (None, 2):
        cfg = default_config(device="cpu", height=24, width=32, mapping={"iters": 12, "densify_from_iter": 0, "densify_until_iter": 100, "pruning_interval": 5})
        g = GaussianModel(cfg); g.training_setup()
        n = 50
        g.densification_postfix(torch.randn(n, 3), torch.randn(n, 1, 3), torch.zeros(n, 0, 3), torch.zeros(n, 1), torch.full((n, 3), -3.0),
                                torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1), torch.rand(n, 3))
        eng = FakeEngine(overflow_at)
        monkeypatch.setattr(fused.FusedEngine, "eligible", staticmethod(lambda cfg, gaussians: True))
        monkeypatch.setattr(fused, "_engine", lambda renderer: eng)
        monkeypatch.setattr(GaussianModel, "_native", lambda self: False)
        mp = fused.FusedMapper(cfg, g, renderer=None, estimate_pose_list=[None])
        mp.camera_extent = 10.0
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            mp.optimize_map(3, 12, [-1], None, torch.tensor([1.0, 0, 0, 0, 0, 0, 0]), torch.rand(3, 24, 32), torch.rand(24, 32), torch.rand(24, 32))
        # iterations 0, 5, 10 prune: [gradient-only call, begin, (prune), end], runs of 4, 4 and 1 iterations between; no ("check",) at the end
        want = [("map_loop", 1), ("begin",), ("end",), ("map_loop", 4), ("map_loop", 1), ("begin",), ("end",), ("map_loop", 4),
                ("map_loop", 1), ("begin",), ("end",), ("map_loop", 1)]
        assert eng.calls == want, eng.calls
        assert not snaps
        if overflow_at is None:
            assert getattr(mp, "unrecovered_overflows", 0) == 0 and not w
        else:
            assert mp.unrecovered_overflows == 1 and any("overflowed" in str(x.message) for x in w)
        assert mp.mapping_iter_count == 12


def test_rel_pose_loss_safe_variant_is_finite_at_the_start_and_literal_elsewhere():
    import torch
    from mm3dgs_slam_amd.loss_utils import rel_pose_loss
    p0 = torch.tensor([0.9, 0.1, -0.2, 0.3, 0.5, -0.4, 1.0])
    c = p0.clone().requires_grad_(True)
    t_l, q_l = rel_pose_loss(c, p0, safe=True)
    (t_l + q_l).backward()
    assert torch.isfinite(c.grad).all() and float(q_l) == 0.0
    c2 = (p0 + torch.tensor([0.02, -0.03, 0.01, 0.04, 0.1, 0.0, -0.1])).requires_grad_(True)
    a = rel_pose_loss(c2, p0, safe=True)
    ga = torch.autograd.grad(a[0] + a[1], c2)[0]
    b = rel_pose_loss(c2, p0, safe=False)
    gb = torch.autograd.grad(b[0] + b[1], c2)[0]
    assert torch.equal(ga, gb) and torch.equal(a[1], b[1])


def test_splatam_window_ranks_keyframes_by_projected_overlap():
    """method == 'splatam': get_covisible_set is the depth-overlap selection of slam/mapper.py:289-372, not the graph walk."""
    import numpy as np
    import torch
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.mapper import KeyFrame, Mapper
    cfg = default_config(device="cpu", height=96, width=128, method="splatam", mapping={"kf_window_size": 4})
    mp = Mapper(cfg, gaussians=None, renderer=None, estimate_pose_list=[None] * 8)
    depth = torch.full((96, 128), 2.0)
    ident = torch.tensor([1.0, 0, 0, 0, 0, 0, 0])
    far = torch.tensor([1.0, 0, 0, 0, 50.0, 0, 0])                 # looks at nothing the current frame sees
    poses = [ident, far, torch.tensor([1.0, 0, 0, 0, 0.05, 0, 0]), far, torch.tensor([1.0, 0, 0, 0, -0.05, 0.02, 0]), ident]
    mp.keyframes = [KeyFrame(10 * i, torch.zeros(3, 96, 128), p, depth) for i, p in enumerate(poses)]
    torch.manual_seed(0); np.random.seed(0)
    sel, times = mp.get_covisible_set(7, ident, torch.zeros(3, 96, 128), depth)
    assert sel[-1] == len(poses) - 1 and times == [10 * k for k in sel]
    assert len(sel) == 3                                           # kf_window_size - 2 overlapping ones + the last keyframe
    assert set(sel[:-1]) <= {0, 2, 4}                              # never the two that see nothing


def test_native_loops_hand_the_reference_loss_of_each_method_to_the_c_loops(monkeypatch):
    """FusedTracker / FusedMapper translate the loss branches of slam/tracker.py:104-155 and slam/mapper.py:836-873 into one
    Mm3dgsLossConfig per loop: the shipped methods (masked mean-L1 [+ Pearson]; (1-l) L1 + l (1-SSIM) + Pearson) and `splatam`
    (masked sums of depth-L1 + 0.5 colour-L1; mean depth-L1 over { gt_depth > 0 } + 0.5 photometric).  Control flow only: a recording
    fake engine stands in for the kernels (which tests/test_gpu_fused.py checks against the torch losses)."""
    import torch
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.gaussian_model import GaussianModel

    class FakeEngine:
        H, W = 24, 32
        dev = "cpu"
        def __init__(self):
            self.track, self.maps, self.grads = [], [], {}
            self.loss, self.out = torch.zeros(4), torch.zeros(6, 24, 32)
        def track_loop(self, n, pose, g, lcfg, gt_color, ref, ad):
            self.track.append((n, lcfg, ref, bool(ad.prior_pose)))
        def map_loop(self, views, g, lcfg, stats, map_adam, grads=None, **hints):
            self.maps.append((len(views), lcfg, [v[2] for v in views], stats is not None))
        def _ensure(self, P, need_grads):
            pass
        def check_capacity(self):
            return True

    fields = ("w_l1", "w_ssim", "w_pearson", "l1_mask", "pearson_mask", "pearson_invert", "sil_thr", "w_depth_l1", "depth_l1_mask", "l1_sum")
    as_dict = lambda c: {k: round(float(getattr(c, k)), 6) for k in fields}
    gt_color, gt_depth = torch.rand(3, 24, 32), torch.rand(24, 32) + 0.5
    for method in ("vigs", "splatam"):
        cfg = default_config(device="cpu", height=24, width=32, method=method, tracking={"iters": 7, "use_depth_estimate_loss": True, "use_imu_loss": True,
                                                                                        "imu_T_weight": 1.0, "imu_q_weight": 0.1},
                             mapping={"iters": 9})
        g = GaussianModel(cfg); g.training_setup()
        n = 30
        g.densification_postfix(torch.randn(n, 3), torch.randn(n, 1, 3), torch.zeros(n, 0, 3), torch.zeros(n, 1), torch.full((n, 3), -3.0),
                                torch.tensor([[1.0, 0, 0, 0]]).repeat(n, 1), torch.rand(n, 3))
        eng = FakeEngine()
        monkeypatch.setattr(fused.FusedEngine, "eligible", staticmethod(lambda cfg, gaussians: True))
        monkeypatch.setattr(fused, "_engine", lambda renderer: eng)
        monkeypatch.setattr(GaussianModel, "_native", lambda self: False)
        tr = fused.FusedTracker(cfg, g, renderer=None, estimate_pose_list=[None])
        q, T = torch.tensor([1.0, 0, 0, 0]), torch.zeros(3)
        tr.optimize_cam(1, 7, None, q, T, gt_color, gt_depth, None)
        mp = fused.FusedMapper(cfg, g, renderer=None, estimate_pose_list=[None])
        mp.camera_extent = 10.0
        mp.optimize_map(3, 9, [-1], None, torch.tensor([1.0, 0, 0, 0, 0, 0, 0]), gt_color, gt_depth, None)
        (n_it, lc, ref, prior), = eng.track
        assert n_it == 7 and ref is not None and torch.equal(ref, gt_depth)
        if method == "splatam":
            assert as_dict(lc) == dict(w_l1=0.5, w_ssim=0.0, w_pearson=0.0, l1_mask=3.0, pearson_mask=0.0, pearson_invert=0.0, sil_thr=0.99,
                                       w_depth_l1=1.0, depth_l1_mask=3.0, l1_sum=1.0)
            assert not prior                                  # the IMU residual is not part of the splatam branch
        else:
            assert as_dict(lc) == dict(w_l1=1.0, w_ssim=0.0, w_pearson=0.05, l1_mask=1.0, pearson_mask=3.0, pearson_invert=1.0, sil_thr=0.99,
                                       w_depth_l1=0.0, depth_l1_mask=0.0, l1_sum=0.0)
            assert prior
        assert sum(m[0] for m in eng.maps) == 9
        for _, lc, refs, with_stats in eng.maps:
            assert all(r is not None and torch.equal(r, gt_depth) for r in refs)
            if method == "splatam":
                assert as_dict(lc) == dict(w_l1=0.4, w_ssim=0.1, w_pearson=0.0, l1_mask=0.0, pearson_mask=0.0, pearson_invert=0.0, sil_thr=0.5,
                                           w_depth_l1=1.0, depth_l1_mask=2.0, l1_sum=0.0)
                assert not with_stats                         # splatam never collects densification statistics
            else:
                assert as_dict(lc) == dict(w_l1=0.8, w_ssim=0.2, w_pearson=0.05, l1_mask=0.0, pearson_mask=2.0, pearson_invert=0.0, sil_thr=0.5,
                                           w_depth_l1=0.0, depth_l1_mask=0.0, l1_sum=0.0)


def test_checkpoints_results_and_resume_in_the_reference_formats(tmp_path):
    """slam/SLAM.py:286-292,488-500 (point_cloud/iteration_<n>/point_cloud.ply for `save_iterations` and the final map), :294-373
    (results.npz) and :90-104 + slam/mapper.py:65-71 (`iteration` in the configuration resumes: map, poses, keyframes, graph)."""
    import os
    import random
    import numpy as np
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.renderer import Renderer
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    from oracle.raster_ref import RefRasterizer
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    cfg = default_config(device="cpu", height=32, width=48, tracking={"iters": 2}, mapping={"iters": 3, "kf_every": 1, "min_covisibility": 2.0},
                         outputdir=str(tmp_path), save_iterations=[1], debug={"get_runtime_stats": True, "create_video": False, "save_keyframes": False})
    seq = SyntheticSequence(cfg, 3, 500, seed=5, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer)
    slam.run()
    for it in (1, 3):
        assert os.path.exists(tmp_path / "point_cloud" / f"iteration_{it}" / "point_cloud.ply")
    res = np.load(tmp_path / "results.npz", allow_pickle=True)
    assert list(res.keys()) == ["pose_est", "pose_gt", "keyframes", "ate_rmse", "psnr_list", "ssim_list", "lpips_list", "avg_tracking_it_time",
                                "avg_mapping_it_time"]
    assert res["pose_est"].shape == (3, 7) and res["pose_gt"].shape == (3, 7) and len(res["psnr_list"]) == 3 and len(res["lpips_list"]) == 0
    assert float(res["avg_tracking_it_time"]) > 0 and float(res["avg_mapping_it_time"]) > 0
    assert [kf["idx"] for kf in res["keyframes"]] == [kf.idx for kf in slam.mapper.keyframes]
    # resume from the final checkpoint
    cfg2 = dict(cfg, iteration=3)
    again = SLAM(cfg2, seq, rasterizer_cls=RefRasterizer)
    g0, g1 = slam.gaussians, again.gaussians
    assert g1._xyz.shape == g0._xyz.shape and g1._xyz.shape[0] > 0
    for name in ("_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"):
        assert torch.allclose(getattr(g1, name).detach(), getattr(g0, name).detach(), atol=1e-6), name
    assert [kf.idx for kf in again.mapper.keyframes] == [kf.idx for kf in slam.mapper.keyframes]
    assert len(slam.mapper.keyframes) >= 2
    # (the reference rebuilds the graph with update_covisibility_graph(k) for every k, slam/mapper.py:65-71, which compares keyframe k
    #  with keyframes[:-1] -- itself included: a resumed graph carries self-loops the live one does not; mirrored, not "fixed")
    graph = lambda m: {k: {j for j in v if j != k} for k, v in m.covisibility_graph.items() if set(v) - {k}}
    assert graph(again.mapper) == graph(slam.mapper) and graph(slam.mapper)
    for a, b in zip(again.estimate_pose_list, slam.estimate_pose_list):
        assert torch.allclose(a, b.detach(), atol=1e-7)
    n_before = g1._xyz.shape[0]
    again.step(0)       # frame 0 of a resumed run does NOT reseed every pixel (slam/mapper.py:409-418: `"iteration" not in cfg`)
    assert again.gaussians._xyz.shape[0] < n_before + 0.5 * 32 * 48


def test_frames_without_sensor_depth_align_the_monocular_estimate_to_the_map_every_frame():
    """`use_gt_depth: false` (what configs/TUM.yml:8 ships): slam/SLAM.py:392-463 hands the raw monocular estimate to the tracker, renders
    the map once at the tracked pose, fits the estimate to it by least squares and gives the mapper the rescaled depth.  SLAM.step does
    the same when the sequence provides an estimate (SyntheticSequence.est: a stand-in for the network, which is out of scope): frame 0
    takes the reference's arbitrary first-frame scale, later frames must land on the MAP's scale -- whatever that is -- to a few percent."""
    import random
    import numpy as np
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.depth_utils import scale_depth_estimate
    from mm3dgs_slam_amd.renderer import Renderer
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    from oracle.raster_ref import RefRasterizer
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    cfg = default_config(device="cpu", height=32, width=48, use_gt_depth=False, tracking={"iters": 3}, mapping={"iters": 4, "kf_every": 1, "min_covisibility": 2.0})
    seq = SyntheticSequence(cfg, 3, 600, seed=5, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer)
    seen = []
    real = slam.mapper.run_frame
    slam.mapper.run_frame = lambda idx, color, depth, est_scaled, *a, **k: (seen.append(est_scaled.clone()), real(idx, color, depth, est_scaled, *a, **k))[1]
    for i in range(3):
        slam.step(i)
    color, depth, _ = seq[0]
    # frame 0: 1 / (est + 0.001) * png_depth_scale / 10 (slam/SLAM.py:428-434) -- here about half the true depth; the map lives at that scale
    assert torch.allclose(seen[0], 1.0 / (seq.est(0) + 0.001) * 500.0)
    assert float(slam.mapper.camera_extent) == float(seen[0].max()) / cfg["scene_radius_depth_ratio"]
    # later frames: the rescaled estimate agrees with the depth the map renders at the tracked pose
    for i in (1, 2):
        d, sil = slam.mapper._render_depth_sil(slam.estimate_pose_list[i])
        m = (sil > 0.99) & (seq[i][1] > 0)
        assert int(m.sum()) > 100
        rel = ((seen[i] - d).abs() / d)[m]
        assert float(rel.median()) < 0.05, float(rel.median())
    # and the map was seeded from the rescaled estimate, not from the sensor depth (half scale: z well below the sensor's 1.5 m minimum)
    assert float(slam.gaussians.get_xyz[:, 2].median()) < 0.8 * float(depth[depth > 0].median())
    # a direct call reproduces what the frame used
    again = scale_depth_estimate(cfg, 2, seq.est(2), seq[2][1], lambda: slam.mapper._render_depth_sil(slam.estimate_pose_list[2]))
    assert torch.isfinite(again).all()


def test_seed_fraction_thins_the_seeding_with_a_fixed_per_frame_subset():
    """`mapping.seed_fraction` (a workload knob of bench.py, not in the reference): the subset of a frame's pixels that may seed a Gaussian
    is a pure function of (frame index, pixel count, fraction, device) -- every rank of a multi-GPU window draws the same one -- and 1.0
    is the reference's one-Gaussian-per-valid-pixel seeding (slam/mapper.py:600-688)."""
    from mm3dgs_slam_amd.mapper import seed_subset
    a, b, c = seed_subset(5000, 3, 0.5, "cpu"), seed_subset(5000, 3, 0.5, "cpu"), seed_subset(5000, 4, 0.5, "cpu")
    assert a.dtype == torch.bool and a.shape == (5000,) and torch.equal(a, b) and not torch.equal(a, c)
    assert abs(float(a.float().mean()) - 0.5) < 0.05
    assert bool(seed_subset(100, 0, 1.0, "cpu").all())          # torch.rand is in [0, 1): nothing is dropped at 1.0
    n = {}
    for frac in (1.0, 0.5):
        cfg = _cfg()
        cfg["mapping"].update(iters=2, seed_fraction=frac)
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        seq = SyntheticSequence(cfg, 1, 500, seed=1, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
        slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer)
        slam.step(0)
        n[frac] = int(slam.gaussians.get_xyz.shape[0])
    assert 0.35 * n[1.0] < n[0.5] < 0.65 * n[1.0], n


def test_native_tracking_loop_host_side_keeps_the_best_candidate(monkeypatch):
    """keep_best_candidate on the native tracking loop (round 5): FusedTracker hands the library an 8-float { loss, pose } buffer
    (Mm3dgsPoseAdam.best) and reads the tracked pose from it; over tests/cpu_engine.py (the documented semantics of mm3dgs_slam_track on CPU)
    the result is the torch-graph Tracker's best candidate, which -- with learning rates that overshoot -- is not its last iterate."""
    import random
    import numpy as np
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.renderer import Renderer
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    from mm3dgs_slam_amd.tracker import Tracker
    from oracle.raster_ref import RefRasterizer
    from tests import cpu_engine
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    cfg = default_config(device="cpu", height=32, width=48, tracking={"iters": 12, "position_lr": 0.05, "rotation_lr": 0.03}, mapping={"iters": 3, "kf_every": 1})      # (steps larger than the offset: the loss goes up and down)
    seq = SyntheticSequence(cfg, 2, 500, seed=5, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    slam = SLAM(cfg, seq, rasterizer_cls=RefRasterizer, render_mode="reference", native_loops=False)
    slam.step(0)
    color, depth, gt_pose = seq[1]
    start = (gt_pose + torch.tensor([0.0, 0.01, -0.008, 0.006, 0.03, -0.02, 0.03])).contiguous()
    patches, registry = cpu_engine.install(fused)
    for name, value in patches.items():
        monkeypatch.setattr(fused.FusedEngine if name == "eligible" else fused, name, value)
    outs = {}
    for name, cls, keep in (("graph", Tracker, True), ("graph_last", Tracker, False), ("native", fused.FusedTracker, True)):
        trk = cls(cfg, slam.gaussians, slam.renderer, [None, None], keep_best_candidate=keep)
        q = start[:4].clone().requires_grad_(True); T = start[4:].clone().requires_grad_(True)
        opt = torch.optim.Adam([{"params": [T], "lr": cfg["tracking"]["position_lr"]}, {"params": [q], "lr": cfg["tracking"]["rotation_lr"]}])
        trk.optimize_cam(1, 12, opt, q, T, color, depth, depth)
        outs[name] = torch.cat([q.detach(), T.detach()])
    assert any(c[0] == "track" for e in registry.values() for c in e.calls)
    # (the same iteration wins in both programs; the two render the bundle in different pass structures, and Adam at these learning rates
    #  carries their last-bit differences forward over 12 steps -- another iteration would be off by a whole step, 3e-2 .. 5e-2)
    assert torch.allclose(outs["graph"], outs["native"], rtol=0, atol=2e-3), (outs["graph"], outs["native"])
    assert (outs["graph"] - outs["graph_last"]).abs().max() > 1e-3
