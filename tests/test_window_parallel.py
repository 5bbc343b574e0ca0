"""Multi-GPU mapping window on CPU: world_size 2 over gloo with the oracle rasterizer injected.  Both ranks must end
with identical maps; the single-rank path must leave gradients untouched."""
import os
import random
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build(window, native=False, ba=False):
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.renderer import Renderer
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    from oracle.raster_ref import RefRasterizer
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    cfg = default_config(device="cpu", height=32, width=48, tracking={"iters": 2}, mapping={"iters": 5 if ba else 3, "kf_every": 1, "do_BA": ba})
    seq = SyntheticSequence(cfg, 3, 500, seed=5, renderer=Renderer(cfg, rasterizer_cls=RefRasterizer))
    return SLAM(cfg, seq, rasterizer_cls=RefRasterizer, render_mode="reference" if native else "fused", window=window, native_loops=native)


def _install_cpu_engine(setattr_fn=None):
    """The native loops' host side (fused.py) over tests/cpu_engine.py: the C-ABI loop semantics on CPU."""
    from mm3dgs_slam_amd import fused
    from tests import cpu_engine
    patches, registry = cpu_engine.install(fused)
    for name, value in patches.items():
        target = fused.FusedEngine if name == "eligible" else fused
        (setattr_fn or setattr)(target, name, value)
    return registry


def _state(slam):
    g = slam.gaussians
    return {"xyz": g._xyz.detach(), "op": g._opacity.detach(), "scaling": g._scaling.detach(), "f_dc": g._features_dc.detach(),
            "acc": g.xyz_gradient_accum.clone(), "denom": g.denom.clone(), "radii": g.max_radii2D.clone(),
            "poses": torch.stack(slam.estimate_pose_list[:3])}


def _run(slam, ba=False):
    for i in range(3):
        slam.step(i)
    if ba:
        # a window of THREE views (two keyframe views + the current frame): with two views per optimiser step some steps render no
        # current-frame view, i.e. leave the only optimised pose (reference quirk, mapper.py) without a gradient
        color, depth, _ = slam.seq[2]
        slam.mapper.optimize_map(2, 6, [0] * (len(slam.mapper.keyframes) + 1) + [-1] if len(slam.mapper.keyframes) < 2 else [0, 1, -1], None, slam.estimate_pose_list[2], color, depth, None)


def _moments(slam):
    opt = slam.gaussians.optimizer
    out = {}
    for gr in opt.param_groups:
        st = opt.state.get(gr["params"][0], {})
        if "exp_avg" in st:
            out["m_" + gr["name"]], out["v_" + gr["name"]] = st["exp_avg"].clone(), st["exp_avg_sq"].clone()
    return out


def _worker(rank, world, port, out, native=False, ba=False, optimizer="allreduce", tag="r"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2 if world <= 2 else 1)
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    if native:
        _install_cpu_engine()
    slam = _build(WindowParallel(rank, world, optimizer=optimizer), native, ba)
    _run(slam, ba)
    st = _state(slam)
    if native:
        from mm3dgs_slam_amd import fused
        st["view_log"] = list(fused._engine(slam.renderer).view_log)
        st.update(_moments(slam))
        st["sharded_steps"] = torch.tensor(slam.mapper.window.sharded_steps)
    torch.save(st, os.path.join(out, f"{tag}{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_window_matches_across_ranks(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert a["xyz"].shape == b["xyz"].shape and a["xyz"].shape[0] > 0
    assert torch.equal(a["xyz"], b["xyz"]) and torch.equal(a["op"], b["op"]) and torch.equal(a["poses"], b["poses"])
    # ... and what they agree on is the right thing: the single-process window-batch = 2 run (the same two views per optimiser
    # step, gradients summed locally instead of by the all-reduce -- SURVEY.md 8e's parity baseline for a G-rank run)
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.set_num_threads(2)
    one = _build(WindowParallel(0, 1, batch=2))
    for i in range(3):
        one.step(i)
    ref = _state(one)
    for k in ref:
        assert ref[k].shape == a[k].shape, k
        assert torch.allclose(ref[k], a[k], rtol=1e-5, atol=1e-7), (k, (ref[k] - a[k]).abs().max())
    # the batch really changes the optimisation (guards against a test that compares two single-view runs)
    single = _build(WindowParallel(0, 1))
    for i in range(3):
        single.step(i)
    assert not torch.allclose(_state(single)["xyz"], ref["xyz"], rtol=1e-5, atol=1e-7)


def test_window_reduce_single_rank_is_identity():
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    slam = _build(WindowParallel(0, 1))
    slam.step(0)
    g = slam.gaussians
    color, depth, pose = slam.seq[0]
    res = slam.renderer.render(g, pose)
    (res["render"].sum() + res["depth"].sum()).backward()
    before = g._xyz.grad.clone()
    w = slam.mapper.window
    n, c, r = w.reduce(g, w.view_stats(res["viewspace_points"], res["visibility_filter"], res["radii"]))
    assert torch.equal(before, g._xyz.grad)
    assert torch.allclose(c[:, 0], res["visibility_filter"].float()) and n.shape == (g._xyz.shape[0], 1)
    assert torch.equal(r, torch.where(res["visibility_filter"], res["radii"], torch.zeros_like(res["radii"])).float())



def test_native_window_orchestration_two_ranks_equal_window_batch_two_and_the_torch_graph_window(tmp_path, monkeypatch):
    """The N > 1 path of the NATIVE loops (what `bench.py --gpus N` runs: per-view mm3dgs_slam_map with gradient outputs, local
    accumulation, ONE flat all-reduce, mm3dgs_adam) on CPU: fused.py's orchestration over the CPU stand-in engine, two gloo ranks
    == one rank with window-batch 2 == the torch-graph loop with the same window."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert a["xyz"].shape == b["xyz"].shape and a["xyz"].shape[0] > 0
    assert torch.equal(a["xyz"], b["xyz"]) and torch.equal(a["op"], b["op"]) and torch.equal(a["poses"], b["poses"])
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.set_num_threads(2)
    registry = _install_cpu_engine(monkeypatch.setattr)
    one = _build(WindowParallel(0, 1, batch=2), native=True)
    for i in range(3):
        one.step(i)
    assert type(one.mapper).__name__ == "FusedMapper" and any(c[0] == "map" for e in registry.values() for c in e.calls)
    ref = _state(one)
    for k in ref:
        assert ref[k].shape == a[k].shape, k
        assert torch.allclose(ref[k], a[k], rtol=1e-5, atol=1e-7), (k, (ref[k] - a[k]).abs().max())
    monkeypatch.undo()
    graph = _build(WindowParallel(0, 1, batch=2))
    for i in range(3):
        graph.step(i)
    tg = _state(graph)
    for k in tg:
        assert tg[k].shape == ref[k].shape, k
        assert torch.allclose(tg[k], ref[k], rtol=2e-4, atol=2e-6), (k, (tg[k] - ref[k]).abs().max())


def test_native_window_orchestration_eight_ranks_equal_window_batch_eight(tmp_path, monkeypatch):
    """The shape of the driver's 8-GPU run (`bench.py --gpus 8`: one rank per GPU, every rank ONE view per optimiser step, one flat
    all-reduce, identical Adam everywhere) on CPU: eight gloo ranks over the CPU stand-in engine must end bit-identical to each other
    and equal to one rank with window-batch 8 (SURVEY.md 8e's parity baseline of a G-rank run).  With 8 views per step the refillable
    keyframe stack (slam/mapper.py:803-807) is emptied and refilled inside single steps -- the rank slices must still partition it."""
    port = _free_port()
    mp.spawn(_worker, args=(8, port, str(tmp_path), True), nprocs=8, join=True)
    states = [torch.load(tmp_path / f"r{r}.pt") for r in range(8)]
    logs = [s.pop("view_log") for s in states]
    a = states[0]
    assert a["xyz"].shape[0] > 0 and int(a["sharded_steps"]) == 0
    for b in states[1:]:
        for k in a:
            assert torch.equal(a[k], b[k]), k
    # Round 5 -- the same eight ranks with the OPTIMISER sharded (WindowParallel(optimizer="reduce_scatter"): reduce-scatter of the flat
    # gradient, mm3dgs_adam on each rank's 1 / 8 of the elements -- slices that cut through the five parameter groups --, all-gather of
    # the parameters, the moments gathered before every pruning step and at the end of each loop): bit for bit the all-reduce path's
    # parameters, statistics, poses AND Adam moments, on every rank.  (gloo has no reduce-scatter: WindowParallel sums by all-reduce and
    # keeps its slice there -- the same sums; what this holds is the sharded step, the parameter exchange and the moment bookkeeping.)
    mp.spawn(_worker, args=(8, _free_port(), str(tmp_path), True, False, "reduce_scatter", "s"), nprocs=8, join=True)
    for r in range(8):
        s = torch.load(tmp_path / f"s{r}.pt")
        assert s.pop("view_log") == logs[r]
        assert int(s.pop("sharded_steps")) > 0
        for k in a:
            if k != "sharded_steps":
                assert torch.equal(a[k], s[k]), (r, k, (a[k] - s[k]).abs().max())
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.set_num_threads(2)
    registry = _install_cpu_engine(monkeypatch.setattr)
    one = _build(WindowParallel(0, 1, batch=8), native=True)
    for i in range(3):
        one.step(i)
    assert type(one.mapper).__name__ == "FusedMapper" and any(c[0] == "map" for e in registry.values() for c in e.calls)
    ref = _state(one)
    # (eight-way sums: the ring all-reduce and the local accumulation add the same eight gradients in different orders, and Adam(eps=1e-15)
    #  divides by sqrt(v) of near-zero gradients -- measured 1.5e-6 on an opacity logit, against 1e-2 between window-batch 8 and 2)
    for k in ref:
        assert ref[k].shape == a[k].shape, k
        assert torch.allclose(ref[k], a[k], rtol=1e-4, atol=1e-5), (k, (ref[k] - a[k]).abs().max())
    # the ranks' slices partition every step's eight views: step by step, the union of what the eight engines rendered is what the
    # window-batch-8 engine rendered (Adam normalises the gradient scale, so a collapsed slicing would barely show in the numbers above)
    one_log = next(iter(registry.values())).view_log
    steps = [c for c in one_log if len(c) == 1]           # (window mode: one mm3dgs_slam_map call per view)
    n_steps = len(logs[0])
    assert all(len(l) == n_steps for l in logs) and len(steps) == 8 * n_steps
    for s_ in range(n_steps):
        assert sorted(l[s_][0] for l in logs) == sorted(c[0] for c in steps[8 * s_:8 * s_ + 8]), s_
        assert [l[s_][0] for l in logs] == [c[0] for c in steps[8 * s_:8 * s_ + 8]], s_      # rank r takes the r-th of the step's views
    assert any(len(set(l[s_][0] for l in logs)) > 1 for s_ in range(n_steps))


def test_two_rank_window_with_bundle_adjustment_equals_window_batch_two(tmp_path):
    """do_BA with a sharded window (the torch-graph loop: fused.py hands such a configuration to it).  A pose that no rank rendered in
    a step must keep grad = None on every rank -- Adam skips it, moments and step counter untouched -- as the single-rank
    window-batch-2 run (and the reference's zero_grad(set_to_none=True) loop, slam/mapper.py:803-825,944-948) does; summing
    "missing = zero" gradients instead would step the current pose by momentum in the iterations that render two keyframes."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), False, True), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(a["xyz"], b["xyz"]) and torch.equal(a["poses"], b["poses"])
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.set_num_threads(2)
    one = _build(WindowParallel(0, 1, batch=2), ba=True)
    _run(one, True)
    ref = _state(one)
    for k in ref:
        assert ref[k].shape == a[k].shape, k
        assert torch.allclose(ref[k], a[k], rtol=1e-5, atol=1e-7), (k, (ref[k] - a[k]).abs().max())
    # bundle adjustment really moved the poses (guards against comparing two runs without it)
    plain = _build(WindowParallel(0, 1, batch=2), ba=False)
    _run(plain, True)
    assert not torch.allclose(_state(plain)["poses"], ref["poses"], rtol=1e-6, atol=1e-8)


def test_native_two_rank_window_with_bundle_adjustment_equals_window_batch_two_and_the_torch_graph(tmp_path, monkeypatch):
    """Round 4: do_BA with a sharded window on the NATIVE loops (fused.py no longer hands it to the torch-graph loop): every view writes
    its pose gradient out (Mm3dgsMapView.dpose_out_or_null), the step's pose gradients are summed over the ranks, and every replica steps
    exactly the poses some rank rendered (mm3dgs_adam).  Two gloo ranks over the CPU stand-in engine == one rank with window-batch 2 ==
    the torch-graph window loop (slam/mapper.py:742-760,803-825,944-948)."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), True, True), nprocs=2, join=True)
    a, b = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    a.pop("view_log"); b.pop("view_log")
    for k in a:
        assert torch.equal(a[k], b[k]), k
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.set_num_threads(2)
    registry = _install_cpu_engine(monkeypatch.setattr)
    one = _build(WindowParallel(0, 1, batch=2), native=True, ba=True)
    _run(one, True)
    assert type(one.mapper).__name__ == "FusedMapper" and any(c[0] == "map" for e in registry.values() for c in e.calls)
    ref = _state(one)
    for k in ref:
        assert ref[k].shape == a[k].shape, k
        assert torch.allclose(ref[k], a[k], rtol=1e-5, atol=1e-7), (k, (ref[k] - a[k]).abs().max())
    monkeypatch.undo()
    graph = _build(WindowParallel(0, 1, batch=2), ba=True)
    _run(graph, True)
    tg = _state(graph)
    # Against the torch-graph loop: the poses (the thing this path adds) tightly; the map as a population.  The two programs step the
    # pose with gradients that differ in the last bits, their poses differ by ~5e-9, and in the one optimiser step that renders the
    # same keyframe twice a float32 decision of the rasterizer (1/255 / T < 1e-4) flips for a few pixels: ~20 of 1530 opacity logits
    # then differ by up to 0.1 lr (measured: gradients equal to 1e-8 in the steps before, 181 gradient entries off in that step;
    # without bundle adjustment the poses are bit-identical and the same window agrees to 4e-6 everywhere).
    assert torch.allclose(tg["poses"], ref["poses"], rtol=0, atol=1e-6), (tg["poses"] - ref["poses"]).abs().max()
    for k in tg:
        assert tg[k].shape == ref[k].shape, k
        d = (tg[k] - ref[k]).abs()
        off = d > 2e-6 + 2e-4 * ref[k].abs()
        assert float(off.float().mean()) < 0.03 and float(d.max()) < 0.02, (k, float(off.float().mean()), float(d.max()))
    plain = _build(WindowParallel(0, 1, batch=2), ba=False)
    _run(plain, True)
    assert not torch.allclose(_state(plain)["poses"], ref["poses"], rtol=1e-6, atol=1e-8)


def _solo_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    torch.set_num_threads(2)
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    _install_cpu_engine()
    slam = _build(WindowParallel(0, 1, always_reduce=True), native=True)
    for i in range(3):
        slam.step(i)
    torch.save(_state(slam), os.path.join(out, "solo.pt"))
    dist.destroy_process_group()


def test_forced_collectives_on_one_rank_equal_the_plain_single_view_loop(tmp_path, monkeypatch):
    """WindowParallel(always_reduce=True) with world_size 1: the multi-GPU orchestration of the native loops (gradient-output
    mm3dgs_slam_map, the flat all-reduce -- the identity here --, mm3dgs_adam) must reproduce the in-kernel-Adam single-view loop.
    The GPU suite runs the same thing over backend "nccl" (RCCL) on the HIP kernels (tests/test_gpu_rccl.py)."""
    mp.spawn(_solo_worker, args=(1, _free_port(), str(tmp_path)), nprocs=1, join=True)
    a = torch.load(tmp_path / "solo.pt")
    _install_cpu_engine(monkeypatch.setattr)
    plain = _build(None, native=True)
    for i in range(3):
        plain.step(i)
    ref = _state(plain)
    for k in ref:
        assert ref[k].shape == a[k].shape, k
        assert torch.allclose(ref[k], a[k], rtol=1e-5, atol=1e-7), (k, (ref[k] - a[k]).abs().max())


@pytest.mark.parametrize("ba", [False, True])
def test_window_step_with_the_fused_adam_and_projection_call_equals_the_two_call_step(ba, monkeypatch):
    """fused.py's two forms of a sharded optimiser step over the CPU stand-in engine: mm3dgs_slam_adam_project (the step from the reduced
    gradients + the next view's projection; the next step's keyframe picks are drawn one call early and its first view is marked
    MM3DGS_FWD_PROJECTED) against mm3dgs_adam + a self-projecting next call -- the same keyframe picks, the same map, the same poses;
    every projected=True call is matched to the pose buffer the fused call was given (tests/cpu_engine.py asserts it)."""
    from mm3dgs_slam_amd import fused
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    torch.set_num_threads(2)
    registry = _install_cpu_engine(monkeypatch.setattr)
    states, logs, n_fused = [], [], []
    for fuse in (True, False):
        monkeypatch.setattr(fused.FusedMapper, "fuse_adam_project", fuse)
        registry.clear()
        slam = _build(WindowParallel(0, 1, batch=2), native=True, ba=ba)
        _run(slam, ba=ba)
        states.append(_state(slam))
        eng = next(iter(registry.values()))
        logs.append(list(eng.view_log))
        n_fused.append(sum(1 for c in eng.calls if c[0] == "adam_project"))
    assert n_fused[0] > 0 and n_fused[1] == 0, n_fused
    assert logs[0] == logs[1]                      # the same views, in the same order, in every step
    for k in states[0]:
        assert torch.equal(states[0][k], states[1][k]), (k, (states[0][k] - states[1][k]).abs().max())


@pytest.mark.parametrize("ba", [False, True])
def test_sharded_optimiser_step_two_ranks_is_bit_identical_to_the_all_reduce_step(ba, tmp_path):
    """Two gloo ranks, native orchestration over the CPU stand-in engine: WindowParallel(optimizer="reduce_scatter") against
    optimizer="allreduce" -- parameters, statistics, poses and both Adam moments bit for bit, with and without bundle adjustment (whose
    frozen-Gaussian mask is applied to the local gradients before the sum on the sharded path, to the summed gradients on the other)."""
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True, ba, "allreduce", "a"), nprocs=2, join=True)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), True, ba, "reduce_scatter", "s"), nprocs=2, join=True)
    for r in range(2):
        a, s = torch.load(tmp_path / f"a{r}.pt"), torch.load(tmp_path / f"s{r}.pt")
        assert a.pop("view_log") == s.pop("view_log")
        assert int(a.pop("sharded_steps")) == 0 and int(s.pop("sharded_steps")) > 0
        assert any(k.startswith("m_") for k in a)
        for k in a:
            assert torch.equal(a[k], s[k]), (r, k, (a[k] - s[k]).abs().max())


def test_shard_bounds_partition_the_flat_gradient():
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    for world in (1, 2, 3, 8):
        for n in (14 * 1, 14 * 37, 14 * 1530, 14 * 157649):
            spans = [WindowParallel(r, world).shard_bounds(n) for r in range(world)]
            S = spans[0][0]
            assert all(s[0] == S for s in spans) and S % 4 == 0 and world * S >= n and world * S <= n + 4 * world + 3
            covered = 0
            for r, (_, lo, hi) in enumerate(spans):
                assert lo == min(r * S, n) and lo <= hi <= n
                covered += hi - lo
            assert covered == n
    w = WindowParallel(0, 8)
    assert not w.shard_optimizer(157649) and w.shard_optimizer(1_000_000) and not WindowParallel(0, 1).shard_optimizer(1_000_000)
    assert WindowParallel(0, 8, optimizer="allreduce").shard_optimizer(1_000_000) is False
