#!/usr/bin/env python
"""Headline benchmark: SLAM frames/sec (track + map) on the TUM fr1/desk-shaped synthetic workload
(BASELINE.json configs[1]: 640x480, ~150k Gaussians, full track+map on 1 MI355X).

A "step" is ONE SLAM frame at the reference's iteration budget (configs/TUM.yml:32,44): Tracker.run_frame = 100 x
{fused 6-channel render -> masked-L1 loss -> backward -> pose Adam} followed by Mapper.run_frame = keyframe logic +
150 x {render -> 0.8 L1 + 0.2 (1-SSIM) + 0.05 Pearson(depth) -> backward -> stats/prune -> map Adam}.  Nothing is
skipped inside the timed region.  Inputs (RGB-D frames, the seeded map) are resident in HBM before timing starts.

N > 1 (launched by torch.distributed.run, one rank per GPU, RCCL): the mapping window is sharded -- every mapping
iteration each rank renders a different keyframe and the Gaussian gradients are all-reduced (window_parallel.py);
tracking is a sequential single-view optimisation and is replicated ("replicas only" for that half).  `value` counts
frame-equivalents of non-redundant work: (tracking views once + mapping views of all ranks) / 250 views per frame.

One JSON line on rank 0, with `roofline` (dominant kernel = backward compositor; algorithmic bytes per SURVEY.md 8d with
the measured N, duration from HIP events recorded on the launch stream inside the C-ABI library) and `cpu_baseline` (the
PyTorch-CPU oracle timed on the host cores on a bounded sample).
"""
import argparse
import json
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gaussians", type=int, default=150000)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--track-iters", type=int, default=100)
    ap.add_argument("--map-iters", type=int, default=150)
    ap.add_argument("--seed-fraction", type=float, default=0.0,
                    help="fraction of frame-0 pixels that seed a Gaussian (0 = choose it so the map has ~--gaussians)")
    ap.add_argument("--policy", default="async", choices=["async", "exact"])
    ap.add_argument("--render-mode", default="fused", choices=["fused", "reference"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL; gloo for plumbing tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--phases", action="store_true", help="diagnostic: per-phase wall time of a frame (adds device syncs)")
    ap.add_argument("--profile", type=int, default=2, help="HIP-event timing inside the library during the timed region: "
                    "2 = only the roofline kernel (backward compositor), 1 = every kernel (adds ~16 events/iteration), 0 = off")
    ap.add_argument("--cpu-baseline-gaussians", type=int, default=50000)
    ap.add_argument("--workload", default="slam", choices=["slam", "c3", "c4", "c5"],
                    help="slam = BASELINE.json configs[1] (the headline line); c3 = configs[2]: UT-MM-shaped 640x330 RGB-D + IMU (configs/UTMM.yml "
                         "settings, IMU dead-reckoning for the pose prediction + the IMU relative-pose residual in the tracking loss), full track+map; "
                         "c4 = configs[3]: Replica-room0-shaped 1200x680, one Gaussian per valid frame-0 pixel (~0.78 M; the map of the 8-GPU window run), "
                         "full track+map on the GPUs given (with --gpus N the mapping window is sharded N ways like the headline workload); "
                         "c5 = configs[4]: synthetic 1920x1080, 3 M Gaussians, SH degree 3, rasterizer forward+backward sweep (a step = one render + "
                         "backward of one view)")
    ap.add_argument("--steady-frames", type=int, default=100,
                    help="slam, 1 GPU: after the timed region keep running this many more frames and report them as `steady_state` (0 = skip)")
    ap.add_argument("--full-seed-steps", type=int, default=10,
                    help="slam, 1 GPU: a second run with the reference-faithful seeding (one Gaussian per valid frame-0 pixel, ~292 k), "
                         "this many timed frames, reported as `full_seed` (0 = skip)")
    ap.add_argument("--moving-frames", type=int, default=60,
                    help="slam, 1 GPU: a further run on a hand-held sweep at TUM fr1/desk's pace (up to 1.4 cm / 0.8 deg per frame: a keyframe every 5 frames, "
                         "seeding, a growing map, a filling window), this many timed frames, reported as `moving` (0 = skip)")
    ap.add_argument("--mono-frames", type=int, default=20,
                    help="slam, 1 GPU: a further run WITHOUT sensor depth (use_gt_depth: false, what configs/TUM.yml:8 ships): every frame renders the map once "
                         "more and fits a synthetic monocular estimate to it by least squares (slam/SLAM.py:411-448), reported as `mono_depth` (0 = skip)")
    ap.add_argument("--window-batch", type=int, default=1, help="views per rank and optimiser step in the mapping window (SURVEY 8e)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="1 GPU: initialise the process group anyway (world_size 1) and run the mapping loop through the multi-GPU orchestration "
                         "(gradient-output loops, the flat all-reduce over RCCL, separate Adam launch) -- the code path of --gpus N, measurable on one GPU")
    ap.add_argument("--grow-to", type=int, default=-1,
                    help="c3 / c4: map size at which the timed region starts (BASELINE.json quotes ~300 k / ~1 M).  The reference's seeding gives one Gaussian per "
                         "valid frame-0 pixel (211 k at 640x330, 795 k at 1200x680); the stated sizes are maps GROWN by keyframes, so the run sweeps a wider "
                         "scene (trajectory_desk, amp 1.6) at the full iteration budget, untimed, until the map has this many Gaussians.  -1: the configuration's "
                         "stated size; 0: no growth phase (round 4's lines: the bounded trajectory on the frame-0 map)")
    ap.add_argument("--grow-max-frames", type=int, default=160)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="--gpus N > 1 (or --window-batch / --force-collectives).  weak (the driver's contract): every rank renders --map-iters views per frame, "
                         "each optimiser step sums N views, `value` counts those views as frame-equivalents.  strong: the frame's --map-iters views IN TOTAL are "
                         "split over the ranks -- ceil(map-iters / N) optimiser steps of N views each -- and `value` is plain frames per second")
    ap.add_argument("--optimizer", choices=("auto", "allreduce", "reduce_scatter"), default="auto",
                    help="multi-GPU window: all-reduce + replicated Adam, or reduce-scatter -> Adam on 1 / N of the elements -> all-gather of the parameters "
                         "(auto: the latter from 500 k Gaussians on; window_parallel.py)")
    return ap.parse_args()


def self_spawn(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU ourselves (the contract's torchrun line does the
    same from outside).  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def instrument_phases(slam):
    """--phases: wall time of each host-level phase of a frame with a device sync around it (diagnostic; perturbs `value`)."""
    import collections
    acc = collections.OrderedDict()
    def wrap(obj, name, label):
        fn = getattr(obj, name)
        def timed(*a, **k):
            torch.cuda.synchronize(); t = time.perf_counter()
            r = fn(*a, **k)
            torch.cuda.synchronize(); acc[label] = acc.get(label, 0.0) + time.perf_counter() - t
            return r
        setattr(obj, name, timed)
    wrap(slam.tracker, "optimize_cam", "track.optimize_cam")
    wrap(slam.tracker, "run_frame", "track.run_frame(total)")
    wrap(slam.mapper, "get_covisible_set", "map.get_covisible_set")
    wrap(slam.mapper, "need_new_keyframe", "map.need_new_keyframe")
    wrap(slam.mapper, "initialize_new_gaussians", "map.initialize_new_gaussians")
    wrap(slam.mapper, "add_keyframe", "map.add_keyframe")
    wrap(slam.mapper, "optimize_map", "map.optimize_map")
    wrap(slam.mapper, "run_frame", "map.run_frame(total)")
    return acc


def prewarm(dev, H, W, n_gaussians, frac):
    """Untimed process warm-up: a 6-frame SLAM run at the benchmark's image size and map size but with a handful of
    iterations per frame and a keyframe every second frame, so that every host-side code path of a frame (tracking,
    keyframe test, new-keyframe seeding, densification statistics, pruning, both mapping loops) has run once with the
    tensor shapes of the timed run.  The first use of a torch operator (per kernel variant) in a process loads its code
    object -- tens of milliseconds each on ROCm -- and the first large allocations go to the driver; without this a short
    timed region is dominated by those one-offs at its first keyframe.  Nothing of the run is kept but the allocator's
    cached blocks."""
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    cfg = default_config(device=dev, height=H, width=W, tracking={"iters": 3},
                         mapping={"iters": 8, "kf_every": 2, "min_covisibility": 2.0, "densify_until_iter": 4, "pruning_interval": 2,
                                  "densification_interval": 2, "seed_fraction": frac})
    seq = SyntheticSequence(cfg, 6, n_gaussians, seed=1)
    SLAM(cfg, seq).run(reraise=True)
    torch.cuda.synchronize()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(args):
    """PyTorch-CPU oracle on the host cores, bounded sample of config C1 (BASELINE.json configs[0]: 640x480, 50k
    Gaussians, tracking): whole tracking iterations (fused 6-channel render fwd + masked-L1 + backward through the
    torch-graph Renderer with the oracle rasterizer injected) repeated until ~10 s of CPU work or 5 iterations, then
    scaled to the iterations of a frame."""
    from mm3dgs_slam_amd import synthetic as syn
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.renderer import Renderer
    from mm3dgs_slam_amd.slam import _FixedMap
    from oracle.raster_ref import RefRasterizer
    cores = min(os.cpu_count() or 1, 16)        # the oracle is a Python loop over tiles of small tensor ops: more threads only add overhead
    torch.set_num_threads(cores)
    H, W, P = args.height, args.width, args.cpu_baseline_gaussians
    cfg = default_config(device="cpu", height=H, width=W)
    c = cfg["cam"]
    color, depth = syn.rgbd_frame(H, W, seed=0)
    G = syn.seed_gaussians(color, depth, c["fx"], c["fy"], c["cx"], c["cy"], P, seed=0)
    pc = _FixedMap(G, cfg)
    R = Renderer(cfg, rasterizer_cls=RefRasterizer)
    pose = torch.tensor([1.0, 0, 0, 0, 0, 0, 0], requires_grad=True)
    n, t0 = 0, time.perf_counter()
    while n < 5 and (n == 0 or time.perf_counter() - t0 < 10.0):
        pose.grad = None
        r = R.render(pc, pose)
        sil = r["depth"][1]
        loss = (r["render"] - color).abs()[:, sil > 0.99].mean()
        loss.backward()
        n += 1
    sec = (time.perf_counter() - t0) / n
    per_frame = sec * (args.track_iters + args.map_iters)
    return {"value": 1.0 / per_frame, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n} tracking iteration(s) (fused 6-channel render fwd + masked-L1 + backward) of the PyTorch-CPU oracle at "
                      f"{W}x{H}, {P} Gaussians (BASELINE.json configs[0]) = {sec:.2f} s each, x{args.track_iters + args.map_iters} "
                      f"iterations/frame; the reference itself has no CPU path (CUDA-only rasterizer)",
            "sec_per_iteration": sec}


def pmc_traffic(match, workload="slam"):
    """Fabric bytes per launch (L2 misses + write-backs: FETCH_SIZE / WRITE_SIZE count the requests the XCD L2s send to the Infinity
    Fabric, Infinity-Cache hits included -- MI355X_MICROARCH.md, section HBM) of the kernel whose name contains every string in `match`,
    from the committed rocprofv3 PMC passes OF THIS WORKLOAD (profiles/*_<workload>_pmc_*.csv, produced by tools/profile_round.sh over
    the same bench command: FETCH_SIZE and WRITE_SIZE in separate passes, KB units, FETCH_SIZE doubled as the guide prescribes for
    16-B/lane reads on gfx950).  None when no summary of this workload is committed (round 3 replayed the configs[1] counters into
    the c3 / c4 lines)."""
    import csv
    import glob
    vals, used = {}, []
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{workload}_pmc_{c}.csv")))
        if not files:
            return None, None
        rows = [r for r in csv.DictReader(open(files[-1])) if all(m in r["kernel"] for m in match)]
        if not rows:
            return None, None
        vals[c] = float(rows[0]["mean_counter_value"]) * 1024.0
        used.append(os.path.relpath(files[-1], ROOT))
    return 2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"], "replayed from " + ", ".join(used) + " (separate rocprofv3 --pmc passes; not measured in this run)"


def c5_workload(args, rank, world, dev):
    """BASELINE.json configs[4]: synthetic 1920x1080, 3 M Gaussians, SH degree 3 (M = 16 view-dependent coefficients, sigma 0.1),
    forward + backward throughput of the rasterizer through the C ABI (generic path: SH colour in-kernel, all input gradients).
    A step = one render + backward of one view; N ranks render N different views of the same map (independent: "weak")."""
    import ctypes as C
    from mm3dgs_slam_amd import _lib, synthetic as syn
    from mm3dgs_slam_amd.rasterizer import GaussianRasterizationSettings, _camera, _ptr, _stream
    lib = _lib.load()
    H, W, P, deg = 1080, 1920, args.gaussians if args.gaussians != 150000 else 3000000, 3
    fov = 60.0
    import math
    fx = fy = (W / 2) / math.tan(math.radians(fov) / 2)
    cx, cy = W / 2 - 0.5, H / 2 - 0.5
    color, depth = syn.rgbd_frame(H, W, seed=1)
    G = {k: v.to(dev) for k, v in syn.seed_gaussians(color, depth, fx, fy, cx, cy, P, seed=1, isotropic=False).items()}
    gen = torch.Generator().manual_seed(2)
    shs = torch.cat([G["f_dc"], (torch.randn(P, 15, 3, generator=gen) * 0.1).to(dev)], 1).contiguous()
    view, proj, campos, tx, ty = syn.camera_matrices(H, W, fx, fy, cx, cy, w2c=syn.small_pose(3 + rank, angle=0.03, trans=0.05))
    view, proj, campos = view.to(dev), proj.to(dev), campos.to(dev)
    bg = torch.zeros(3, device=dev)
    rs = GaussianRasterizationSettings(H, W, tx, ty, bg, 1.0, view, proj, deg, campos, False, False)
    cam = _camera(rs, bg, view, proj, campos)
    means, opac = G["xyz"].contiguous(), torch.sigmoid(G["opacity"]).contiguous()
    scales, rots = torch.exp(G["scaling"]).contiguous(), torch.nn.functional.normalize(G["rotation"]).contiguous()
    u8 = dict(dtype=torch.uint8, device=dev)
    out = torch.empty(3, H, W, device=dev); radii = torch.empty(P, dtype=torch.int32, device=dev)
    geom = torch.empty(lib.mm3dgs_geom_bytes(P), **u8); img = torch.empty(lib.mm3dgs_image_bytes(H, W), **u8)
    host_n = torch.empty(4, dtype=torch.int32).pin_memory()
    M, Cn = 16, 3
    fargs = (P, M, Cn, _ptr(means), _ptr(shs), None, _ptr(opac), _ptr(scales), _ptr(rots), None)
    _lib.check(lib.mm3dgs_forward_geom(C.byref(cam), *fargs, _ptr(radii), _ptr(geom), _ptr(img), C.c_void_p(host_n.data_ptr()), _stream()))
    torch.cuda.synchronize()
    N = int(host_n[0]); n_cap = int(N * 1.05) + 65536
    binning = torch.empty(lib.mm3dgs_binning_bytes(n_cap), **u8)
    scratch = torch.empty(lib.mm3dgs_backward_scratch_bytes(P, n_cap), **u8)
    dL = torch.randn(3, H, W, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    d = dict(means=torch.empty(P, 3, device=dev), m2d=torch.empty(P, 3, device=dev), shs=torch.empty(P, M, 3, device=dev),
             opac=torch.empty(P, 1, device=dev), scales=torch.empty(P, 3, device=dev), rots=torch.empty(P, 4, device=dev))

    def step():
        _lib.check(lib.mm3dgs_forward(C.byref(cam), *fargs, _ptr(out), _ptr(radii), _ptr(geom), _ptr(img), _ptr(binning), n_cap, _stream()))
        _lib.check(lib.mm3dgs_backward(C.byref(cam), P, M, Cn, _ptr(means), _ptr(shs), None, _ptr(opac), _ptr(scales), _ptr(rots), None,
                                       _ptr(radii), _ptr(geom), _ptr(img), _ptr(binning), n_cap, _ptr(dL), _ptr(scratch), _ptr(d["means"]),
                                       _ptr(d["m2d"]), _ptr(d["shs"]), None, _ptr(d["opac"]), _ptr(d["scales"]), _ptr(d["rots"]), None, None,
                                       None, None, 0, _stream()))
    Pv = int((radii > 0).sum())
    T = ((W + 15) // 16) * ((H + 15) // 16)
    r = -(-(32 + (T - 1).bit_length()) // 8)
    alg_fwd = P * (44 + 12 * M) + Pv * (36 + 4 * Cn) + 8 * P + 12 * N + 24 * N * r + 8 * N + (8 * N + 8 * T) + N * (28 + 4 * Cn) + H * W * (4 * Cn + 8)
    alg_bwd = N * (28 + 4 * Cn) + H * W * (4 * Cn + 8) + N * (24 + 4 * Cn) + Pv * (96 + 12 * M + 4 * Cn) + Pv * (44 + 12 * M)
    return step, dict(H=H, W=W, P=P, Pv=Pv, N=N, M=M, C=Cn, alg_fwd=alg_fwd, alg_bwd=alg_bwd,
                      workload=f"synthetic {W}x{H}, fov {fov:.0f} deg, {P} Gaussians seeded like the reference's first frame (anisotropic jitter), "
                               f"SH degree 3 (16 coefficients, sigma 0.1), rasterizer forward + backward with all input gradients through the "
                               f"C ABI (generic path), {N} (tile, splat) pairs, {Pv} visible")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(args))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback for the product path)"
    if os.environ.get("MM3DGS_BENCH_SINGLE_DEVICE"):      # plumbing test of the N > 1 path on a 1-GPU box (with --backend gloo)
        local = 0
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    import torch.distributed as dist
    if world > 1 or args.force_collectives:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("NCCL_DEBUG", "WARN")       # (no version banner on stdout: the driver reads ONE JSON line there)
        if world == 1:
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                s_ = socket.socket(); s_.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s_.getsockname()[1]); s_.close()
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev))
        else:
            dist.init_process_group(args.backend)

    from mm3dgs_slam_amd import _lib, rasterizer
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
    from mm3dgs_slam_amd.window_parallel import WindowParallel
    _lib.load()
    rasterizer.set_binning_policy(args.policy)
    torch.manual_seed(0); random.seed(0); np.random.seed(0)

    collective = world > 1 or args.force_collectives

    def barrier():
        torch.cuda.synchronize()
        if collective:
            dist.barrier()
        torch.cuda.synchronize()

    def finish(out, failures=()):
        # the JSON line is the LAST thing on stdout: RCCL writes a version banner through C stdio (buffered on a pipe, flushed at exit --
        # after a line printed from Python), so the group is torn down and C stdio flushed first
        if collective:
            dist.barrier()
            dist.destroy_process_group()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        if rank == 0:
            print(json.dumps(out), flush=True)
        if failures:           # a multi-GPU line whose self-check failed must not look like a measurement
            log("MULTI-GPU SELF-CHECK FAILED: " + "; ".join(failures))
            sys.exit(3)

    def rank_failures(info):
        """What makes an N-rank line invalid: a process group of another size than --gpus, an all-reduce that does not sum over all
        ranks, replicas that drifted apart."""
        bad = []
        if info is None:
            return bad
        if info["ranks_seen"] != args.gpus:
            bad.append(f"process group has {info['ranks_seen']} ranks, --gpus {args.gpus}")
        if abs(info["allreduce_checksum"] - info["allreduce_expected"]) > 1e-9:
            bad.append(f"all-reduce of rank + 1 gave {info['allreduce_checksum']}, expected {info['allreduce_expected']}")
        if info.get("replicas_identical") is False:
            bad.append("the replicas do not hold identical maps")
        return bad

    def verify_ranks(slam_=None):
        """Self-check of the multi-GPU run for the line's reader: how many ranks the process group really has, an all-reduce whose
        result is known in closed form (sum of rank + 1), and -- the property the window design rests on -- that every replica
        holds the same map at the end (min == max over the ranks of a checksum of the Gaussian parameters)."""
        if not collective:
            return None
        t = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        info = {"backend": dist.get_backend(), "ranks_seen": dist.get_world_size(), "allreduce_checksum": float(t.item()),
                "allreduce_expected": world * (world + 1) / 2.0}
        if slam_ is not None:
            g_ = slam_.gaussians
            c = torch.stack([g_._xyz.double().sum(), g_._opacity.double().sum(), g_._scaling.double().sum(),
                             torch.tensor(float(g_._xyz.shape[0]), device=dev, dtype=torch.float64)])
            lo, hi = c.clone(), c.clone()
            dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
            info["replicas_identical"] = bool(torch.equal(lo, hi))
            info["map_checksum"] = [float(v) for v in hi]
        return info

    if args.workload == "c5":
        step, info = c5_workload(args, rank, world, dev)
        for _ in range(max(args.warmup, 1)):
            step()
        _lib.profile_read(); _lib.profile_enable(1)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        elapsed = time.perf_counter() - t0
        _lib.profile_enable(0)
        prof = _lib.profile_read()
        if collective:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        c5_check = verify_ranks()
        ev_us = float(_lib.load().mm3dgs_profile_event_overhead_ms(None)) * 1e3      # the bracketing event pair's own share of an interval
        kus = {k: max(v[1] / v[0] * 1e3 - ev_us, 0.0) for k, v in prof.items() if v[0]}
        gpu_s = sum(kus.values()) * 1e-6
        H, W = info["H"], info["W"]
        out = {"metric": "raster Mpix/s (forward + backward), synthetic 1920x1080, 3M Gaussians, SH3", "value": H * W * args.steps * world / elapsed / 1e6,
               "unit": "Mpix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": info["workload"], "gaussians": info["P"], "image": [H, W],
                          "multi_gpu": {"description": "independent views of the same map, one per rank; no data-path collective", **(c5_check or {})} if collective else "single GPU"},
               "kernel_us": kus, "event_overhead_us": ev_us,
               "roofline": {"bound": "hbm", "kernel": "whole forward + backward pass (sum of its kernels)",
                            "achieved": (info["alg_fwd"] + info["alg_bwd"]) / gpu_s / 1e9 if gpu_s else None, "peak": 8000.0, "unit": "GB/s",
                            "frac": (info["alg_fwd"] + info["alg_bwd"]) / gpu_s / 1e9 / 8000.0 if gpu_s else None, "traffic": None,
                            "algorithmic_bytes_fwd": info["alg_fwd"], "algorithmic_bytes_bwd": info["alg_bwd"],
                            "note": "algorithmic bytes per SURVEY.md 8d (forward incl. the contract's 6-pass global radix sort term) with the measured "
                                    "Pv and N; duration = sum of the HIP-event kernel times of one pass"}}
        out["roofline"]["traffic"], src = pmc_traffic(("",), "c5_pass")
        if out["roofline"]["traffic"]:
            out["roofline"]["traffic_source"] = src
            out["roofline"]["traffic_kind"] = "L2-miss (fabric) bytes of the whole pass: HBM + Infinity-Cache hits"
            out["roofline"]["traffic_over_algorithmic"] = out["roofline"]["traffic"] / (info["alg_fwd"] + info["alg_bwd"])
        finish(out, rank_failures(c5_check))
        return

    # the reference seeds one Gaussian per valid frame-0 pixel (~292k at 640x480); BASELINE.json's configs[1] is quoted
    # at ~150k Gaussians, so the seeding is thinned to hit that count (stated in config.workload)
    frac = args.seed_fraction or min(1.0, args.gaussians / (0.95 * args.height * args.width))
    extras = world == 1 and rank == 0 and not args.force_collectives
    steady = args.steady_frames if extras else 0

    strong = args.scaling == "strong"
    views_total = args.map_iters
    if strong:      # the frame's mapping views in total, split over the ranks (and the window batch): fewer optimiser steps of more views each
        args.map_iters = -(-args.map_iters // max(world * args.window_batch, 1))
    c3 = args.workload == "c3"
    if c3:
        args.height, args.width = 330, 640
        frac = args.seed_fraction or 1.0      # 640x330 has ~200 k valid pixels: the reference's one-Gaussian-per-pixel seeding, not thinned
    c4 = args.workload == "c4"
    if c4:
        args.height, args.width = 680, 1200
        frac = args.seed_fraction or 1.0      # the reference's seeding: ~0.78 M Gaussians from frame 0, growing with every keyframe
        steady = 0
    grow_to = 0
    if c3 or c4:
        grow_to = (300000 if c3 else 1000000) if args.grow_to < 0 else args.grow_to
    grow_frames_max = args.grow_max_frames if grow_to else 0

    def build(frac_, n_frames, n_target, motion="bounded", top=None):
        if c3:
            from mm3dgs_slam_amd.config import utmm_config
            cfg = utmm_config(device=dev, tracking={"iters": args.track_iters, "use_imu_loss": True, "imu_T_weight": 1.0, "imu_q_weight": 0.1},
                              mapping={"iters": args.map_iters, "seed_fraction": frac_})
        else:
            cfg = default_config(device=dev, height=args.height, width=args.width, tracking={"iters": args.track_iters},
                                 mapping={"iters": args.map_iters, "seed_fraction": frac_}, **(top or {}))
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        seq = SyntheticSequence(cfg, n_frames, n_target, seed=0, motion=motion)        # untimed: builds the RGB-D frames on the GPU
        window = (WindowParallel(rank, world, batch=args.window_batch, always_reduce=args.force_collectives, optimizer=args.optimizer)
                  if (world > 1 or args.window_batch > 1 or args.force_collectives) else None)
        if window is not None:
            window.timing = True
        return SLAM(cfg, seq, render_mode=args.render_mode, window=window)

    log("process warm-up (6-frame SLAM run with a few iterations per frame: loads every operator once)")
    prewarm(dev, args.height, args.width, args.gaussians, frac)
    log("building the synthetic RGB-D sequence")
    slam = build(frac, args.warmup + args.steps + 1 + steady + grow_frames_max, args.gaussians, motion="desk_wide" if grow_to else "bounded")
    log("frame 0 (seeding + first mapping, untimed)")
    _lib.profile_read()
    _lib.profile_enable(args.profile)       # (sampled from here on: the launches before the timed region are reported separately)
    slam.step(0)                                                          # untimed: seeds the map from frame 0 (+ first mapping)
    torch.cuda.synchronize()
    log(f"map has {slam.gaussians.get_xyz.shape[0]} Gaussians; warmup")
    torch.manual_seed(0); random.seed(0); np.random.seed(0)              # identical keyframe picks on every rank
    for i in range(1, 1 + args.warmup):
        slam.step(i)
    # c3 / c4: grow the map to the configuration's stated size with keyframes (untimed, full budget), then time the frames that follow
    grown = 0
    P_seeded = int(slam.gaussians.get_xyz.shape[0])
    while grow_to and grown < grow_frames_max and int(slam.gaussians.get_xyz.shape[0]) < grow_to:
        slam.step(1 + args.warmup + grown)
        grown += 1
        if grown % 20 == 0:
            log(f"  growing: frame {grown}, {slam.gaussians.get_xyz.shape[0]} Gaussians, {len(slam.mapper.keyframes)} keyframes")
    if grow_to:
        log(f"map grown from {P_seeded} to {slam.gaussians.get_xyz.shape[0]} Gaussians in {grown} untimed frames ({len(slam.mapper.keyframes)} keyframes)")
    first_timed = 1 + args.warmup + grown

    phases = instrument_phases(slam) if args.phases else None
    prof_before = _lib.profile_read()       # frame 0 + warm-up frames
    _lib.profile_enable(args.profile)
    barrier()
    log("timed region")
    t0 = time.perf_counter()
    for i in range(first_timed, first_timed + args.steps):
        slam.step(i)
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.profile_enable(False)
    prof = _lib.profile_read()
    log(f"timed region done: {elapsed:.2f} s; mapping loops re-run after a binning overflow: {getattr(slam.mapper, 'loop_reruns', 0)}")
    if phases:
        for k, v in phases.items():
            log(f"  phase {k:34s} {v / args.steps * 1e3:8.2f} ms/frame")
    if collective:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    rank_check = verify_ranks(slam)
    if rank_check is not None and slam.mapper.window is not None:
        st = slam.mapper.window.allreduce_stats()
        # the gradient all-reduce sits between an iteration's backward projection and its Adam launch, with nothing to overlap it (the
        # next projection needs the stepped parameters): this is time added to every optimiser step
        rank_check["allreduce_calls"], rank_check["allreduce_ms_per_step"] = st["calls"], st["ms_per_call"]
        rank_check["allreduce_bytes_per_step"] = st["bytes_per_call"]
        w_ = slam.mapper.window
        sharded_ = w_.shard_optimizer(int(slam.gaussians.get_xyz.shape[0]))
        rank_check["optimizer"] = ("reduce-scatter -> Adam on 1 / N of the elements -> all-gather of the parameters" if sharded_ else
                                   "flat all-reduce -> identical Adam on every replica (fused with the next view's projection)")
        rank_check["optimizer_sharded_steps"] = int(w_.sharded_steps)
        rank_check["optimiser_note"] = (f"N > 1 changes the optimiser, not only the speed: every mapping step sums the gradients of {world * args.window_batch} views "
                                        "(the reference takes ONE view per Adam step, slam/mapper.py:803-807); `value` counts those views as frame-equivalents")

    P_now = int(slam.gaussians.get_xyz.shape[0])
    # measured N (tile-splat pairs) of a representative render, for the algorithmic-bytes figure
    with torch.no_grad():
        slam.renderer.render(slam.gaussians, slam.estimate_pose_list[first_timed + args.steps - 1])
    hdr = rasterizer.last_header()
    N = hdr["num_rendered"]
    H, W, C = args.height, args.width, 6 if args.render_mode == "fused" else 3
    views_per_frame = args.track_iters + args.map_iters
    vps = world * args.window_batch
    frame_equiv = (args.track_iters + args.map_iters * vps) / views_per_frame
    if strong:      # a frame is a frame: its views_total mapping views were rendered by all ranks together
        frame_equiv = 1.0
    value = args.steps * frame_equiv / elapsed
    passes = 1 if args.render_mode == "fused" else 2
    renders = args.steps * views_per_frame
    mpix = H * W * passes * renders * frame_equiv / elapsed / 1e6

    out = {
        "metric": "SLAM frames/sec (track+map), " + ("UT-MM-shaped 640x330 RGB-D + IMU" if c3 else ("Replica-room0-shaped 1200x680" if c4 else "TUM fr1/desk-shaped 640x480")), "value": value, "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"UT-MM-shaped synthetic RGB-D + IMU {W}x{H} (configs/UTMM.yml: intrinsics x 1/2, isotropic Gaussians, pose prediction "
                                f"by IMU dead-reckoning over synthetic 100 Hz samples, Pearson depth term and IMU relative-pose residual (weights 1.0 / 0.1) "
                                f"in the tracking loss), {P_now} Gaussians, " if c3 else
                                (f"Replica-room0-shaped synthetic RGB-D {W}x{H} (TUM intrinsics scaled to the image), {P_now} Gaussians, " if c4 else
                                 f"TUM fr1/desk-shaped synthetic RGB-D {W}x{H} (configs/TUM.yml intrinsics), {P_now} Gaussians, ")) +
                               f"full track+map per frame: {args.track_iters} tracking + {args.map_iters} mapping iterations "
                               f"(reference budget), frame-0 seeding thinned to {frac:.2f} of the pixels, " +
                               (f"map grown by keyframes from {P_seeded} (the reference's one Gaussian per valid frame-0 pixel) to the configuration's stated size over {grown} "
                                f"untimed frames of a hand-held sweep (trajectory_desk, amplitudes x 1.6) that continues through the timed frames, " if grow_to else "") +
                               f"render_mode={args.render_mode}, "
                               f"binning={args.policy}; also in this line: `steady_state` = the {steady} frames that follow the timed region of "
                               f"the same run, `full_seed` = a second run seeded like the reference (one Gaussian per valid frame-0 pixel)",
                   "gaussians": P_now, "image": [H, W], "iterations_per_frame": views_per_frame, "grown": bool(grow_to),
                   **({"strong_scaling": f"{views_total} mapping views per frame in total = {args.map_iters} optimiser steps of {vps} view(s); value = frames per second"} if strong else {}),
                   "multi_gpu": ({"description": f"mapping window sharded: {world} rank(s) x {args.window_batch} view(s) per optimiser step, one all-reduce of the "
                                                f"Gaussian gradients per step; tracking replicated", **(rank_check or {})}
                                 if (vps > 1 or collective) else "single GPU")},
        "raster_mpix_per_s_fwd_bwd": mpix,
        "render_iterations_per_s": renders * frame_equiv / elapsed,
        "num_rendered_pairs": N,
    }
    # ---- roofline: the three compositing kernels of the timed region, each against the algorithmic bytes of SURVEY.md 8d with the
    # measured N (HIP events recorded on the launch stream inside the library, every 64th launch); `roofline` is the one with the
    # largest total time, the others follow in `roofline_other`
    T_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    r_passes = -(-(32 + max(T_tiles - 1, 1).bit_length()) // 8)
    bwd_comp = N * (28 + 4 * C) + H * W * (4 * C + 8) + N * (24 + 4 * C)            # 8d "Backward": composite re-read + per-pixel + gradient scatter
    loss_grad_pass = H * W * 4 * (9 + 3 + 3 + 3)                                    # folded mapping-loss gradient pass: 9 SSIM maps, rgb, gt, depth / silhouette / reference
    fwd_sort_comp = 24 * N * r_passes + 8 * N + (8 * N + 8 * T_tiles) + N * (28 + 4 * C) + H * W * (4 * C + 8)   # 8d "Forward": sort + ranges + composite read + image write
    fused_track = bool(prof.get("track_fwd_bwd", (0, 0.0))[0])      # the tracking iterations ran sort + forward + backward compositing as one launch
    kernels = {
        "composite_bwd": ("composite_bwd_kernel<6,1> (mapping; the mapping loss's gradient-image pass runs in its prologue; round 6: two-phase reduction, list-major block records)", bwd_comp + loss_grad_pass,
                          "N(28+4C) + HW(4C+8) + N(24+4C) [SURVEY 8d backward composite] + 18 HW 4 [folded loss gradient pass: 9 SSIM maps + rgb + gt + depth/sil/ref]",
                          args.map_iters * vps),
        "composite_bwd_track": ("composite_bwd_kernel<6,2> (tracking; loss folded in; round 6: the pose chain -- one pose row per tile instead of gradient records)", bwd_comp, "N(28+4C) + HW(4C+8) + N(24+4C) [SURVEY 8d backward composite; the N(24+4C) gradient-scatter term is contract bytes the pose chain no longer moves]",
                                args.track_iters),
        "track_fwd_bwd": ("sort_composite_fwd_bwd_track_kernel (tracking: per-tile sort + block lists + forward + backward compositing in one launch; masked-L1 loss folded in; round 6: the pose chain -- one pose row per tile, no gradient records, no backward-projection launch)",
                          fwd_sort_comp + bwd_comp,
                          "24 N r + 8N + 8N + 8T + N(28+4C) + HW(4C+8) [SURVEY 8d forward] + N(28+4C) + HW(4C+8) + N(24+4C) [SURVEY 8d backward composite], r = %d" % r_passes,
                          args.track_iters),
        "composite_fwd": ("sort_composite_fwd_kernel<6> (per-tile sort + block lists + forward compositing in one launch)", fwd_sort_comp,
                          "24 N r + 8N [sort, r = %d radix passes of the contract] + 8N + 8T [ranges] + N(28+4C) + HW(4C+8) [composite] (SURVEY 8d forward)" % r_passes,
                          args.map_iters * vps + (0 if fused_track else args.track_iters)),
    }
    # HIP events bracket a launch with their own queue packets: the interval is ~5 us longer than the kernel (rocprofv3's begin / end
    # timestamps); the library measures that share on empty kernels (2 T(1 launch) - T(2 launches)) and it is subtracted here
    ev_overhead = float(_lib.load().mm3dgs_profile_event_overhead_ms(None)) * 1e-3
    recs = []
    for key, (label, alg_bytes, formula, per_frame) in kernels.items():
        n_k, ms_k = prof.get(key, (0, 0.0))
        if not n_k:
            continue
        raw = ms_k / n_k * 1e-3
        dur = max(raw - ev_overhead, 0.25 * raw)      # the bracketing event pair's own share of the interval, calibrated on empty kernels
        ach = alg_bytes / dur / 1e9
        recs.append({"bound": "hbm", "kernel": label, "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None,
                     "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_formula": formula, "avg_launch_us": dur * 1e6,
                     "avg_event_interval_us": raw * 1e6, "event_overhead_us": ev_overhead * 1e6,
                     "timed_launches": n_k, "launches_per_frame": per_frame, "ms_per_frame": dur * 1e3 * per_frame})
    if recs:
        recs.sort(key=lambda r_: -r_["ms_per_frame"])
        pmc_names = {"composite_bwd_kernel<6,1>": ("composite_bwd_kernel", "<6, 1"), "composite_bwd_kernel<6,2>": ("composite_bwd_kernel", "<6, 2"),
                     "sort_composite_fwd_kernel<6>": ("sort_composite_fwd_kernel",), "sort_composite_fwd_bwd_track_kernel": ("sort_composite_fwd_bwd_track_kernel",)}
        for r_ in recs:
            for pre, match in pmc_names.items():
                if r_["kernel"].startswith(pre):
                    r_["traffic"], r_["traffic_source"] = pmc_traffic(match, args.workload)
                    if r_["traffic"]:
                        # the counters see what leaves the XCDs' L2s for the Infinity Fabric -- HBM *or* the 256 MiB Infinity Cache, and an
                        # iteration's working set (~200 MB) fits the latter: this is L2-miss traffic, NOT bytes over the HBM pins, so the rate
                        # below is reported as a fabric rate and never as a fraction of the HBM roofline.  Above 1x the
                        # algorithmic bytes: re-reads / records; below: bytes of the contract this design never moves (the 6-pass radix sort)
                        r_["traffic_kind"] = "L2-miss (fabric) bytes: HBM + Infinity-Cache hits"
                        r_["fabric_GBps"] = r_["traffic"] / (r_["avg_launch_us"] * 1e-6) / 1e9
                        r_["traffic_over_algorithmic"] = r_["traffic"] / r_["algorithmic_bytes_per_launch"]
            if r_["kernel"].startswith("sort_composite_fwd"):
                # the contract's forward figure prices a 6-pass global radix sort (24 N r + 8 N) this design never runs -- the tile bins
                # are sorted in LDS.  The same fraction on the bytes that exist (the contract's figure minus that term), next to the contract's:
                moved = r_["algorithmic_bytes_per_launch"] - (24 * N * r_passes + 8 * N)
                r_["bytes_without_the_contracts_radix_sort"] = moved
                r_["frac_without_the_contracts_radix_sort"] = moved / (r_["avg_launch_us"] * 1e-6) / 1e9 / 8000.0
            if r_["kernel"].startswith("composite_bwd_kernel<6,1>"):
                # SURVEY.md 8d's secondary ceiling: per-(pixel, Gaussian) evaluations E = 256 * N before any early-out, ~25 flop + 1 exp
                # each, against the dense f32 VALU peak -- the ceiling this kernel actually runs into (profiles/r01_sq_counters.md)
                d_ = r_["avg_launch_us"] * 1e-6
                r_["valu"] = {"evaluations_per_launch": 256 * N, "achieved_gevals_per_s": 256 * N / d_ / 1e9, "flop_per_evaluation": 25,
                              "achieved_tflops": 256 * N * 25 / d_ / 1e12, "peak_tflops": 157.3, "frac": 256 * N * 25 / d_ / 1e12 / 157.3}
        out["roofline"] = recs[0]
        out["roofline_other"] = recs[1:]
        out["kernel_us"] = {k: max(v[1] / v[0] * 1e3 - ev_overhead * 1e6, 0.0) for k, v in prof.items() if v[0]}
        # the same averages over every launch of the run after the process warm-up (frame 0, warm-up frames, timed frames): what
        # rocprofv3 --stats of this command averages over, up to the few launches of the process warm-up
        whole = {k: (prof[k][0] + prof_before.get(k, (0, 0.0))[0], prof[k][1] + prof_before.get(k, (0, 0.0))[1]) for k in prof}
        out["kernel_us_whole_run"] = {k: max(v[1] / v[0] * 1e3 - ev_overhead * 1e6, 0.0) for k, v in whole.items() if v[0]}
    if steady:
        log(f"steady state: {steady} more frames of the same run")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        first = first_timed + args.steps
        for i in range(first, first + steady):
            slam.step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        with torch.no_grad():
            slam.renderer.render(slam.gaussians, slam.estimate_pose_list[first + steady - 1])
        N_end = rasterizer.last_header()["num_rendered"]
        out["steady_state"] = {"frames": steady, "value": steady / el, "unit": "frames/s", "ms_per_frame": el / steady * 1e3,
                               "gaussians_at_end": int(slam.gaussians.get_xyz.shape[0]), "keyframes": len(slam.mapper.keyframes),
                               "num_rendered_pairs_at_end": N_end, "pairs_growth": N_end / max(N, 1),
                               "note": f"frames {first}..{first + steady - 1} of the run above (same map, same budget, keyframe work included).  The map keeps "
                                       f"optimising: with the thinned seeding the splats grow to close the gaps, so the (tile, splat) pairs N -- the unit every "
                                       f"kernel's work is proportional to -- grow by `pairs_growth` while the Gaussian count stays put"}
    if extras and args.full_seed_steps and frac < 1.0 and not c3 and not c4:
        log("full-seed run (one Gaussian per valid frame-0 pixel)")
        del slam
        torch.cuda.empty_cache()
        slam2 = build(1.0, 2 + args.full_seed_steps + 1, int(0.95 * args.height * args.width))
        slam2.step(0)
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        for i in (1, 2):
            slam2.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, 3 + args.full_seed_steps):
            slam2.step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["full_seed"] = {"frames": args.full_seed_steps, "value": args.full_seed_steps / el, "unit": "frames/s", "ms_per_frame": el / args.full_seed_steps * 1e3,
                            "gaussians": int(slam2.gaussians.get_xyz.shape[0]),
                            "note": "reference-faithful frame-0 seeding (slam/mapper.py:437-474: every valid-depth pixel), same iteration budget"}
    if extras and args.moving_frames and not c3 and not c4:
        log("moving-camera run")
        try:
            del slam2
        except NameError:
            pass
        slam = None
        torch.cuda.empty_cache()
        n_mv = 3 + args.moving_frames
        slam3 = build(frac, n_mv, args.gaussians, motion="desk")
        slam3.step(0)
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        for i in (1, 2):
            slam3.step(i)
        torch.cuda.synchronize()
        P0, kf0 = int(slam3.gaussians.get_xyz.shape[0]), len(slam3.mapper.keyframes)
        _lib.profile_read(); _lib.profile_enable(args.profile)      # (the hot kernels of THESE frames, sampled like the headline's: growth vs waste)
        t0 = time.perf_counter()
        for i in range(3, n_mv):
            slam3.step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        _lib.profile_enable(False)
        prof_mv = _lib.profile_read()
        with torch.no_grad():
            slam3.renderer.render(slam3.gaussians, slam3.estimate_pose_list[n_mv - 1])
        N_mv = rasterizer.last_header()["num_rendered"]
        errs = slam3.pose_errors()
        out["moving"] = {"frames": args.moving_frames, "value": args.moving_frames / el, "unit": "frames/s", "ms_per_frame": el / args.moving_frames * 1e3,
                         "gaussians_start": P0, "gaussians_end": int(slam3.gaussians.get_xyz.shape[0]), "keyframes_start": kf0,
                         "keyframes_end": len(slam3.mapper.keyframes), "final_translation_error_cm": errs[-1] * 100.0,
                         "num_rendered_pairs_at_end": N_mv, "pairs_vs_headline": N_mv / max(N, 1),
                         "kernel_us": ({k: max(v[1] / v[0] * 1e3 - ev_overhead * 1e6, 0.0) for k, v in prof_mv.items() if v[0]} if args.profile else None),
                         "rmse_translation_error_cm": float(np.sqrt(np.mean(np.square(errs)))) * 100.0,
                         "note": "same iteration budget on a hand-held sweep at TUM fr1/desk's pace (mm3dgs_slam_amd.slam.trajectory_desk: pan +-10 deg at up to "
                                 "0.8 deg / frame, sideways +-0.25 m at up to 1.4 cm / frame; mean 0.94 cm / 0.61 deg per frame) over a scene 1.8x wider than the first "
                                 "view: mapping.kf_every = 5 spaces the keyframes (slam/mapper.py:141-173), every keyframe seeds the newly seen region, the map and "
                                 "the mapping window keep growing inside the timed frames (round 3's `moving` line ran a gentler trajectory: a keyframe every ~15 frames)"}
        del slam3
    if extras and args.mono_frames and not c3 and not c4:
        log("run without sensor depth (per-frame depth alignment)")
        slam = None
        torch.cuda.empty_cache()
        n_mo = 3 + args.mono_frames
        slam4 = build(frac, n_mo, args.gaussians, top={"use_gt_depth": False})
        slam4.step(0)
        torch.manual_seed(0); random.seed(0); np.random.seed(0)
        for i in (1, 2):
            slam4.step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(3, n_mo):
            slam4.step(i)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        out["mono_depth"] = {"frames": args.mono_frames, "value": args.mono_frames / el, "unit": "frames/s", "ms_per_frame": el / args.mono_frames * 1e3,
                             "gaussians": int(slam4.gaussians.get_xyz.shape[0]), "keyframes": len(slam4.mapper.keyframes),
                             "note": "use_gt_depth: false as configs/TUM.yml:8 ships it: the tracker gets a synthetic monocular estimate (inverse-depth-like, arbitrary "
                                     "scale, 3 % smooth error; the network itself is out of scope), every frame renders the map once more at the tracked pose and fits "
                                     "the estimate to it by least squares (slam/SLAM.py:411-448, depth_utils.scale_depth_estimate), the mapper seeds from and regresses "
                                     "(Pearson term) on the rescaled estimate; the headline line runs with sensor depth, where that render + fit do not exist"}
        del slam4
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        log("cpu baseline (oracle on the host cores)")
        out["cpu_baseline"] = cpu_baseline(args)
        log("done")
    finish(out, rank_failures(rank_check))


if __name__ == "__main__":
    main()
