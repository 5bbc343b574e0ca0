"""Developer diagnostic: per-10-frame timing and map size over a long synthetic sequence (run on the GPU box)."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd import _lib, rasterizer
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
_lib.load(); rasterizer.set_binning_policy("async")
torch.manual_seed(0); random.seed(0); np.random.seed(0)
cfg = default_config(device="cuda:0", height=480, width=640, mapping={"seed_fraction": float(os.environ.get("LONGRUN_SEED_FRACTION", "0.51"))})
seq = SyntheticSequence(cfg, n, 150000, seed=0)
slam = SLAM(cfg, seq)
slam.step(0); torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(1, n):
    if i in (11, 101) and os.environ.get('LONGRUN_PROFILE'):
        _lib.profile_read(); _lib.profile_enable(1)
    slam.step(i)
    if i in (20, 110) and os.environ.get('LONGRUN_PROFILE'):
        torch.cuda.synchronize(); _lib.profile_enable(0)
        pr = _lib.profile_read()
        r = slam.renderer._fused_engine.radii.float()
        r = r[r > 0]
        qs = torch.quantile(r[:100000], torch.tensor([0.5, 0.9, 0.99, 0.999], device=r.device)).tolist()
        print(f"   radii px: median/p90/p99/p99.9 = {qs}  max {float(r.max())}  >24px: {int((r > 24).sum())}  >45px: {int((r > 45).sum())}", flush=True)
        print("   kernel us:", {k: round(v[1] / v[0] * 1e3, 1) for k, v in pr.items() if v[0]}, flush=True)
    if i % 10 == 0:
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng = slam.renderer._fused_engine
        print(f"frames {i-9:3d}-{i:3d}: {(t1 - t0) * 100:.1f} ms/frame  P={slam.gaussians.get_xyz.shape[0]}  keyframes={len(slam.mapper.keyframes)}  "
              f"n_cap={eng.n_cap} ratio={eng.ratio:.2f} max_tile_len={eng.max_tile_len} direct={eng.direct} overflows={getattr(eng, 'overflows', 0)}", flush=True)
        t0 = time.perf_counter()

