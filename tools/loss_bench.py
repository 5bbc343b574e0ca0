"""Developer timing of the fused loss kernels per variant (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_fused import _setup, DEV
from mm3dgs_slam_amd.fused import FusedEngine, _loss_cfg
cfg, g, R, pose, color, depth = _setup(P=150000, H=480, W=640)
eng = FusedEngine(R)
eng.forward(pose, g, need_grads=True); eng.check_capacity()
variants = {"track": (_loss_cfg(eng.H, eng.W, 1.0, 0.0, 0.0, 1, 0, 1, 0.99), None),
            "map_l1_ssim": (_loss_cfg(eng.H, eng.W, 0.8, 0.2, 0.0, 0, 0, 0, 0.5), None),
            "map_full": (_loss_cfg(eng.H, eng.W, 0.8, 0.2, 0.05, 0, 2, 0, 0.5), depth)}
for name, (lc, ref) in variants.items():
    for _ in range(5): eng.loss_call(lc, color, ref)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): eng.loss_call(lc, color, ref)
    e1.record(); torch.cuda.synchronize()
    print(name, "us/call", e0.elapsed_time(e1) / 50 * 1e3)

# the mapping loop's form (tile sums from the forward epilogue, 2 launches, 4-plane gradient) vs the standalone form, timed by the
# library's own HIP events around the loss launches of mm3dgs_slam_map
from mm3dgs_slam_amd import _lib
for no_rows in ("1", "0"):
    os.environ["MM3DGS_NO_FORWARD_ROWS"] = no_rows
    eng.max_tile_len = 100
    lc, ref = variants["map_full"]
    views = [(pose.contiguous(), color.contiguous(), ref.contiguous())] * 20
    eng._ensure(int(g._xyz.shape[0]), True)
    eng.map_loop(views[:3], g, lc, None, None, grads=eng.grads)
    torch.cuda.synchronize()
    _lib.profile_read(); _lib.profile_enable(1)
    eng.map_loop(views, g, lc, None, None, grads=eng.grads)
    torch.cuda.synchronize()
    _lib.profile_enable(0)
    prof = _lib.profile_read()
    print("map loop, forward rows" if no_rows == "0" else "map loop, standalone loss", {k: round(v[1] / v[0] * 1e3, 2) for k, v in prof.items() if v[0]})
