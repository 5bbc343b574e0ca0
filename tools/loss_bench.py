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
