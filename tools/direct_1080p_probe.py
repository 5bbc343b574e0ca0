"""Probe (GPU box): direct bins on a tile grid of more than 4096 tiles (1920x1080 = 8160), which the library keeps on the packed
bins.  Run under `timeout`: a first attempt at this size did not terminate.
    MM3DGS_DIRECT_MAX_TILES=12288 timeout 120 python tools/direct_1080p_probe.py [P] [stage]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from mm3dgs_slam_amd import synthetic as syn
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import FusedEngine
from mm3dgs_slam_amd.gaussian_model import GaussianModel
from mm3dgs_slam_amd.renderer import Renderer

DEV = "cuda:0"
P = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
cfg = default_config(device=DEV, height=H, width=W)
c = cfg["cam"]
color, depth = syn.rgbd_frame(H, W, seed=1)
G = {k: v.to(DEV) for k, v in syn.seed_gaussians(color, depth, c["fx"], c["fy"], c["cx"], c["cy"], P, seed=1, isotropic=False).items()}
g = GaussianModel(cfg)
g.training_setup()
g.densification_postfix(G["xyz"], G["f_dc"], torch.zeros(P, 0, 3, device=DEV), G["opacity"], G["scaling"], G["rotation"], G["rgb"])
pose = torch.tensor([0.999, 0.01, -0.02, 0.015, 0.02, -0.01, 0.03], device=DEV)
eng = FusedEngine(Renderer(cfg))
w = torch.randn(6, H, W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
outs = []
for k in range(3):
    print("render", k, flush=True)
    si = eng.forward(pose, g, need_grads=True)
    torch.cuda.synchronize()
    print("  forward done; direct =", eng.direct, "n_cap", eng.n_cap, flush=True)
    eng.dL.copy_(w)
    eng.backward(si, grads=eng.grads, dpose=eng.dpose)
    torch.cuda.synchronize()
    print("  backward done", flush=True)
    outs.append((eng.out.clone(), eng.radii.clone(), eng.dpose.clone(), {k_: v.clone() for k_, v in eng.grads.items()}))
    ok = eng.check_capacity()
    print("  capacity ok:", ok, "max_tile_len", eng.max_tile_len, flush=True)
a, b = outs[0], outs[-1]
print("direct == packed:", torch.equal(a[0], b[0]), torch.equal(a[1], b[1]), torch.equal(a[2], b[2]), all(torch.equal(a[3][k], b[3][k]) for k in a[3]))
