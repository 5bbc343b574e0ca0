"""Developer diagnostic: distribution of the per-sub-tile list lengths at SLAM size (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from mm3dgs_slam_amd import synthetic as syn
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import FusedEngine
from mm3dgs_slam_amd.gaussian_model import GaussianModel
from mm3dgs_slam_amd.renderer import Renderer

H, W, P = 480, 640, int(sys.argv[1]) if len(sys.argv) > 1 else 157000
dev = "cuda"
K = dict(syn.TUM_INTRINSICS)
fx, fy, cx, cy = K["fx"], K["fy"], K["cx"], K["cy"]
color, depth = syn.rgbd_frame(H, W, seed=0)
G = {k: v.to(dev) for k, v in syn.seed_gaussians(color, depth, fx, fy, cx, cy, P, seed=0, isotropic=True).items()}
cfg = default_config(device=dev, height=H, width=W)
gm = GaussianModel(cfg); gm.training_setup()
gm.densification_postfix(G["xyz"], G["f_dc"], torch.zeros(P, 0, 3, device=dev), G["opacity"], G["scaling"], G["rotation"], G["rgb"])
eng = FusedEngine(Renderer(cfg))
pose = torch.tensor([1.0, 0, 0, 0, 0, 0, 0], device=dev)
eng.forward(pose, gm, need_grads=True); eng.check_capacity()
torch.cuda.synchronize()
T = ((W + 15) // 16) * ((H + 15) // 16)
al = lambda n: (n + 255) // 256 * 256
off = 256 + al(T * 4) + al((T + 1) * 4) + al(T * 4)
raw = eng.img_state.view(torch.uint8)[off:off + 64 * T].cpu().numpy().view(np.uint32)
sc = raw.astype(np.int64).reshape(T * 4, 4)          # [sub-tile (= wave)][4x4 block (= 16-lane row)]
it = sc.max(1)                                        # wave iterations = longest of its four block lists
print("block lists", sc.size, "entries", sc.sum(), "mean", sc.mean(), "p50/p90/p99/max", np.percentile(sc, [50, 90, 99]), sc.max())
print("waves", it.size, "iterations", it.sum(), "mean", it.mean(), "max", it.max(), " row occupancy", sc.sum() / (4 * it.sum()))
print("ideal iterations per SIMD", it.sum() / 1024)
