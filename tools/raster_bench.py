"""Developer timing of the rasterizer alone at SLAM size (run on the GPU box)."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mm3dgs_slam_amd import synthetic as syn, rasterizer as R

ap = argparse.ArgumentParser()
ap.add_argument("--P", type=int, default=150000)
ap.add_argument("--H", type=int, default=480)
ap.add_argument("--W", type=int, default=640)
ap.add_argument("--C", type=int, default=3)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--policy", default="async")
ap.add_argument("--aniso", action="store_true")
ap.add_argument("--fused", action="store_true", help="time the fused SLAM engine (mapping-mode backward) instead of the autograd path")
a = ap.parse_args()
dev = "cuda"
K = dict(syn.TUM_INTRINSICS)
sx, sy = a.W / K["W"], a.H / K["H"]
fx, fy, cx, cy = K["fx"] * sx, K["fy"] * sy, K["cx"] * sx, K["cy"] * sy
color, depth = syn.rgbd_frame(a.H, a.W, seed=0)
G = syn.seed_gaussians(color, depth, fx, fy, cx, cy, a.P, seed=0, isotropic=not a.aniso)
G = {k: v.to(dev) for k, v in G.items()}
view, proj, campos, tx, ty = syn.camera_matrices(a.H, a.W, fx, fy, cx, cy)
rs = R.GaussianRasterizationSettings(a.H, a.W, tx, ty, torch.zeros(3, device=dev), 1.0, view.to(dev), proj.to(dev), 0,
                                     campos.to(dev), False, False)
rast = R.GaussianRasterizer(rs)
means = G["xyz"].clone().requires_grad_(True)
opac = torch.sigmoid(G["opacity"]).requires_grad_(True)
scales = torch.exp(G["scaling"]).requires_grad_(True)
rots = G["rotation"].clone().requires_grad_(True)
shs = G["f_dc"].clone().requires_grad_(True)
m2d = torch.zeros_like(means, requires_grad=True)
extra = None
if a.C == 6:
    z = means.detach()[:, 2:3]
    extra = torch.cat([z, torch.ones_like(z), z * z], 1).requires_grad_(True)
R.set_binning_policy(a.policy)
if a.fused:
    from mm3dgs_slam_amd.config import default_config
    from mm3dgs_slam_amd.fused import FusedEngine
    from mm3dgs_slam_amd.gaussian_model import GaussianModel
    from mm3dgs_slam_amd.renderer import Renderer
    cfg = default_config(device=dev, height=a.H, width=a.W)
    gm = GaussianModel(cfg); gm.training_setup()
    gm.densification_postfix(G["xyz"], G["f_dc"], torch.zeros(a.P, 0, 3, device=dev), G["opacity"], G["scaling"], G["rotation"], G["rgb"])
    eng = FusedEngine(Renderer(cfg))
    pose = torch.tensor([1.0, 0, 0, 0, 0, 0, 0], device=dev)
    si = eng.forward(pose, gm, need_grads=True); eng.check_capacity()
    eng.dL.normal_()
    def it():
        s_ = eng.forward(pose, gm, need_grads=True)
        eng.backward(s_, grads=eng.grads)
    for _ in range(5): it()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.iters): it()
    torch.cuda.synchronize()
    print(json.dumps(dict(P=a.P, fused=True, fwdbwd_ms=(time.perf_counter() - t0) * 1e3 / a.iters)))
    sys.exit(0)

def fwd():
    return rast(means3D=means, means2D=m2d, opacities=opac, shs=shs, scales=scales, rotations=rots, extra_channels=extra)

img, radii = fwd()
torch.cuda.synchronize()
hdr = None
print("visible", int((radii > 0).sum()), "img mean", float(img.mean()), R.last_header())
w = torch.randn_like(img)
def timeit(fn, n):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n, (time.perf_counter() - t0) * 1e3 / n
def f_only():
    with torch.no_grad(): fwd()
def f_b():
    i, _ = fwd(); (i * w).sum().backward()
if os.environ.get("MM3DGS_STATS"):
    f_b(); torch.cuda.synchronize(); print("STATS", R.last_header()); sys.exit(0)
tf = timeit(f_only, a.iters)
tfb = timeit(f_b, a.iters)
print(json.dumps(dict(P=a.P, H=a.H, W=a.W, C=a.C, policy=a.policy, fwd_ms=tf[0], fwd_wall_ms=tf[1], fwdbwd_ms=tfb[0], fwdbwd_wall_ms=tfb[1])))
