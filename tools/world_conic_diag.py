"""Developer diagnostic (GPU box): precision of the native projection (pixel centre, conic, depth-bundle z) against the float64 oracle's
projection stage, world-frame means vs camera-frame means.   python tools/world_conic_diag.py [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_fused import _setup
from mm3dgs_slam_amd.fused import FusedEngine
from mm3dgs_slam_amd.pose_utils import quad2rotation
from oracle.raster_ref import RefSettings, preprocess_ref

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for world in (False, True):
    for seed in range(n):
        cfg, g, R, pose, color, depth = _setup(P=3000, H=120, W=160, seed=seed)
        cfg["pipeline"]["transform_means_python"] = not world
        eng = FusedEngine(R)
        eng.forward(pose, g, need_grads=True); eng.check_capacity()
        torch.cuda.synchronize()
        P = g._xyz.shape[0]
        rec = eng.geom[:P * 48].view(torch.float32).reshape(P, 12).cpu().double()
        radii = eng.radii.cpu()
        # float64 oracle projection with the matrices slam/renderer.py builds
        dt = torch.float64
        p = pose.detach().cpu().to(dt)
        Rm = quad2rotation(p[None, :4].to(dt))[0].to(dt)
        W2C = torch.eye(4, dtype=dt); W2C[:3, :3] = Rm; W2C[:3, 3] = p[4:7]
        Pm = R.projection_matrix.detach().cpu().to(dt)
        xyz = g._xyz.detach().cpu().to(dt)
        if world:
            V = W2C.t(); means = xyz
        else:
            V = torch.eye(4, dtype=dt); means = xyz @ Rm.t() + p[4:7]
        s = RefSettings(eng.H, eng.W, R.tanfovx, R.tanfovy, torch.zeros(3, dtype=dt), 1.0, V, V @ Pm, 0, torch.zeros(3, dtype=dt))
        pre = preprocess_ref(means, None, torch.sigmoid(g._opacity.detach().cpu().to(dt)), None, torch.zeros(P, 3, dtype=dt),
                             torch.exp(g._scaling.detach().cpu().to(dt)), torch.nn.functional.normalize(g._rotation.detach().cpu().to(dt)), None, s)
        vis = (radii > 0) & (pre["radii"] > 0)
        rad_mismatch = int(((radii > 0) != (pre["radii"] > 0)).sum()) + int((radii[vis] != pre["radii"][vis]).sum())
        con = torch.stack([rec[:, 2], rec[:, 3], rec[:, 4]], 1)
        rel = ((con - pre["conic"]).abs() / pre["conic"].abs().max(1, keepdim=True).values)[vis]
        dxy = (rec[:, :2] - pre["xy"]).abs()[vis]
        print(f"{'world ' if world else 'camera'} seed {seed}: visible {int(vis.sum())}  radii mismatches {rad_mismatch}  conic rel err median {float(rel.median()):.1e} "
              f"p99 {float(rel.flatten().kthvalue(int(0.99 * rel.numel())).values):.1e} max {float(rel.max()):.1e}   xy err px median {float(dxy.median()):.1e} max {float(dxy.max()):.1e}", flush=True)
