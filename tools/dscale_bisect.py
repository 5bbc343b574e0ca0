"""Developer diagnostic (GPU box): where the native path's d_scaling excess over the float32 oracle sits on the stress scenes --
spread over the Gaussians (arithmetic) or carried by a few (a decision), and what those few have in common.   python tools/dscale_bisect.py [seed ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_fused import native_vs_oracle, _setup

for seed in [int(a) for a in sys.argv[1:]] or [27, 26, 21]:
    m = native_vs_oracle(seed, direct=True, floor=True, raw=True)
    raw = m["raw"]
    cfg, g, R, pose, color, depth = _setup(P=3000, H=120, W=160, seed=seed)
    print(f"seed {seed}: d_scaling HIP {m['d_scaling']:.2e}  float32 oracle {m['f32:d_scaling']:.2e};  d_xyz {m['d_xyz']:.2e} / {m['f32:d_xyz']:.2e};  d_rotation {m['d_rotation']:.2e} / {m['f32:d_rotation']:.2e}")
    o = raw["oracle"]["scaling"].double()
    for label in ("hip", "f32"):
        x = raw[label]["scaling"].double().reshape(o.shape)
        e2 = ((x - o) ** 2).sum(1)
        tot = float(e2.sum())
        top = torch.topk(e2, 8)
        print(f"  {label}: rel-L2 {(tot ** 0.5) / float(o.norm()):.2e}; the 8 worst Gaussians carry {100 * float(top.values.sum()) / tot:.1f} % of the squared error; per-axis rel-L2 "
              + " ".join(f"{float((x[:, k] - o[:, k]).norm() / o[:, k].norm()):.1e}" for k in range(3)))
        if label == "hip":
            ls = g._scaling.detach().cpu()
            rad = raw["radii"]
            for i in top.indices.tolist()[:8]:
                print(f"      id {i}: share {100 * float(e2[i]) / tot:.1f} %  grad oracle {[f'{v:.2e}' for v in o[i].tolist()]}  hip {[f'{v:.2e}' for v in x[i].tolist()]}  log-scales {[round(v, 2) for v in ls[i].tolist()]}  "
                      f"opacity {float(torch.sigmoid(g._opacity[i])):.3f}  radius {int(rad[i])}")
    # the error as a function of the splat's on-screen size
    x = raw["hip"]["scaling"].double().reshape(o.shape)
    err = (x - o).norm(dim=1)
    rad = raw["radii"].double()
    for lo, hi in ((1, 4), (4, 8), (8, 16), (16, 32), (32, 1000)):
        sel = (rad >= lo) & (rad < hi)
        if int(sel.sum()):
            f = raw["f32"]["scaling"].double().reshape(o.shape)
            print(f"  radius [{lo}, {hi}): {int(sel.sum())} Gaussians, rel-L2 hip {float(err[sel].norm() / o[sel].norm()):.2e}  f32 {float((f - o)[sel].norm() / o[sel].norm()):.2e}")
