"""Loading of the G9 fixtures (tests/golden/make_golden_slam.py): the 64x48 set (`g9_*`: float32 inputs, full final parameters) and the
160x120 set of round 4 (`g9L_*`: inputs stored like a dataset stores them -- 8-bit colour, 16-bit depth at TUM's png_depth_scale 5000,
float16 monocular stand-ins -- and per-column quantiles of the final parameters).  Both come back in one shape: float32 arrays the reference
run consumed, bit for bit."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
QS = [0.02, 0.1, 0.25, 0.5, 0.75, 0.9, 0.98]


def load_frames(prefix="g9"):
    F = np.load(os.path.join(HERE, "golden", f"{prefix}_frames.npz"))
    out = dict(H=int(F["H"]), W=int(F["W"]), gt_poses=F["gt_poses"], imu=F["imu"], tstamps=F["tstamps"])
    if "color_u8" in F:
        out["color"] = (F["color_u8"].astype(np.float32) / np.float32(255.0)).astype(np.float32)
        out["depth"] = (F["depth_u16"].astype(np.float32) / np.float32(5000.0)).astype(np.float32)
        out["est"] = F["est_f16"].astype(np.float32)
        out["est_scaled"] = F["est_scaled_f16"].astype(np.float32)
    else:
        for k in ("color", "depth", "est", "est_scaled"):
            out[k] = F[k]
    return out


def load_variant(prefix, variant):
    return np.load(os.path.join(HERE, "golden", f"{prefix}_{variant}.npz"))


def final_quantiles(G, name, t):
    """(this run's, the reference run's) per-column quantiles QS of a final parameter array."""
    import torch
    cols = t.detach().cpu().reshape(t.shape[0], -1).float()
    got = torch.quantile(cols, torch.tensor(QS), dim=0)
    if "q_" + name in G:
        ref = torch.from_numpy(G["q_" + name]).float()
    else:
        r = torch.from_numpy(G[name])
        ref = torch.quantile(r.reshape(r.shape[0], -1).float(), torch.tensor(QS), dim=0)
    return got, ref
