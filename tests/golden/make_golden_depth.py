"""G11: the per-frame depth alignment of the reference (build container only).

    python tests/golden/make_golden_depth.py

`utils/depth_utils.py::get_scale_shift` (:44-99, the least-squares fit of a monocular inverse-depth estimate to a rendered depth) is
executed as is -- `cv2` and `matplotlib`, imported at its module top and absent here, are stubbed -- on three synthetic cases: a dense
mask, a sparse mask with holes in the rendered depth, and a constant-offset estimate.  Stored: inputs, (scale, shift), and the
`1 / (scale * est + shift)` image `slam/SLAM.py:440-448` derives from them."""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg          # noqa: E402

mg.stub_modules()
for name in ("matplotlib", "matplotlib.pyplot"):
    sys.modules.setdefault(name, types.ModuleType(name))


def main():
    from utils.depth_utils import get_scale_shift
    out = {}
    g = torch.Generator().manual_seed(5)
    H, W = 60, 80
    yy, xx = torch.meshgrid(torch.arange(H).float(), torch.arange(W).float(), indexing="ij")
    for k in range(3):
        depth = 1.5 + 2.0 * torch.rand(H, W, generator=g) * (0.5 + 0.5 * torch.sin(xx / 9.0 + k)) ** 2 + 0.3
        est = (700.0 + 200.0 * k) / (depth * (1.0 + 0.02 * torch.cos(yy / 7.0)) + 0.4 + 0.1 * k) + 0.5 * torch.randn(H, W, generator=g)
        sil = torch.rand(H, W, generator=g)
        if k == 1:
            depth = torch.where(torch.rand(H, W, generator=g) < 0.3, torch.zeros_like(depth) + 1e9, depth)      # far holes: inverse ~ 0 but > 0
            sil = sil * 0.5 + 0.5
        mask = (sil > (0.2 if k != 1 else 0.8)) & (est > 1e-6)
        with mg._CpuMode():
            scale, shift = get_scale_shift(est.clone(), depth.clone(), mask.clone(), method="LS")
        out[f"c{k}_est"], out[f"c{k}_depth"], out[f"c{k}_mask"] = mg.t2n(est), mg.t2n(depth), mask.numpy()
        out[f"c{k}_scale"], out[f"c{k}_shift"] = mg.t2n(scale), mg.t2n(shift)
        out[f"c{k}_scaled"] = mg.t2n(1.0 / (scale * est + shift))
        print(k, float(scale), float(shift), int(mask.sum()))
    path = os.path.join(HERE, "g11_depth_align.npz")
    np.savez_compressed(path, **out)
    print("written", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
