"""Developer diagnostic (GPU box): where the gradient error of a full-size tile-sampled comparison sits -- spread over all touched Gaussians
(arithmetic) or carried by a few (a decision that differs).   python tools/fullsize_diag.py C4_replica_1M"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests import test_gpu_fullsize as tf
from tests.test_gpu_fused import native_vs_oracle

name = sys.argv[1] if len(sys.argv) > 1 else "C4_replica_1M"
H, W, P, iso, _ = tf.CONFIGS[name]
m = native_vs_oracle(seed=7, direct=True, slam_like=True, iso=iso, floor=True, setup=tf._bench_map(name), n_tiles=tf.N_TILES, raw=True)
print({k: (f"{v:.2e}" if isinstance(v, float) else v) for k, v in m.items() if k not in ("raw", "tiles", "tile_errors")})
print("tile errors (rel-L2, worst pixel):", {t: f"{e:.1e} {px:.1e}" for t, (e, px) in m["tile_errors"].items()})
raw = m["raw"]
for name_ in raw["hip"]:
    h, o, f = raw["hip"][name_].double(), raw["oracle"][name_].double(), raw["f32"][name_].double()
    h = h.reshape(o.shape[0], -1); o = o.reshape(o.shape[0], -1); f = f.reshape(o.shape[0], -1)
    for label, x in (("hip", h), ("f32 oracle", f)):
        e2 = ((x - o) ** 2).sum(1)
        tot = float(e2.sum())
        top = torch.topk(e2, 10)
        live = int((o.abs().sum(1) > 0).sum())
        rel = (e2.sqrt() / (o.norm(dim=1) + 1e-30))[o.abs().sum(1) > 0]
        print(f"{name_:9s} {label:10s}: rel-L2 {((tot ** 0.5) / float(o.norm())):.2e}; top-10 Gaussians carry {float(top.values.sum()) / max(tot, 1e-300) * 100:.1f} % of the squared error "
              f"(ids {top.indices[:5].tolist()}); per-Gaussian relative error median {float(rel.median()):.1e} p90 {float(rel.quantile(0.9)):.1e} p99 {float(rel.quantile(0.99)):.1e} ({live} touched)")
