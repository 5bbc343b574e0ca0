"""Parity over a sweep of random scenes (GPU box): every case of tests/parity_util.make_case for a range of seeds, HIP path through the C ABI
against the float64 oracle.  Prints, per quantity, the median and the maximum error and how many scenes exceed the 1e-5 camera-gradient bar
(float32 decision flips, see tests/test_gpu_parity.py)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import parity_util as pu

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rows = []
for i in range(n):
    seed = 1000 + i
    deg = i % 4
    kw = dict(P=2000, H=80, W=112, seed=seed, sh_degree=deg, posed=True)
    if i % 3 == 1:
        kw["extras"] = 3 if deg == 0 else 0
    m = pu.compare(pu.make_case(**kw))
    m.pop("case", None)
    rows.append(m)
    print(seed, "deg", deg, {k: f"{v:.1e}" for k, v in m.items() if isinstance(v, float)}, flush=True)
keys = [k for k in rows[0] if isinstance(rows[0][k], float)]
keys = sorted(set(k for r in rows for k in r if isinstance(r[k], float)))
print()
print("| quantity | median | max | scenes > 1e-5 |")
print("|---|---|---|---|")
for k in keys:
    v = [r[k] for r in rows if k in r]
    print(f"| {k} | {statistics.median(v):.1e} | {max(v):.1e} | {sum(1 for x in v if x > 1e-5)} / {len(v)} |")
print("radii mismatches:", sum(r.get("radii_mismatch", 0) for r in rows))

# ---- the native SLAM path (fused forward + backward, direct bins) against the oracle through the torch-graph renderer, with the
# oracle's own float32 evaluation of the same scenes beside it (these scenes -- strongly anisotropic splats, random opacities, a
# white-noise gradient image -- are built to stress the chain rules; float32 costs 1e-6 .. 1e-3 on them whatever the implementation)
from tests.test_gpu_fused import native_vs_oracle
rows2 = []
for seed in range(20, 20 + max(n // 3, 4)):
    m = native_vs_oracle(seed, direct=True, floor=True)
    rows2.append(m)
    print("native", seed, {k: f"{v:.1e}" for k, v in m.items()}, flush=True)
print()
print("| native SLAM path | HIP median | HIP max | float32 oracle median | float32 oracle max | scenes where HIP > 2x float32 oracle |")
print("|---|---|---|---|---|---|")
for k in [k for k in rows2[0] if not k.startswith("f32:")]:
    v = [r[k] for r in rows2]; f = [r["f32:" + k] for r in rows2]
    worse = sum(1 for r in rows2 if r[k] > 2.0 * r["f32:" + k])
    print(f"| {k} | {statistics.median(v):.1e} | {max(v):.1e} | {statistics.median(f):.1e} | {max(f):.1e} | {worse} / {len(v)} |")
