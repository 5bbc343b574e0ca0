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
