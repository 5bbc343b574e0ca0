"""Developer experiment (GPU box): errors of the native SLAM path on the stress scenes of tools/parity_sweep.py next to the float32
oracle's own, for the library selected by MM3DGS_LIB (tools/build_variant.sh: exact division / exact exp in the compositors).
    python tools/dscale_probe.py [n_seeds]"""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fused import native_vs_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rows = [native_vs_oracle(seed, direct=True, floor=True) for seed in range(20, 20 + n)]
print("lib:", os.environ.get("MM3DGS_LIB", "(product)"))
for k in [k for k in rows[0] if not k.startswith("f32:")]:
    v = [r[k] for r in rows]; f = [r["f32:" + k] for r in rows]
    print(f"  {k:12s} HIP median {statistics.median(v):.2e} max {max(v):.2e}   float32 oracle median {statistics.median(f):.2e} max {max(f):.2e}")
