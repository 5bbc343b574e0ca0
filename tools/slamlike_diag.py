"""Developer diagnostic (GPU box): the SLAM-like parity scenes of tests/test_gpu_fused.py seed by seed, HIP and the float32 oracle against
the float64 oracle (is a miss of the 1e-5 pose-gradient bar the kernel's, or float32's on that scene?).
    python tools/slamlike_diag.py 33 34 30"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fused import native_vs_oracle
for seed in [int(a) for a in sys.argv[1:]] or [33, 34, 30]:
    for iso in (False, True):
        m = native_vs_oracle(seed, direct=True, slam_like=True, iso=iso, floor=True)
        print(f"seed {seed} iso {iso}: " + "  ".join(f"{k} {m[k]:.1e}/{m['f32:' + k]:.1e}" for k in m if not k.startswith("f32:")), flush=True)
