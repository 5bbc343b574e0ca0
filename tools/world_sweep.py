"""Developer diagnostic (GPU box): the native `transform_means_python: false` path against the float64 oracle over a few seeds, next to the
float32 evaluation of the oracle itself (what float32 arithmetic costs on the same scene).  python tools/world_sweep.py [n_seeds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_gpu_fused import native_vs_oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for world in (True, False):
    for slam_like in (False, True):
        for seed in range(n):
            m = native_vs_oracle(seed + (30 if slam_like else 0), direct=True, world=world, floor=True, slam_like=slam_like, P=4000 if slam_like else 3000)
            ks = ("img", "d_pose", "d_xyz", "d_opacity", "d_scaling", "d_rotation", "d_f_dc")
            print(("world " if world else "camera") + (" slam-like" if slam_like else " stress   ") + f" seed {seed}: " +
                  "  ".join(f"{k} {m[k]:.1e} (f32 {m['f32:' + k]:.1e})" for k in ks), flush=True)
