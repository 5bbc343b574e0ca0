"""Prints, per frame, how the NATIVE loops (HIP kernels) follow the G9 reference trajectories (the measurements that
tests/test_gpu_golden_slam.py asserts on).  GPU box:   python tools/g9_native_check.py [--large | --shipped] [variant ...]"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from tests.test_gpu_golden_slam import run_variant

if __name__ == "__main__":
    prefix = "g9"
    if "--large" in sys.argv:
        sys.argv.remove("--large"); prefix = "g9L"
    if "--shipped" in sys.argv:
        sys.argv.remove("--shipped"); prefix = "g9D"
    for variant in (sys.argv[1:] or (["vigs", "imu"] if prefix == "g9D" else ["vigs", "vigs_rotfrozen", "splatam", "ba", "imu", "estdepth", "white_bg", "sh2_python", "no_transform"])):
        try:
            slam, G, rows = run_variant(variant, verbose=True, prefix=prefix)
        except Exception as e:      # keep going: this is a survey
            print(f"  {variant}: FAILED {type(e).__name__}: {e}")
            continue
        after = np.array([random.random(), float(np.random.rand()), float(torch.rand(1))])
        print(f"  {variant}: RNG streams end equal: {bool(np.allclose(after, G['rng_after']))}  overflow re-runs: {getattr(slam.mapper, 'loop_reruns', 0)}")
