"""Developer diagnostic (GPU box): are two runs of the native loops over a G9 variant bit-identical?
    python tools/determinism_check.py ba vigs"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_golden_slam import run_variant
for variant in sys.argv[1:] or ["ba", "vigs"]:
    runs = []
    for rep in range(3):
        slam, G, rows = run_variant(variant)
        g = slam.gaussians
        runs.append((torch.stack([p.detach().clone() for p in slam.estimate_pose_list[:len(rows)]]), g._xyz.detach().clone(), g._opacity.detach().clone(),
                     [r["pose_diff"] for r in rows], [r["P"] for r in rows]))
    for rep in (1, 2):
        same_pose = torch.equal(runs[0][0], runs[rep][0])
        same_map = runs[0][1].shape == runs[rep][1].shape and torch.equal(runs[0][1], runs[rep][1]) and torch.equal(runs[0][2], runs[rep][2])
        print(f"{variant}: run 0 vs run {rep}: poses identical {same_pose}, map identical {same_map}; pose diffs to the reference {['%.1e' % v for v in runs[rep][3]]} P {runs[rep][4]}"
              + ("" if same_pose else f"  max |pose difference| {float((runs[0][0] - runs[rep][0]).abs().max()):.2e}"))
