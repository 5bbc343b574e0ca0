#!/bin/bash
# Developer probe: time of sort_composite_fwd with phases switched off (MM3DGS_EXP bits 2 = no compositing, 4 = no list emission,
# 8 = no sort).  Forward launches only; results are invalid by construction.  Run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cat > /tmp/probe.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from mm3dgs_slam_amd import synthetic as syn
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import FusedEngine
from mm3dgs_slam_amd.gaussian_model import GaussianModel
from mm3dgs_slam_amd.renderer import Renderer
dev = "cuda"; H, W, P = 480, 640, 157000
K = dict(syn.TUM_INTRINSICS)
color, depth = syn.rgbd_frame(H, W, seed=0)
G = {k: v.to(dev) for k, v in syn.seed_gaussians(color, depth, K["fx"], K["fy"], K["cx"], K["cy"], P, seed=0, isotropic=True).items()}
cfg = default_config(device=dev, height=H, width=W)
gm = GaussianModel(cfg); gm.training_setup()
gm.densification_postfix(G["xyz"], G["f_dc"], torch.zeros(P, 0, 3, device=dev), G["opacity"], G["scaling"], G["rotation"], G["rgb"])
eng = FusedEngine(Renderer(cfg))
pose = torch.tensor([1.0, 0, 0, 0, 0.02, 0.01, 0.03], device=dev)
eng.forward(pose, gm, need_grads=True); eng.check_capacity()
for _ in range(60):
    eng.forward(pose, gm, need_grads=True)
torch.cuda.synchronize()
PY
for E in ${PROBE_EXPS:-0 2 6 10 14}; do
  rm -rf /tmp/p_pr
  MM3DGS_EXP=$E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_pr -o pr -- python /tmp/probe.py > /dev/null 2>&1
  python - "$E" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p_pr/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "sort_composite" in r["Name"] or "scatter_scan" in r["Name"] or "slam_preprocess_fwd" in r["Name"] or "project_bin" in r["Name"]:
        print("EXP", sys.argv[1], r["Name"].split("(")[0][:40], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3)
PY
done
