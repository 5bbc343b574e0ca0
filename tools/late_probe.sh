#!/bin/bash
# Developer probe: kernel times of forward launches on the map after N frames of the bench run, with MM3DGS_EXP timing bits (results invalid
# by construction when a bit is set).   PROBE_EXPS="0 64" bash tools/late_probe.sh [frames]
# Needs the probe build: tools/build_variant.sh probes -DMM3DGS_PROBES (the product library carries no probe).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_probes.so
[ -f "$MM3DGS_LIB" ] || { echo "build the probe library first: tools/build_variant.sh probes -DMM3DGS_PROBES"; exit 1; }
N=${1:-100}
cat > /tmp/late_probe.py <<PY
import os, sys, random
sys.path.insert(0, ".")
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import _engine
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
frames = $N
exp = os.environ.pop("PROBE_EXP", "0")
cfg = default_config(device="cuda", height=480, width=640, mapping={"seed_fraction": 0.51})
torch.manual_seed(0); random.seed(0); np.random.seed(0)
seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0)
slam = SLAM(cfg, seq)
for i in range(frames):
    slam.step(i)
eng = _engine(slam.renderer)
pose = slam.estimate_pose_list[frames - 1].detach().float().contiguous()
torch.cuda.synchronize()
os.environ["MM3DGS_EXP"] = exp
import ctypes
with torch.no_grad():
    for _ in range(40):
        eng.forward(pose, slam.gaussians, need_grads=True)
torch.cuda.synchronize()
PY
for E in ${PROBE_EXPS:-0 64}; do
  rm -rf /tmp/p_lp
  PROBE_EXP=$E rocprofv3 --kernel-trace --output-format csv -d /tmp/p_lp -o lp -- python /tmp/late_probe.py > /dev/null 2>&1
  python - "$E" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p_lp/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
# the last 40 launches of each kernel of interest = the probe's forwards
for key in ("slam_project_bin_kernel", "sort_composite_fwd_kernel"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]][-40:]
    if d: print("EXP", sys.argv[1], key, "n", len(d), "avg us %.2f min %.2f" % (sum(d) / len(d), min(d)))
PY
done
