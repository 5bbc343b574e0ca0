#!/bin/bash
# Developer probe (round 5): where do the mapping iteration's compositor launches spend their time?  The benchmark map after N frames, then 60
# gradient-output mapping iterations (no optimiser step: a static workload) under rocprofv3, per MM3DGS_EXP probe value (results INVALID when a
# bit is set; needs the probe build: tools/build_variant.sh probes -DMM3DGS_PROBES).
#   PROBE_EXPS="0 512 1024 32 1536" bash tools/skeleton_probe.sh [frames]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_probes.so
[ -f "$MM3DGS_LIB" ] || { echo "build the probe library first: tools/build_variant.sh probes -DMM3DGS_PROBES"; exit 1; }
N=${1:-8}
cat > /tmp/skeleton_probe.py <<PY
import os, sys, random
sys.path.insert(0, ".")
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import _engine, _loss_cfg
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
g = slam.gaussians
pose = slam.estimate_pose_list[frames - 1].detach().float().contiguous()
color, depth, _ = seq[frames - 1]
m = cfg["mapping"]
lcfg = _loss_cfg(eng.H, eng.W, 1.0 - m["lambda_dssim"], m["lambda_dssim"], float(m["pearson_weight"]), 0, 2, 0, 0.5)
view = (pose, color.contiguous(), depth.contiguous())
eng._ensure(int(g._xyz.shape[0]), True)
torch.cuda.synchronize()
os.environ["MM3DGS_EXP"] = exp
from mm3dgs_slam_amd import _lib
with torch.no_grad():
    for _ in range(60):
        eng.map_loop([view], g, lcfg, None, None, grads=eng.grads)
    # tracking iterations with the learning rates at 0 (the pose stays put: a static workload whatever the probes do to the gradient)
    tcfg = _loss_cfg(eng.H, eng.W, 1.0, 0.0, 0.0, 1, 0, 1, 0.99)
    p = pose.clone(); mm, vv = torch.zeros(7, device="cuda"), torch.zeros(7, device="cuda"); st = torch.zeros(1, dtype=torch.int32, device="cuda")
    ad = _lib.Mm3dgsPoseAdam()
    ad.pose, ad.m, ad.v, ad.step = p.data_ptr(), mm.data_ptr(), vv.data_ptr(), st.data_ptr()
    ad.lr_q, ad.lr_t, ad.beta1, ad.beta2, ad.eps = 0.0, 0.0, 0.9, 0.999, 1e-8
    eng.track_loop(60, p, g, tcfg, color.contiguous(), None, ad)
torch.cuda.synchronize()
PY
for E in ${PROBE_EXPS:-0 512 1024 32 1536 1568}; do
  rm -rf /tmp/p_sk
  PROBE_EXP=$E rocprofv3 --kernel-trace --output-format csv -d /tmp/p_sk -o sk -- python /tmp/skeleton_probe.py > /tmp/sk.out 2>&1
  python - "$E" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p_sk/**/*kernel_trace.csv", recursive=True)
if not f:
    print("EXP", sys.argv[1], "no trace:", open("/tmp/sk.out").read()[-400:]); sys.exit(0)
rows = [r for r in csv.DictReader(open(f[0]))]
out = []
for key in ("composite_bwd_kernel<6, 1", "sort_composite_fwd_kernel", "sort_composite_fwd_bwd_track", "ssim_maps", "slam_preprocess_bwd", "slam_project_bin"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]][-50:]
    if d: out.append(f"{key.split('<')[0][-24:]} {sum(d) / len(d):6.2f}")
print("EXP", sys.argv[1].rjust(5), " | ".join(out))
PY
done
