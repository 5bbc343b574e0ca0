#!/bin/bash
# Developer A/B on a STATIC workload (round 6): the benchmark map after N frames is built ONCE with the product library and saved; then every library
# (product + variant builds of tools/build_variant.sh, compile-time timing probes included -- their results are invalid, which a static workload does not
# mind) runs 60 gradient-output mapping iterations and 60 tracking iterations with the pose learning rates at 0 on it under rocprofv3.
#   bash tools/static_ab.sh <frames> <tag> [<tag> ...]        (tag "product" = the product library)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
N=${1:-8}; shift
cat > /tmp/static_ab.py <<PY
import os, sys, random
sys.path.insert(0, ".")
import numpy as np, torch
from mm3dgs_slam_amd.config import default_config
from mm3dgs_slam_amd.fused import FusedEngine, _loss_cfg
from mm3dgs_slam_amd.gaussian_model import GaussianModel
from mm3dgs_slam_amd.renderer import Renderer
from mm3dgs_slam_amd.slam import SLAM, SyntheticSequence
frames = $N
cfg = default_config(device="cuda", height=480, width=640, mapping={"seed_fraction": 0.51})
path = "/tmp/static_ab_map.pt"
if not os.path.exists(path):
    torch.manual_seed(0); random.seed(0); np.random.seed(0)
    seq = SyntheticSequence(cfg, frames + 1, 150000, seed=0)
    slam = SLAM(cfg, seq)
    for i in range(frames):
        slam.step(i)
    g = slam.gaussians
    color, depth, _ = seq[frames - 1]
    torch.save({"xyz": g._xyz.detach(), "f_dc": g._features_dc.detach(), "opacity": g._opacity.detach(), "scaling": g._scaling.detach(), "rotation": g._rotation.detach(),
                "pose": slam.estimate_pose_list[frames - 1].detach().float(), "color": color, "depth": depth}, path)
    sys.exit(0)
d = torch.load(path)
g = GaussianModel(cfg); g.training_setup()
P = d["xyz"].shape[0]
g.densification_postfix(d["xyz"], d["f_dc"], torch.zeros(P, 0, 3, device="cuda"), d["opacity"], d["scaling"], d["rotation"], torch.zeros(P, 3, device="cuda"))
eng = FusedEngine(Renderer(cfg))
pose, color, depth = d["pose"].contiguous(), d["color"].contiguous(), d["depth"].contiguous()
m = cfg["mapping"]
lcfg = _loss_cfg(eng.H, eng.W, 1.0 - m["lambda_dssim"], m["lambda_dssim"], float(m["pearson_weight"]), 0, 2, 0, 0.5)
view = (pose, color, depth)
from mm3dgs_slam_amd import _lib
with torch.no_grad():
    eng.forward(pose, g, need_grads=True); assert eng.check_capacity()
    eng.forward(pose, g, need_grads=True); assert eng.check_capacity()
    for _ in range(60):
        eng.map_loop([view], g, lcfg, None, None, grads=eng.grads)
    tcfg = _loss_cfg(eng.H, eng.W, 1.0, 0.0, 0.0, 1, 0, 1, 0.99)
    p = pose.clone(); mm, vv = torch.zeros(7, device="cuda"), torch.zeros(7, device="cuda"); st = torch.zeros(1, dtype=torch.int32, device="cuda")
    ad = _lib.Mm3dgsPoseAdam()
    ad.pose, ad.m, ad.v, ad.step = p.data_ptr(), mm.data_ptr(), vv.data_ptr(), st.data_ptr()
    ad.lr_q, ad.lr_t, ad.beta1, ad.beta2, ad.eps = 0.0, 0.0, 0.9, 0.999, 1e-8
    eng.track_loop(60, p, g, tcfg, color, None, ad)
torch.cuda.synchronize()
PY
rm -f /tmp/static_ab_map.pt
python /tmp/static_ab.py > /tmp/static_ab_build.out 2>&1 || { tail -5 /tmp/static_ab_build.out; exit 1; }
for rep in 1 2; do
for t in "$@"; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  rm -rf /tmp/p_sk
  MM3DGS_LIB=$L rocprofv3 --kernel-trace --output-format csv -d /tmp/p_sk -o sk -- python /tmp/static_ab.py > /tmp/sk.out 2>&1
  python - "$t" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p_sk/**/*kernel_trace.csv", recursive=True)
if not f:
    print(sys.argv[1], "no trace:", open("/tmp/sk.out").read()[-400:]); sys.exit(0)
rows = [r for r in csv.DictReader(open(f[0]))]
out = []
for key in ("composite_bwd_kernel<6, 1", "sort_composite_fwd_kernel", "sort_composite_fwd_bwd_track", "slam_preprocess_bwd"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if key in r["Kernel_Name"]][-50:]
    if d: out.append(f"{key.split('<')[0][-24:]} {sum(d) / len(d):6.2f}")
print(f"{sys.argv[1]:10s}", " | ".join(out))
PY
done
done
