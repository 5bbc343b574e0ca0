#!/bin/bash
# A/B of MM3DGS_EXP probe bits on a STATIC scene (tools/raster_bench.py --fused: the map does not evolve, so invalid gradients do not
# change the workload): average kernel times per value.   bash tools/exp_ab.sh 0 16 32
# Needs the probe build: tools/build_variant.sh probes -DMM3DGS_PROBES (the product library carries no probe).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MM3DGS_LIB=$PWD/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_probes.so
[ -f "$MM3DGS_LIB" ] || { echo "build the probe library first: tools/build_variant.sh probes -DMM3DGS_PROBES"; exit 1; }
for E in "$@"; do
  rm -rf /tmp/p_exp
  MM3DGS_EXP=$E rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_exp -o e -- python tools/raster_bench.py --fused --iters 60 > /dev/null 2>&1
  python - "$E" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p_exp/**/*kernel_stats.csv", recursive=True)[0]
out = []
for r in csv.DictReader(open(f)):
    n = r["Name"]
    if any(t in n for t in ("composite_bwd", "slam_preprocess_bwd", "sort_composite_fwd", "project_bin")) and int(r["Calls"]) >= 50:
        out.append(f"{n.split('(')[0][-40:]} {float(r['AverageNs'])/1e3:.1f}")
print("EXP", sys.argv[1], " | ".join(out))
PY
done
