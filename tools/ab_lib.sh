#!/bin/bash
# Same-box per-kernel A/B of the product library against variant builds (tools/build_variant.sh):  bash tools/ab_lib.sh <tag> [<tag> ...]
# Every library runs the bench's headline frames under rocprofv3 twice, alternating (box drift shows as a difference between the repeats).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
A="--steps 10 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0"
for rep in 1 2; do
  for t in product "$@"; do
    if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
    rm -rf /tmp/p_ab; MM3DGS_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ab -o ks -- python bench.py $A > /tmp/ks_ab.out 2>&1
    python - "$t" <<'PY'
import csv, glob, sys, re
f = glob.glob("/tmp/p_ab/**/*kernel_stats.csv", recursive=True)[0]
v = re.search(r'"value": ([0-9.]+)', open("/tmp/ks_ab.out").read())
row = {}
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    for key, pat in (("bwd", "composite_bwd_kernel<6, 1"), ("track", "fwd_bwd_track"), ("fwd", "sort_composite_fwd_kernel"), ("bproj", "slam_bwd_project"), ("ssim", "ssim_maps_kernel<false>"),
                     ("pbin", "slam_project_bin"), ("tbwd", "slam_preprocess_bwd_kernel<true, true"), ("fin", "pose_finish")):
        if pat in n and key not in row:
            row[key] = float(r["AverageNs"]) / 1e3
print(f"{sys.argv[1]:12s} {float(v.group(1)) if v else 0:6.2f} frames/s  " + "  ".join(f"{k} {row.get(k, 0):6.2f}" for k in ("bwd", "track", "fwd", "bproj", "ssim", "pbin", "tbwd", "fin")))
PY
  done
done
