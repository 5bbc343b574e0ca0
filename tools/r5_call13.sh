#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5m; mkdir -p $O
for rep in 1 2; do for t in product binprio; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  rm -rf /tmp/p_mv
  MM3DGS_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_mv -o mv -- python tools/moving_run.py --frames 60 --every 100 > /tmp/mv.out 2>/dev/null
  python - $t <<'PY' | tee -a $O/moving_binprio.txt
import csv, glob, sys
f = glob.glob("/tmp/p_mv/**/*kernel_stats.csv", recursive=True)[0]
row = {}
for r in csv.DictReader(open(f)):
    n = r["Name"].split("(")[0]
    for key, pat in (("bproj", "slam_bwd_project"), ("pbin", "slam_project_bin"), ("tbwd", "slam_preprocess_bwd_kernel<true, true")):
        if pat in n and key not in row: row[key] = float(r["AverageNs"]) / 1e3
print(sys.argv[1], open("/tmp/mv.out").read().strip().splitlines()[-1][:60], " ".join(f"{k} {v:.2f}" for k, v in row.items()))
PY
done; done
