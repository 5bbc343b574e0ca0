#!/bin/bash
# Backward-compositor generations side by side (GPU box, repo root): kernel durations of the fused engine's forward + mapping-mode
# backward (gradient outputs, no folded loss) at SLAM size under rocprofv3, once per MM3DGS_BWD2 value.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for G in ${1:-0 4}; do
  rm -rf /tmp/p_g
  MM3DGS_BWD2=$G rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_g -o g -- python tools/raster_bench.py --fused --iters 60 > /dev/null 2>&1
  python - "$G" <<'PY'
import csv, glob, sys
f = glob.glob("/tmp/p_g/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "composite_bwd" in r["Name"] or "preprocess_bwd" in r["Name"]:
        print("BWD2", sys.argv[1], r["Name"].split("(")[0][:44], r["Calls"], "avg us", round(float(r["AverageNs"]) / 1e3, 1), "min", round(float(r["MinNs"]) / 1e3, 1))
PY
done
