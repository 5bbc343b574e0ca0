#!/bin/bash
# Per-kernel durations of an arbitrary command under rocprofv3 (GPU box, from the repo root): bash tools/kstats_cmd.sh <tag> <cmd...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$TAG -o ks -- "$@" > /tmp/ks_$TAG.out 2>&1
mkdir -p gpurun_out/kstats
cp $(find /tmp/p_$TAG -name "*kernel_stats.csv" | head -1) gpurun_out/kstats/$TAG.csv
python - "$TAG" <<'PY'
import csv, sys
for r in csv.DictReader(open(f"gpurun_out/kstats/{sys.argv[1]}.csv")):
    if float(r["Percentage"]) > 0.4:
        print(f'{r["Name"].split("(")[0][:60]:62s} n={r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:7.1f} us {float(r["Percentage"]):5.1f}%  min {float(r["MinNs"])/1e3:6.1f} max {float(r["MaxNs"])/1e3:6.1f}')
PY
