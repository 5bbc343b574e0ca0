#!/bin/bash
# Per-kernel durations of the headline bench under rocprofv3 (run on the GPU box from the repo root).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_ks
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ks -o ks -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile 0 > /tmp/ks.json 2>/dev/null
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/p_ks/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.5:
        print(f'{r["Name"].split("(")[0][:52]:54s} n={r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:7.1f} us {float(r["Percentage"]):5.1f}%  min {float(r["MinNs"])/1e3:6.1f} max {float(r["MaxNs"])/1e3:6.1f}')
PY
