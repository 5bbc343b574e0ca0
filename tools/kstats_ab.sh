#!/bin/bash
# Per-kernel durations of a short headline bench under rocprofv3, once per value of an environment switch.
# usage (GPU box, repo root): bash tools/kstats_ab.sh VAR "v1 v2 ..."      e.g. MM3DGS_NO_DIRECT_BINS "0 1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
VAR=${1:-MM3DGS_NO_DIRECT_BINS}; VALS=${2:-"0 1"}
for V in $VALS; do
  rm -rf /tmp/p_ks
  env $VAR=$V rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ks -o ks -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --profile 0 > /tmp/ks.json 2>/dev/null
  python - "$VAR=$V" <<'PY'
import csv, glob, json, sys
d = json.load(open("/tmp/ks.json"))
print(sys.argv[1], "fps under rocprof", round(d["value"], 2))
f = glob.glob("/tmp/p_ks/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 0.7:
        print(f'  {r["Name"].split("(")[0][:52]:54s} n={r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:7.1f} us {float(r["Percentage"]):5.1f}%  min {float(r["MinNs"])/1e3:6.1f}')
PY
done
