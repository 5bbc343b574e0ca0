#!/bin/bash
cd "$GRAFT_REPO_ROOT"
A="--steps 10 --warmup 2 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --profile 0"
cd /tmp && export TMPDIR=/tmp
for t in r03 r04; do
  if [ $t = r03 ]; then D=$GRAFT_REPO_ROOT/ab_r03; X=""; else D=$GRAFT_REPO_ROOT; X="--mono-frames 0"; fi
  rm -rf /tmp/p_$t; (cd $D && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$t -o ks -- python bench.py $A $X > /tmp/ks_$t.out 2>&1)
  echo "== $t $(grep -o '"value": [0-9.]*' /tmp/ks_$t.out | head -1)"
  python - "$t" <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/p_{sys.argv[1]}/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 1.0:
        print(f'  {r["Name"].split("(")[0][:56]:58s} n={r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:7.2f} us {float(r["Percentage"]):5.1f}%')
PY
done
