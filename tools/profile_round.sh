#!/bin/bash
# Round profile: rocprofv3 kernel stats of the headline bench + HBM traffic counters of the raster micro-bench.
# Run on the GPU box from the repo root: bash tools/profile_round.sh r01
# (counters are collected in their own passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes)
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -o bench -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile 2 > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python tools/raster_bench.py --fused --iters 10 > /dev/null 2>&1
  python - "$C" "$OUT" <<'PY'
import csv, glob, sys, collections
c, out = sys.argv[1], sys.argv[2]
f = glob.glob(f"/tmp/p_{c}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
with open(f"{out}/slam_pmc_{c}.csv", "w") as fh:
    fh.write("kernel,launches,mean_counter_value\n")
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if "composite" in k or "preprocess" in k or "sort" in k or "scatter" in k or "scan" in k:
            fh.write(f'"{k}",{len(v)},{sum(v)/len(v)}\n')
print(open(f"{out}/slam_pmc_{c}.csv").read())
PY
done
ls -la $OUT
