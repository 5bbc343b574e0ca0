#!/bin/bash
# configs[4] alone (the generic path): the bench line, rocprofv3 kernel stats and the FETCH_SIZE / WRITE_SIZE passes of the same command -> profiles/<tag>_*c5*
# (what tools/profile_round.sh collects for this workload; run on the GPU box from the repo root: bash tools/c5_profile.sh r06)
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT profiles
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > /tmp/p_${C}.json 2>/dev/null
  python - "$C" "$OUT" <<'PY'
import csv, glob, sys, collections
c, out = sys.argv[1:3]
f = glob.glob(f"/tmp/p_{c}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
ours = ("composite", "preprocess", "slam_", "sort", "scatter", "scan", "ssim", "loss", "pose", "adam", "compact", "seed", "prune", "covisibility", "propagate", "camgrad", "tile_order")
with open(f"{out}/c5_pass_pmc_{c}.csv", "w") as fh:
    fh.write("kernel,launches,mean_counter_value\n")
    tot = sum(sum(v) for k, v in acc.items() if any(s in k for s in ours))
    launches = max(len(v) for k, v in acc.items() if "composite_bwd" in k)
    fh.write(f'"whole forward + backward pass (all library kernels)",{launches},{tot / launches}\n')
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if any(s in k for s in ours):
            fh.write(f'"{k}",{len(v)},{sum(v)/len(v)}\n')
print(open(f"{out}/c5_pass_pmc_{c}.csv").read()[:900])
PY
  cp $OUT/c5_pass_pmc_$C.csv profiles/${TAG}_c5_pass_pmc_$C.csv
done
python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_c5.json | cut -c1-300
rm -rf /tmp/p_c5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o c5 -- python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $OUT/bench_c5_kernel_stats.csv
