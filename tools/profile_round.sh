#!/bin/bash
# Round profile: the headline bench line, rocprofv3 kernel stats of the same command, HBM traffic counters of the same command
# (FETCH_SIZE and WRITE_SIZE in their own --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), and the bench lines +
# kernel stats of the other BASELINE configurations (C3 = UTMM-shaped RGB-D + IMU, C4 = Replica-shaped 1200x680 / 0.8 M Gaussians on one GPU,
# C5 = 1080p / 3 M Gaussians / SH degree 3).
# Run on the GPU box from the repo root: bash tools/profile_round.sh r02
set -u
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
# the counter passes run the bench's OWN default steps / warm-up (the same frames, the same map state as the line they annotate)
SHORT="--no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 --profile 0"
# pmc_pass <workload tag> <per-kernel | pass> <bench args...>: FETCH_SIZE and WRITE_SIZE in their own passes -> $OUT/<tag>_pmc_<counter>.csv
pmc_pass() {
  local W=$1 MODE=$2; shift 2
  for C in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/p_$C
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python bench.py "$@" > /tmp/p_${C}.json 2>/dev/null
    python - "$C" "$OUT" "$W" "$MODE" /tmp/p_${C}.json <<'PY'
import csv, glob, sys, collections, json
c, out, w, mode, line = sys.argv[1:6]
f = glob.glob(f"/tmp/p_{c}/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == c:
        acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
ours = ("composite", "preprocess", "slam_", "sort", "scatter", "scan", "ssim", "loss", "pose", "adam", "compact", "seed", "prune", "covisibility", "propagate", "camgrad", "tile_order")
with open(f"{out}/{w}_pmc_{c}.csv", "w") as fh:
    fh.write("kernel,launches,mean_counter_value\n")
    if mode == "pass":
        # one row: the library's kernels of ONE forward + backward pass (sum over the kernels of their per-launch means x launches per pass)
        try:
            steps = json.loads(open(line).read().strip().splitlines()[-1])
            n_pass = steps["steps"] + max(steps["warmup"], 1)
        except Exception:
            n_pass = None
        tot = sum(sum(v) for k, v in acc.items() if any(s in k for s in ours))
        launches = max(len(v) for k, v in acc.items() if "composite_bwd" in k)
        fh.write(f'"whole forward + backward pass (all library kernels)",{launches},{tot / launches}\n')
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        if any(s in k for s in ours):
            fh.write(f'"{k}",{len(v)},{sum(v)/len(v)}\n')
print(open(f"{out}/{w}_pmc_{c}.csv").read()[:1500])
PY
  done
  mkdir -p profiles
  cp $OUT/${W}_pmc_FETCH_SIZE.csv profiles/${TAG}_${W}_pmc_FETCH_SIZE.csv; cp $OUT/${W}_pmc_WRITE_SIZE.csv profiles/${TAG}_${W}_pmc_WRITE_SIZE.csv
}
pmc_pass slam kernel $SHORT
# the bench line below replays these counters as roofline.traffic
python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
# same command as the bench line (minus the follow-up runs), so the per-kernel averages are over the same frames as roofline.avg_launch_us
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -o bench -- python bench.py --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --mono-frames 0 > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
# counters of the other workloads (their bench lines below replay THEIR OWN files; round 3 replayed the configs[1] counters everywhere)
if [ -z "${SKIP_OTHER_PMC:-}" ]; then
  pmc_pass c3 kernel --workload c3 $SHORT
  pmc_pass c4 kernel --workload c4 --steps 5 --warmup 2 $SHORT
  pmc_pass c5_pass pass --workload c5 --steps 5 --warmup 2 --no-cpu-baseline
fi
python bench.py --workload c3 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --mono-frames 0 2>/dev/null | tee $OUT/bench_c3.json | cut -c1-300
python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_c4.json | cut -c1-300
python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_c5.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o c5 -- python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $OUT/bench_c5_kernel_stats.csv
ls -la $OUT
