#!/bin/bash
# Round profile: the headline bench line, rocprofv3 kernel stats of the same command, HBM traffic counters of the same command
# (FETCH_SIZE and WRITE_SIZE in their own --pmc passes, as /opt/skills/guides/MI355X_MICROARCH.md prescribes), and the bench lines +
# kernel stats of the other BASELINE configurations (C3 = UTMM-shaped RGB-D + IMU, C4 = Replica-shaped 1200x680 / 0.8 M Gaussians on one GPU,
# C5 = 1080p / 3 M Gaussians / SH degree 3).
# Run on the GPU box from the repo root: bash tools/profile_round.sh r02
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG; mkdir -p $OUT
# the counter passes run the bench's OWN default steps / warm-up (the same frames, the same map state as the line they annotate)
SHORT="--no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 --profile 0"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/p_$C -o pmc -- python bench.py $SHORT > /dev/null 2>&1
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
        if any(s in k for s in ("composite", "preprocess", "slam_", "sort", "scatter", "scan", "ssim", "loss", "pose", "adam", "compact", "seed", "prune", "covisibility", "propagate")):
            fh.write(f'"{k}",{len(v)},{sum(v)/len(v)}\n')
print(open(f"{out}/slam_pmc_{c}.csv").read())
PY
done
# the bench line below replays these counters as roofline.traffic
mkdir -p profiles; cp $OUT/slam_pmc_FETCH_SIZE.csv profiles/${TAG}_slam_pmc_FETCH_SIZE.csv; cp $OUT/slam_pmc_WRITE_SIZE.csv profiles/${TAG}_slam_pmc_WRITE_SIZE.csv
python bench.py 2>$OUT/bench.err | tee $OUT/bench.json | cut -c1-300
# same command as the bench line (minus the follow-up runs), so the per-kernel averages are over the same frames as roofline.avg_launch_us
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_bench -o bench -- python bench.py --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 --moving-frames 0 > $OUT/bench_under_rocprof.json 2>/dev/null
cp $(find /tmp/p_bench -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv
python bench.py --workload c3 --no-cpu-baseline --full-seed-steps 0 --steady-frames 0 2>/dev/null | tee $OUT/bench_c3.json | cut -c1-300
python bench.py --workload c4 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_c4.json | cut -c1-300
python bench.py --workload c5 --no-cpu-baseline 2>/dev/null | tee $OUT/bench_c5.json | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_c5 -o c5 -- python bench.py --workload c5 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
cp $(find /tmp/p_c5 -name "*kernel_stats.csv" | head -1) $OUT/bench_c5_kernel_stats.csv
ls -la $OUT
