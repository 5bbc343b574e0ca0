#!/bin/bash
# round 5, GPU call 6: configs[3] (3225 tiles: 2.5 rounds of workgroups at five per CU) with fewer compositor workgroups per CU -- do a tile's block records then
# survive in the XCD's L2 until its combine reads them?  MM3DGS_SLAM_LDS_PAD adds never-touched dynamic LDS to the SLAM compositor launches.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5f; mkdir -p $O
for PAD in 0 12000 24000 50000; do
  MM3DGS_SLAM_LDS_PAD=$PAD timeout 300 python bench.py --workload c4 --grow-to 0 --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pad $PAD', round(d['value'], 2), 'frames/s', {k: round(v, 1) for k, v in d['kernel_us'].items()})" | tee -a $O/c4_pad.txt
done
for PAD in 0 24000; do
  rm -rf /tmp/p_f
  MM3DGS_SLAM_LDS_PAD=$PAD timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_f -o pmc -- python bench.py --workload c4 --grow-to 0 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - $PAD <<'PY' | tee -a $O/c4_pad.txt
import csv, glob, sys, collections
f = glob.glob("/tmp/p_f/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == "FETCH_SIZE":
        acc[r["Kernel_Name"].split("(")[0][:48]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:4]:
    print("pad", sys.argv[1], k, len(v), "mean FETCH_SIZE (KB)", round(sum(v) / len(v)))
PY
done
