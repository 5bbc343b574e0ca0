#!/bin/bash
# round 5, GPU call 1: the whole GPU suite on the cleaned-up library, same-box A/B of the backward-compositor trims against the tree before
# them ("base"), the 48-byte generic record experiment on configs[4] ("rec12", with WRITE_SIZE counters), one full bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r5a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/tests.log
tail -5 $O/tests.log
timeout 600 bash tools/ab_lib.sh base > $O/ab_base.txt 2>&1; cat $O/ab_base.txt
timeout 420 bash tools/ab_c5.sh rec12 > $O/ab_c5.txt 2>&1; cat $O/ab_c5.txt
for t in product rec12; do
  if [ $t = product ]; then L=""; else L="$GRAFT_REPO_ROOT/mm3dgs_slam_amd/csrc/variants/libmm3dgs_hip_$t.so"; fi
  rm -rf /tmp/p_w
  MM3DGS_LIB=$L timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_w -o pmc -- python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python - $t <<'PY' | tee -a $O/c5_write_size.txt
import csv, glob, sys, collections
f = glob.glob("/tmp/p_w/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f[0])):
    if r.get("Counter_Name") == "WRITE_SIZE":
        acc[r["Kernel_Name"].split("(")[0][:60]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:6]:
    print(sys.argv[1], k, len(v), "mean WRITE_SIZE", sum(v) / len(v))
PY
done
timeout 600 python bench.py 2> $O/bench.err | tee $O/bench.json | cut -c1-400
