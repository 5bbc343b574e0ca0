#!/bin/bash
# SQ counters of the SLAM render kernels (run on the GPU box from the repo root); prints a markdown table.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/p_sq
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d /tmp/p_sq -o sq -- python tools/raster_bench.py --fused --iters 5 > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/p_sq/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"]
print("| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |")
print("|---|" + "---|" * len(cols))
for k, v in acc.items():
    if any(t in k for t in ("composite", "preprocess", "scatter", "sort")):
        print(f"| {k} | " + " | ".join(f"{sum(v[c]) / max(len(v[c]), 1) / 1e6:.2f} M" if c != "SQ_WAVES" else f"{sum(v[c]) / max(len(v[c]), 1):.0f}" for c in cols) + " |")
PY
