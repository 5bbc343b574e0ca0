#!/bin/bash
# SQ counters of an arbitrary command (GPU box, from the repo root): bash tools/sq_counters_cmd.sh <tag> <kernel-name-filter-regex> <cmd...>
TAG=$1; FILT=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf /tmp/q_$TAG
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --kernel-trace --output-format csv -d /tmp/q_$TAG -o sq -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR \
  --kernel-trace --output-format csv -d /tmp/q2_$TAG -o sq -- "$@" > /dev/null 2>&1
python - "$TAG" "$FILT" <<'PY'
import csv, glob, collections, re, sys
tag, filt = sys.argv[1], sys.argv[2]
for d in (f"/tmp/q_{tag}", f"/tmp/q2_{tag}"):
    f = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not f:
        print("no counters in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    cols = sorted({c for v in acc.values() for c in v})
    print("| kernel | " + " | ".join(c.replace("SQ_", "") for c in cols) + " |")
    for k, v in acc.items():
        if re.search(filt, k):
            print(f"| {k[:48]} | " + " | ".join(f"{sum(v[c]) / max(len(v[c]), 1) / 1e3:.1f}k" for c in cols) + " |")
PY
